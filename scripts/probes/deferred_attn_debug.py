#!/usr/bin/env python3
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_gpu_deferred_attention import _setup, _plain, _deferred
from tinychatengine_amd import capi
dev = torch.device("cuda:0")
atts, o, qkv, res = _setup(dev, 32, 8, 2048, seed=43)
pos = int(sys.argv[1]) if len(sys.argv) > 1 else 320
x0, y0 = _plain(atts[0], o, qkv, res, pos)
x1, y1, slots = _deferred(atts[1], o, qkv, res, pos)
torch.cuda.synchronize()
info = atts[1].deferred
print("slots", info.slots, "chunk", info.chunk, "heads", info.heads, "stride", info.stride)
ws = atts[1].workspace
cnt_bytes = (32 * 4 + 255) & ~255
part = ws[cnt_bytes:cnt_bytes + 32 * info.slots * 132 * 4].view(torch.float32).cpu().numpy().reshape(32, info.slots, 132)
nact = min((pos + info.chunk) // info.chunk, info.slots)
M = part[:, :nact, 0]; Lp = part[:, :nact, 1]; O = part[:, :nact, 4:]
Mx = M.max(axis=1, keepdims=True); w = np.exp(M - Mx)
x_np = ((O * w[..., None]).sum(axis=1) / (Lp * w).sum(axis=1, keepdims=True)).astype(np.float16)
x_ref = x0.cpu().numpy().reshape(32, 128)
print("partials -> x vs plain x: max abs diff", float(np.abs(x_np.astype(np.float32) - x_ref.astype(np.float32)).max()), "M range", float(M.min()), float(M.max()), "L range", float(Lp.min()), float(Lp.max()))
print("y diff max", float((y0.float() - y1.float()).abs().max()), "y0[:4]", y0[0, :4].tolist(), "y1[:4]", y1[0, :4].tolist())
if os.environ.get("TCE_LIB_PATH"):
    dbg = x1.view(torch.float32).cpu().numpy().reshape(-1)[: 64 * 12].reshape(64, 12)
    for lane in (0, 1, 15, 16, 17, 63):
        kpos = lane * 8; h, d0 = kpos >> 7, kpos & 127
        print("lane", lane, "head", h, "d0", d0, "kernel M,L", dbg[lane, 0], dbg[lane, 1], "mem M,L", part[h, 1, 0], part[h, 1, 1])
        print("    kernel O", np.round(dbg[lane, 2:10], 4).tolist(), "\n    mem    O", np.round(part[h, 1, 4 + d0: 12 + d0], 4).tolist(), "Mx, Lx", dbg[lane, 10], dbg[lane, 11])
