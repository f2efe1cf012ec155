#!/usr/bin/env python3
"""Where do two forced forms of the pre-packed GEMM differ?  usage: pk_form_diff.py M N K modeA modeB"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
M, N, K, ma, mb = [int(v) for v in sys.argv[1:6]]
dev = torch.device("cuda:0"); L = capi.lib()
g = torch.Generator(device=dev).manual_seed(91 + N + M)
lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack()
x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
outs = {}
for mode in (ma, mb):
    capi.check(L.tce_w4a16_set_debug_mode(mode))
    y = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
    d = lin.desc(x, y)
    print(mode, capi.describe_dispatch(d))
    capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    outs[mode] = y.float().cpu().numpy()
L.tce_w4a16_set_debug_mode(60)
a, b = outs[ma], outs[mb]
bad = np.argwhere(a != b)
print("differing elements", len(bad), "of", a.size, "max |diff|", float(np.abs(a - b).max()) if len(bad) else 0.0, "nan", int(np.isnan(b).sum()))
if len(bad):
    rows, cols = np.unique(bad[:, 0]), np.unique(bad[:, 1])
    print("rows: n", len(rows), rows[:20], "... cols: n", len(cols), cols[:40])
    print("row%128 hist", np.bincount(bad[:, 0] % 128, minlength=128).nonzero()[0][:40])
    print("col%512 hist", np.bincount(bad[:, 1] % 512, minlength=512).nonzero()[0][:64])
    for r, c in bad[:5]:
        print((int(r), int(c)), a[r, c], b[r, c])
