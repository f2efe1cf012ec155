#!/usr/bin/env python3
"""A/B of the GEMV kernels on the FUSED launches of a decoder layer (RMSNorm prologue, SiLU*mul pairs, residual add) -- which kernel
should carry them?  Alternating in one process over the 32 layers' weights (graphs of 128 launches), best / median of 5 rounds."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tinychatengine_amd import capi
from tinychatengine_amd.decoder_block import DecoderBlock
dev = torch.device("cuda:0"); L = capi.lib()
hidden, heads, ffn = 4096, 32, 11008
kvh = None
if os.environ.get("FUSED_AB_SHAPES") == "llama3-8b": ffn, kvh = 14336, 8  # (default: the 7B-shaped layer BASELINE.json spells)
cos = torch.zeros(64, 128, dtype=torch.float16, device=dev); sin = torch.zeros_like(cos)
blocks = [DecoderBlock(hidden, heads, ffn, 64, dev, cos, sin, seed=i, kv_heads=kvh) for i in range(32)]
hid = torch.randn(1, hidden, device=dev).to(torch.float16)
def descs(kind):
    out = []
    for b in blocks:
        if kind == "norm+qkv": out.append(b.qkv.desc(hid, b.qkv_out, gamma=b.gamma1, eps=1e-6))
        elif kind == "plain qkv": out.append(b.qkv.desc(hid, b.qkv_out))
        elif kind == "o+add": out.append(b.o.desc(b.attn_out, hid, flags=capi.TCE_W4_ADD_TO_C))
        elif kind == "norm+gate/up+silu": out.append(b.gate_up.desc(hid, b.act, flags=capi.TCE_W4_SILU_MUL_PAIRS, gamma=b.gamma2, eps=1e-6))
        elif kind == "gate/up+silu (no norm)": out.append(b.gate_up.desc(hid, b.act, flags=capi.TCE_W4_SILU_MUL_PAIRS))
        elif kind == "down+add": out.append(b.down.desc(b.act, hid, flags=capi.TCE_W4_ADD_TO_C))
    return out
cfgs = {"auto": None, "rowblock_2_4_d2": (2, 4, 1, 2), "rowblock_2_4_d1": (2, 4, 1, 1), "rowblock_4_4_d1": (4, 4, 1, 1), "persistent_2x16_d2": (2, 16, 0, 2), "persistent_4x16_d2": (4, 16, 0, 2),
        "rowblock_4_8_d1": (4, 8, 1, 1), "rowblock_2_8_d2": (2, 8, 1, 2)}
if os.environ.get("FUSED_AB_I8_TILES"): capi.set_gemv_i8(0, int(os.environ["FUSED_AB_I8_TILES"]))  # (tiles per wave of the int8-MFMA kernel, forced)
if os.environ.get("FUSED_AB_AUTO_ONLY"): cfgs = {"auto": None}  # (pre-packed weights: the dispatch takes the int8-MFMA kernel, the knobs above do not apply)
def graph(ds, cfg):
    capi.set_gemv_config(*(cfg or (0, 0, 0, 0)))
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    try:
        with torch.cuda.stream(s):
            sp = C.c_void_p(s.cuda_stream)
            with torch.cuda.graph(g, stream=s):
                for i in range(128):
                    capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp))
    finally:
        capi.set_gemv_config()
    return g
def t(g):
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * 128)
for kind in ("norm+qkv", "plain qkv", "o+add", "norm+gate/up+silu", "gate/up+silu (no norm)", "down+add"):
    ds = descs(kind)
    gs = {}
    for k, v in cfgs.items():
        try:
            gs[k] = graph(ds, v)
        except Exception as e:
            L.tce_reset_last_error()
    res = {k: [] for k in gs}
    for rnd in range(5):
        for k in gs: res[k].append(t(gs[k]))
    print(json.dumps({"launch": kind, "auto_is": capi.describe_dispatch(ds[0]), **{k: [round(min(v), 2), round(float(np.median(v)), 2)] for k, v in res.items()}}), flush=True)
