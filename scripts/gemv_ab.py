#!/usr/bin/env python3
"""A/B of two GEMV geometries on the token's launch shapes, alternating in one process (box-to-box and run-to-run differences are 3 %,
as large as the effect): graphs of 128 launches rotating over the 32 layers' weights, zero-point-8 kernels, best and median of 7 rounds."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tinychatengine_amd import capi
from tinychatengine_amd.decode import SHAPES, DecodeLinears
dev = torch.device("cuda:0"); L = capi.lib()
dl = DecodeLinears(SHAPES[sys.argv[1] if len(sys.argv) > 1 else "baseline-named"], device=dev, group_size=128)
cfgs = {"rows4_waves4": (4, 4, 1, 1), "rows2_waves4": (2, 4, 1, 1), "rows2_waves4_d2": (2, 4, 1, 2), "rows2_waves8_d2": (2, 8, 1, 2), "rows4_waves8": (4, 8, 1, 1)}
if os.environ.get("TCE_AB_WIDE"):  # every compiled row-block geometry without a K split, and the K splits for the small launches
    cfgs.update({"rows1_waves4": (1, 4, 1, 1), "rows1_waves4_d2": (1, 4, 1, 2), "rows4_waves4_d2": (4, 4, 1, 2), "rows2_waves4_d3": (2, 4, 1, 3),
                 "rows2_2x2": (2, 2, 2, 1), "rows2_2x2_d2": (2, 2, 2, 2), "rows1_2x2": (1, 2, 2, 1)})
def graph_for(which, cfg):
    capi.set_gemv_config(*cfg)
    groups = [dl.block_launches(li)[which] for li in range(dl.n_layers)]
    arrs = [(capi.W4A16Desc * len(g))(*g) for g in groups]
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sp = C.c_void_p(s.cuda_stream)
        with torch.cuda.graph(g, stream=s):
            for i in range(128):
                capi.check(L.tce_w4a16_forward_group(arrs[i % len(arrs)], len(groups[0]), sp))
    capi.set_gemv_config()
    return g, arrs
def t(g):
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * 128)  # (lm_head row: per 128 launches of a 32 x 4 graph -- compare the columns, not the value)
def lm_graph(cfg):
    capi.set_gemv_config(*cfg)
    d = dl.lm_head.desc(dl.x, dl.logits)
    lm = (capi.W4A16Desc * 1)(d)
    gu = [dl.block_launches(li)[2] for li in range(dl.n_layers)]
    arrs = [(capi.W4A16Desc * len(g))(*g) for g in gu]
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sp = C.c_void_p(s.cuda_stream)
        with torch.cuda.graph(g, stream=s):
            for i in range(32):  # lm_head is ONE tensor: three gate+up launches (default geometry) flush the memory-side cache in between
                capi.check(L.tce_w4a16_forward(C.byref(lm[0]), sp))
                capi.set_gemv_config()
                for j in range(3):
                    capi.check(L.tce_w4a16_forward_group(arrs[(3 * i + j) % len(arrs)], len(gu[0]), sp))
                capi.set_gemv_config(*cfg)
    capi.set_gemv_config()
    return g, (lm, arrs)
for which, name in ((0, "qkv"), (1, "o"), (2, "gate+up"), (3, "down"), (-1, "lm_head + 3 gate+up (relative only)")):
    gs = {k: (graph_for(which, v) if which >= 0 else lm_graph(v)) for k, v in cfgs.items()}
    res = {k: [] for k in cfgs}
    for rnd in range(7):
        for k in cfgs:
            res[k].append(t(gs[k][0]))
    print(json.dumps({"launch": name, **{k: [round(min(v), 2), round(float(np.median(v)), 2)] for k, v in res.items()}}), flush=True)
