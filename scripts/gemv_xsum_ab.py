#!/usr/bin/env python3
"""In-process A/B of the row-block GEMV's issue order on the token's launches: the rule (x first when the grid is one generation of
workgroups, ...) -- HERE: the per-chunk activation sums by every wave (45) against once per workgroup through an LDS table (47; 46 = the rule) (tce_w4a16_set_debug_mode; an earlier version of this script used 40 / 41, which the GEMM's XCD-rows modes shadowed: that run compared a kernel with itself).
   gpurun -- 'python scripts/gemv_xsum_ab.py > gpurun_out/gemv_xsum_ab.jsonl'"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tinychatengine_amd import capi
from tinychatengine_amd.decode import SHAPES, DecodeLinears
dev = torch.device("cuda:0"); L = capi.lib()
dl = DecodeLinears(SHAPES[sys.argv[1] if len(sys.argv) > 1 else "baseline-named"], device=dev, group_size=128)
def graph_for(which, mode):
    capi.check(L.tce_w4a16_set_debug_mode(mode))
    groups = [dl.block_launches(li)[which] for li in range(dl.n_layers)]
    arrs = [(capi.W4A16Desc * len(g))(*g) for g in groups]
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sp = C.c_void_p(s.cuda_stream)
        with torch.cuda.graph(g, stream=s):
            for i in range(128):
                capi.check(L.tce_w4a16_forward_group(arrs[i % len(arrs)], len(groups[0]), sp))
    capi.check(L.tce_w4a16_set_debug_mode(46))
    return g, arrs
def t(g):
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * 128)
for which, name in ((0, "qkv"), (1, "o"), (2, "gate+up"), (3, "down")):
    gs = {k: graph_for(which, m) for k, m in (("every_wave", 45), ("shared_table", 47))}
    res = {k: [] for k in gs}
    for rnd in range(7):
        for k in gs:
            res[k].append(t(gs[k][0]))
    print(json.dumps({"launch": name, **{k: [round(min(v), 2), round(float(np.median(v)), 2)] for k, v in res.items()}}), flush=True)
