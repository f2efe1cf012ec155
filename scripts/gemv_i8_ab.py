#!/usr/bin/env python3
"""A/B of the two decode kernel families on the token's launch shapes, alternating in one process: the fp16-unpack GEMV on the q4_6 arrays (round 1-3) against the
int8-contraction GEMV on the packed copy (round 4, csrc/w4a16_gemv_i8.hip), and the latter with one / two tiles per wave.  Graphs of 128 launches rotating over the 32
layers' weights (3.4 GB: nothing stays in the 256 MiB memory-side cache), zero-point-8 kernels, best and median of 7 rounds, us per launch + fraction of 8 TB/s."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tinychatengine_amd import capi
from tinychatengine_amd.decode import SHAPES, DecodeLinears
dev = torch.device("cuda:0"); L = capi.lib()
model = sys.argv[1] if len(sys.argv) > 1 else "baseline-named"
dl = DecodeLinears(SHAPES[model], device=dev, group_size=128, prepack=True)
variants = {"fp16_gemv": (1, 0), "i8_rule": (0, 0), "i8_rows1": (0, 1), "i8_rows2": (0, 2)}
def graph_for(which, var):
    capi.set_gemv_i8(*var)
    groups = [dl.block_launches(li)[which] for li in range(dl.n_layers)]
    arrs = [(capi.W4A16Desc * len(g))(*g) for g in groups]
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sp = C.c_void_p(s.cuda_stream)
        with torch.cuda.graph(g, stream=s):
            for i in range(128):
                capi.check(L.tce_w4a16_forward_group(arrs[i % len(arrs)], len(groups[0]), sp))
    capi.set_gemv_i8()
    return g, arrs
def t(g):
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * 128)
for which, name in ((0, "qkv"), (1, "o"), (2, "gate+up"), (3, "down")):
    gs = {k: graph_for(which, v) for k, v in variants.items()}
    bytes_ = sum(capi.algorithmic_bytes(1, d.N, d.K, 128) for d in dl.block_launches(0)[which])
    res = {k: [] for k in variants}
    for rnd in range(7):
        for k in variants:
            res[k].append(t(gs[k][0]))
    print(json.dumps({"model": model, "launch": name, "algorithmic_bytes": bytes_,
                      **{k: {"us_min": round(min(v), 2), "us_median": round(float(np.median(v)), 2), "frac_of_8TBps_median": round(bytes_ / float(np.median(v)) * 1e-6 / 8, 3)} for k, v in res.items()}}), flush=True)
# the whole token as a plan: 129 launches
for var_name, var in (("fp16_gemv", (1, 0)), ("i8_rule", (0, 0))):
    capi.set_gemv_i8(*var)
    plan = dl.make_plan()
    capi.set_gemv_i8()
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5): plan.launch(s)
    torch.cuda.synchronize()
    ts = []
    for rnd in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): plan.launch(s)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20)
    print(json.dumps({"model": model, "token_plan": var_name, "ms_per_token_min": round(min(ts), 4), "tokens_per_s": round(1e3 / min(ts), 1), "token_bytes": dl.token_bytes(),
                      "frac_of_8TBps": round(dl.token_bytes() / min(ts) * 1e-9 / 8, 3)}), flush=True)
