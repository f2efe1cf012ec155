#!/usr/bin/env python3
"""Prefill GEMM on the packed copy, round 5: the 256-row wave tiles (forms 6 / 7: tce_w4a16_set_debug_mode 66, 672..674) against the best 128-row form and the
dispatcher's choice, per (M, N, K): us per launch and TFLOP/s, weights rotating over three copies, the forms alternating inside one process.
usage: gemm_pk256_sweep.py [MxNxK ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0"); L = capi.lib(); st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1000.0 / reps)
    return min(ts)
g = torch.Generator(device=dev).manual_seed(1)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
if not shapes:
    shapes = [(M, N, K) for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (14336, 4096), (4096, 14336)) for M in (256, 512, 1024, 2048, 4096)]
cache = {}
for (M, N, K) in shapes:
    if (N, K) not in cache:
        cache.clear(); torch.cuda.empty_cache()
        cache[(N, K)] = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
    lins = cache[(N, K)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    descs = [l.desc(x, out) for l in lins]; it = [0]
    def run():
        capi.check(capi.w4a16_forward(descs[it[0] % 3], st)); it[0] += 1
    row = {"M": M, "N": N, "K": K}
    L.tce_w4a16_set_debug_mode(690)
    for name, mode in (("rows128_auto", 60), ("rows256_whole", 66), ("rows256_cut2", 672), ("rows256_cut3", 673), ("rows256_cut4", 674), ("rows128_auto_again", 60)):
        L.tce_w4a16_set_debug_mode(mode)
        try:
            row[name] = round(timed(run), 2)
            if mode != 60: row[name + "_is"] = " ".join(capi.describe_dispatch(descs[0]).split()[1:4])
        except Exception as e:  # noqa: BLE001
            row[name] = None; L.tce_reset_last_error()
        L.tce_w4a16_set_debug_mode(60)
    L.tce_w4a16_set_debug_mode(691)
    row["auto"] = round(timed(run), 2); row["auto_is"] = " ".join(capi.describe_dispatch(descs[0]).split()[:4])
    fl = 2.0 * M * N * K
    b128 = min(row["rows128_auto"], row["rows128_auto_again"])
    b256 = min(v for k, v in row.items() if k.startswith("rows256") and isinstance(v, float))
    row["TF_rows128"] = round(fl / b128 / 1e6, 1); row["TF_rows256"] = round(fl / b256 / 1e6, 1); row["TF_auto"] = round(fl / row["auto"] / 1e6, 1); row["rows256_over_rows128"] = round(b128 / b256, 3)
    print(json.dumps(row), flush=True)
