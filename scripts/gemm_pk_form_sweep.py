#!/usr/bin/env python3
"""Prefill GEMM on the packed copy: for every (M, N, K) the dispatcher's choice against every forced form -- whole tiles with one / two quartets, 128 x 256 tiles,
the k range cut in 2 / 3 / 4 runs, the 64-row LDS-DMA kernel -- to check the cost models (tce_w4a16_set_debug_mode 60..64, 640..644, 69).  us per launch, weights
rotating over three copies.  usage: gemm_pk_form_sweep.py [MxNxK ...]"""
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
    shapes = [(M, N, K) for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (6144, 4096), (14336, 4096), (4096, 14336)) for M in (192, 256, 384, 512, 768, 1024, 2048)]
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
    for name, modes in (("auto", (60,)), ("whole_1q", (61,)), ("whole_2q", (62,)), ("tile_128x256", (63,)), ("cut2", (642,)), ("cut3", (643,)), ("cut4", (644,)), ("dma_64row", (69,))):
        for m_ in modes: L.tce_w4a16_set_debug_mode(m_)
        try:
            row[name] = round(timed(run), 2)
        except Exception as e:  # noqa: BLE001
            row[name] = None; L.tce_reset_last_error()
        if name == "auto": row["auto_is"] = " ".join(capi.describe_dispatch(descs[0]).split()[:4])
        L.tce_w4a16_set_debug_mode(60)
    best = min((v, k) for k, v in row.items() if isinstance(v, float) and k != "auto")
    row["best"] = best[1]; row["auto_over_best"] = round(row["auto"] / best[0], 3)
    print(json.dumps(row), flush=True)
