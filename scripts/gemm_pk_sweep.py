"""Times the pre-packed 128-row GEMM (both forms) against the LDS-DMA GEMM on the prefill shapes.
   gpurun -- 'python scripts/gemm_pk_sweep.py > gpurun_out/gemm_pk_sweep.jsonl'"""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4

dev = torch.device("cuda:0")
L = capi.lib()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=30, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000.0 / reps)
    return min(ts), sorted(ts)[1]


shapes = [(4096, 4096), (11008, 4096), (4096, 11008), (14336, 4096)]
Ms = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096]
g = torch.Generator(device=dev).manual_seed(1)
for (N, K) in shapes:
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]  # rotate weights
    for M in Ms:
        x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        descs = [l.desc(x, out) for l in lins]
        it = [0]

        def run():
            capi.check(capi.w4a16_forward(descs[it[0] % 3], st))
            it[0] += 1

        row = {"M": M, "N": N, "K": K}
        for name, mode in (("dma", 69), ("pk1", 61), ("pk2", 62), ("pk3", 63), ("pk4", 64), ("auto", 60)):
            L.tce_w4a16_set_debug_mode(mode)
            us, med = timed(run)
            row[name + "_us"] = round(us, 2)
            row[name + "_TF"] = round(2.0 * M * N * K / us / 1e6, 1)
            if name in ("auto", "pk4"):
                row[name + "_dispatch"] = capi.describe_dispatch(descs[0])
        L.tce_w4a16_set_debug_mode(60)
        print(json.dumps(row), flush=True)
