#!/usr/bin/env python3
"""In-process A/B of the pre-packed GEMM's accumulator rescale: two v_pk_mul_f32 per accumulator tile (the kernel as shipped; forced to
the one-quartet 128x128 form, debug mode 61) against four v_mul_f32 (this round's first form, kept as ablation bit 64: mode 664).
Boxes differ by 4-8 % on MFMA work, so only numbers taken in one process compare.
   gpurun -- 'python scripts/gemm_pk_rescale_ab.py > gpurun_out/gemm_pk_rescale_ab.jsonl'"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinychatengine_amd import lab; lab.use_lab()  # (the scalar-rescale instantiation: the diagnostics build)
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4

dev = torch.device("cuda:0")
L = capi.lib()


def timed(fn, reps=20, warm=6):
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
    return round(min(ts), 2)


g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K) in [(2048, 4096, 4096), (2048, 11008, 4096), (2048, 4096, 11008), (512, 14336, 4096), (512, 11008, 4096), (4096, 4096, 4096)]:
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    i = [0]

    def run():
        i[0] += 1
        lins[i[0] % 3].forward(x, out)
    row = {"M": M, "N": N, "K": K, "packed_us": [], "scalar_us": []}
    for rep in range(3):
        for mode, key in ((61, "packed_us"), (664, "scalar_us")):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            row[key].append(timed(run))
    capi.check(L.tce_w4a16_set_debug_mode(600))
    capi.check(L.tce_w4a16_set_debug_mode(60))
    fl = 2.0 * M * N * K
    row["packed_TF"] = round(fl / min(row["packed_us"]) / 1e6, 1)
    row["scalar_TF"] = round(fl / min(row["scalar_us"]) / 1e6, 1)
    print(json.dumps(row), flush=True)
    del lins
