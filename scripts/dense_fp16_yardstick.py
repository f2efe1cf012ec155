#!/usr/bin/env python3
"""What the vendor's dense fp16 GEMM (hipBLASLt through torch.matmul, fp16 weights, no dequantization) does on the prefill shapes the
W4A16 GEMM is measured on, same harness (three weight copies in rotation, best of three batches).  A yardstick for the M = 512 / 2048
bars, not a product path: the W4A16 kernels read a quarter of the weight bytes and dequantize on the way.
   gpurun -- 'python scripts/dense_fp16_yardstick.py > gpurun_out/dense_fp16_yardstick.jsonl'"""
import json
import torch

dev = torch.device("cuda:0")


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
    return min(ts)


for (N, K) in [(4096, 4096), (11008, 4096), (4096, 11008), (14336, 4096)]:
    ws = [torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02 for _ in range(3)]
    for M in (512, 2048, 4096):
        x = torch.randn(M, K, device=dev, dtype=torch.float16)
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        i = [0]

        def run():
            i[0] += 1
            torch.matmul(x, ws[i[0] % 3].t(), out=out)
        us = timed(run)
        print(json.dumps({"M": M, "N": N, "K": K, "dense_fp16_us": round(us, 2), "dense_fp16_TF": round(2.0 * M * N * K / us / 1e6, 1)}), flush=True)
    del ws
