#!/usr/bin/env python3
"""Prefill GEMM: TFLOP/s against M for a fixed 4096x4096 weight (does the rate rise with more workgroups per CU?)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from tune import ring, dev, time_graph, capi
def main():
    L = capi.lib()
    N, K = 4096, 4096
    sets = ring(N, K, 128, min_bytes=3e8)
    for M in (64, 512, 2048, 8192):
        x = torch.randn(M, K, device=dev).to(torch.float16); out = torch.empty(M, N, dtype=torch.float16, device=dev)
        ds = [capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(), zeros=s[2].data_ptr(), C=out.data_ptr()) for s in sets]
        row = {"M": M, "N": N, "K": K}
        for v in (None, (4, 1), (4, 2), (8, 1), (8, 2), (2, 2), (104, 1), (104, 2)):
            capi.set_gemm_config(*(v or (0, 0)))
            us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 8 if M >= 2048 else 16)
            row[str(v or "auto")] = [round(us, 1), round(2.0 * M * N * K / us / 1e6)]
        capi.set_gemm_config()
        print(json.dumps(row), flush=True)
if __name__ == "__main__":
    main()
