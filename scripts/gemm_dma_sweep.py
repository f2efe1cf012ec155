#!/usr/bin/env python3
"""Prefill GEMM: the older register-staged kernel (tile ids m,n) against the LDS-DMA kernel (ids 200 + m, n) and the dispatcher's own choice;
us and TFLOP/s per launch (hipGraph of launches cycling a > 300 MB ring of weight sets, so the weights come from HBM)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from tune import ring, dev, time_graph, capi


def main():
    L = capi.lib()
    for (M, N, K) in [(512, 4096, 4096), (512, 11008, 4096), (512, 4096, 11008), (128, 4096, 4096), (2048, 4096, 4096), (4096, 4096, 4096), (4096, 11008, 4096)]:
        sets = ring(N, K, 128, min_bytes=3e8)
        x = torch.randn(M, K, device=dev).to(torch.float16); out = torch.empty(M, N, dtype=torch.float16, device=dev)
        ds = [capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(), zeros=s[2].data_ptr(), C=out.data_ptr()) for s in sets]
        row = {"M": M, "N": N, "K": K}
        for _ in range(3):  # burn-in: the first timings after allocating a ring run 10 % slow (clock ramp, first-touch mappings)
            time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 16)
        for v in (None, (4, 1), (4, 2), (8, 2), (204, 1), (204, 2), (208, 2), (204, 4)):
            capi.set_gemm_config(*(v or (0, 0)))
            us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 8 if M >= 2048 else 16)
            row[str(v or "auto")] = [round(us, 1), round(2.0 * M * N * K / us / 1e6)]
        capi.set_gemm_config()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
