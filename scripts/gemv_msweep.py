#!/usr/bin/env python3
"""Decode GEMV at M = 1..16 (small batches / speculative decoding): time per launch for fixed weights."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from tune import ring, dev, time_graph, capi
def main():
    L = capi.lib()
    for (N, K) in [(4096, 4096), (22016, 4096), (4096, 11008)]:
        sets = ring(N, K, 128, min_bytes=1.0e9)
        row = {"N": N, "K": K}
        for M in (1, 2, 3, 4, 5, 8, 9, 12, 16):
            x = torch.randn(M, K, device=dev).to(torch.float16); out = torch.empty(M, N, dtype=torch.float16, device=dev)
            ds = [capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(), zeros=s[2].data_ptr(), C=out.data_ptr()) for s in sets]
            us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 64)
            row[f"M={M}"] = round(us, 2)
            if 2 <= M <= 16:  # the paths the small-batch kernel replaced (GEMV with repeated MFMAs up to 8, GEMM above)
                capi.check(L.tce_w4a16_set_debug_mode(29))
                row[f"M={M} old"] = round(time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 64), 2)
                capi.check(L.tce_w4a16_set_debug_mode(20))
        print(json.dumps(row), flush=True)
if __name__ == "__main__":
    main()
