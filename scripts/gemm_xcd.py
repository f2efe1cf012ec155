#!/usr/bin/env python3
"""Prefill GEMM: XCD grid shape (debug modes 41/42/44/48 = 1/2/4/8 XCD rows over the row blocks) x tile, us and TFLOP/s."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from tune import ring, dev, time_graph, capi
def main():
    L = capi.lib()
    for (M, N, K) in [(512, 4096, 4096), (512, 11008, 4096), (512, 4096, 11008), (2048, 4096, 4096)]:
        sets = ring(N, K, 128, min_bytes=3e8)
        x = torch.randn(M, K, device=dev).to(torch.float16); out = torch.empty(M, N, dtype=torch.float16, device=dev)
        ds = [capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(), zeros=s[2].data_ptr(), C=out.data_ptr()) for s in sets]
        for v in ((4, 1), (4, 2), (8, 1), (8, 2)):
            row = {"M": M, "N": N, "K": K, "tile": v}
            capi.set_gemm_config(*v)
            for xm in (1, 2, 4, 8):
                capi.check(L.tce_w4a16_set_debug_mode(40 + xm))
                us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 8 if M >= 2048 else 16)
                row[f"xm{xm}"] = [round(us, 1), round(2.0 * M * N * K / us / 1e6)]
            print(json.dumps(row), flush=True)
        capi.set_gemm_config(); L.tce_w4a16_set_debug_mode(41)
if __name__ == "__main__":
    main()
