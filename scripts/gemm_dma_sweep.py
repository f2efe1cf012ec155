"""Prefill GEMM: classic (4,2) against the LDS-DMA kernel tiles (ids 200 + m_tiles), XCD grid rows 1 and 8; us and TFLOP/s."""
import ctypes as C, json, os, sys
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import torch
from tune import ring, dev, time_graph, capi
L = capi.lib()
for (M, N, K) in [(512, 4096, 4096), (512, 11008, 4096), (512, 4096, 11008), (2048, 4096, 4096), (4096, 4096, 4096)]:
    sets = ring(N, K, 128, min_bytes=3e8)
    x = torch.randn(M, K, device=dev).to(torch.float16); out = torch.empty(M, N, dtype=torch.float16, device=dev)
    ds = [capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(), zeros=s[2].data_ptr(), C=out.data_ptr()) for s in sets]
    row = {"M": M, "N": N, "K": K}
    for v in ((4, 2), (204, 1), (204, 2), (208, 1), (208, 2), (204, 4), (202, 2)):
        capi.set_gemm_config(*v)
        for xm in (1, 8):
            capi.check(L.tce_w4a16_set_debug_mode(40 + xm))
            us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 8 if M >= 2048 else 16)
            row[f"{v} xm{xm}"] = [round(us, 1), round(2.0 * M * N * K / us / 1e6)]
    print(json.dumps(row), flush=True)
capi.set_gemm_config()
