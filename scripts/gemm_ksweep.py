"""Prefill GEMM at M = 512, N = 4096 against K: the slope is the cost of a k-block, the intercept the launch + prologue + epilogue."""
import ctypes as C, json, os, sys
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import torch
from tune import ring, dev, time_graph, capi
L = capi.lib()
M, N = 512, 4096
for K in (512, 1024, 2048, 4096, 8192):
    sets = ring(N, K, 128, min_bytes=3e8)
    x = torch.randn(M, K, device=dev).to(torch.float16); out = torch.empty(M, N, dtype=torch.float16, device=dev)
    ds = [capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(), zeros=s[2].data_ptr(), C=out.data_ptr()) for s in sets]
    row = {"M": M, "N": N, "K": K}
    for v in ((4, 2), (204, 2), (204, 1)):
        capi.set_gemm_config(*v)
        us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 16)
        row[f"{v}"] = round(us, 1)
    print(json.dumps(row), flush=True)
capi.set_gemm_config()
