"""W4A16 at 17 <= M <= 384: the MFMA GEMM alone (small-batch kernel limited to M <= 16: debug mode 1016) against the dispatcher (debug mode 1128 = its default)."""
import ctypes as C, json, os, sys
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import torch
from tune import ring, dev, time_graph, capi
L = capi.lib()
for (N, K) in [(4096, 4096), (11008, 4096), (12288, 4096), (22016, 4096), (4096, 11008)]:
    sets = ring(N, K, 128, min_bytes=3e8)
    for M in (24, 32, 48, 64, 128, 192, 256, 384):
        x = torch.randn(M, K, device=dev).to(torch.float16); out = torch.empty(M, N, dtype=torch.float16, device=dev)
        ds = [capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(), zeros=s[2].data_ptr(), C=out.data_ptr(), flags=4) for s in sets]
        row = {"N": N, "K": K, "M": M}
        for name, mode in (("gemm", 1016), ("auto", 1128)):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            for _ in range(2):
                us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 16)
            row[name] = round(us, 1)
        print(json.dumps(row), flush=True)
capi.check(L.tce_w4a16_set_debug_mode(1128))
