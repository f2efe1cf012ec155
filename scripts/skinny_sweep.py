import ctypes as C, json, os, sys
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import torch
from tune import ring, dev, time_graph, capi
L = capi.lib()
for (N, K) in [(4096, 4096), (22016, 4096), (4096, 11008), (12288, 4096), (128256, 4096)]:
    sets = ring(N, K, 128, min_bytes=1.0e9)
    for M in (4, 16):
        x = torch.randn(M, K, device=dev).to(torch.float16); out = torch.empty(M, N, dtype=torch.float16, device=dev)
        ds = [capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(), zeros=s[2].data_ptr(), C=out.data_ptr(), flags=4) for s in sets]
        row = {"N": N, "K": K, "M": M}
        for mode in (20, 21, 22, 24, 28, 30):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            row["shared" if mode == 30 else f"ks{mode-20}"] = round(time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 32), 2)
        capi.check(L.tce_w4a16_set_debug_mode(20))
        print(json.dumps(row), flush=True)
