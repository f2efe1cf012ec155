#!/usr/bin/env python3
"""tce_w8a8_matmul at decode-sized M: the wave-per-column kernels (M <= 8) against the MFMA tiles (tce_w4a16_set_debug_mode(73): the rule without them), weights rotating through HBM."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tune import dev, time_graph, capi
L = capi.lib()
g = torch.Generator(device=dev).manual_seed(3)
ri = lambda *s: torch.randint(-127, 128, s, device=dev, generator=g, dtype=torch.int32).to(torch.int8)
for (N, K) in ((768, 768), (3072, 768), (768, 3072), (2048, 2048), (8192, 2048), (2048, 8192), (4096, 4096), (16384, 4096), (4096, 16384)):
    nsets = max(2, min(48, int(3e8 // (N * K))))
    Ws = [(ri(N, K), ri(N)) for _ in range(nsets)]
    for M in (1, 2, 4, 5, 8, 9, 12, 16):
        A = ri(M, K); o = torch.empty(M, N, dtype=torch.int8, device=dev)
        ds = [capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=A.data_ptr(), B=W.data_ptr(), bias=b.data_ptr(), C=o.data_ptr(), alpha=0.0005, beta=0.02, q_min=-128, q_max=127,
                            bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8) for (W, b) in Ws]
        row = {"M": M, "N": N, "K": K}
        for name, mode in (("rule", 70), ("mfma_only", 73)):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            row[name] = round(min(time_graph(lambda i, sp: capi.check(L.tce_w8a8_matmul(C.byref(ds[i % nsets]), sp)), 32) for _ in range(2)), 2)
        capi.check(L.tce_w4a16_set_debug_mode(70))
        print(json.dumps(row), flush=True)
    del Ws
    torch.cuda.empty_cache()
