#!/usr/bin/env python3
"""tce_w8a8_matmul (W8A8B8O8Linear: int8 in, int8 out, int8 bias) on prefill-sized problems at the three OPT widths: us, TOP/s (dense int8 peak ~5000)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tune import dev, time_graph, capi
L = capi.lib()
g = torch.Generator(device=dev).manual_seed(3)
ri = lambda *s: torch.randint(-127, 128, s, device=dev, generator=g, dtype=torch.int32).to(torch.int8)
modes = [int(a) for a in sys.argv[1:]] or [75]  # tce_w4a16_set_debug_mode: 75 the rule, 76 / 77 the 128-row tiles forced with 128 / 64 columns, 78 off
for M in (512, 2048):
    for N, K in ((768, 768), (3072, 768), (768, 3072), (2048, 2048), (8192, 2048), (2048, 8192), (4096, 4096), (16384, 4096), (4096, 16384)):
        nsets = max(2, int(3e8 // (N * K)))
        A = ri(M, K)
        sets = []
        for _ in range(nsets):
            W, b, o = ri(N, K), ri(N), torch.empty(M, N, dtype=torch.int8, device=dev)
            d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=A.data_ptr(), B=W.data_ptr(), bias=b.data_ptr(), C=o.data_ptr(), alpha=0.0005, beta=0.02, q_min=-128, q_max=127,
                              bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8)
            sets.append((d, W, b, o))
        row = {"M": M, "N": N, "K": K}
        for mode in modes:
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            us = min(time_graph(lambda i, sp: capi.check(L.tce_w8a8_matmul(C.byref(sets[i % nsets][0]), sp)), 32) for _ in range(3))
            row[f"mode{mode}_us"] = round(us, 2)
            row[f"mode{mode}_TOPs"] = round(2.0 * M * N * K / us / 1e6, 1)
        capi.check(L.tce_w4a16_set_debug_mode(75))
        print(json.dumps(row), flush=True)
        del sets
        torch.cuda.empty_cache()
