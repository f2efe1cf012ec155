#!/usr/bin/env python3
"""tce_w8a8_matmul on the OPT-125M / 1.3B linears (BASELINE config 4 and neighbours) at 512 / 108 / 16 rows: us per launch (graphs of 32 launches over
rotating weight sets).  For same-session A/Bs of two builds (TCE_LIB_PATH)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tune import dev, time_graph, capi
L = capi.lib()
g = torch.Generator(device=dev).manual_seed(3)
ri = lambda *s: torch.randint(-127, 128, s, device=dev, generator=g, dtype=torch.int32).to(torch.int8)
rows = []
for m_ in os.environ.get("W8A8_MODES", "").split(","):
    if m_: capi.check(L.tce_w4a16_set_debug_mode(int(m_)))  # e.g. 171 / 172 / 174: the deep-pipeline 64 x 64 kernel forced with 1 / 2 / 4 quartets; 179 off
shapes = [tuple(int(v) for v in x.split("x")) for x in os.environ["W8A8_SHAPES"].split(",")] if os.environ.get("W8A8_SHAPES") else None
default = [(M, N, K) for M in (512, 108, 16) for N, K in ((768, 768), (3072, 768), (768, 3072), (2048, 2048), (8192, 2048), (2048, 8192))]
for M, N, K in (shapes or default):
    if True:
        nsets = max(2, min(64, int(3e8 // (N * K))))
        A = ri(M, K)
        sets = []
        for _ in range(nsets):
            if os.environ.get("W8A8_OUT", "int8") == "fp32":  # the fp32-output form with an fp32 bias accumulating into C (out_proj / fc2 of the OPT layers)
                W, b, o = ri(N, K), torch.randn(N, device=dev, generator=g), torch.zeros(M, N, dtype=torch.float32, device=dev)
                d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=A.data_ptr(), B=W.data_ptr(), bias=b.data_ptr(), C=o.data_ptr(), alpha=0.0005, q_min=-128, q_max=127,
                                  bias_kind=capi.TCE_BIAS_FP32, out_kind=capi.TCE_OUT_FP32, accumulate=1)
            else:
                W, b, o = ri(N, K), ri(N), torch.empty(M, N, dtype=torch.int8, device=dev)
                d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=A.data_ptr(), B=W.data_ptr(), bias=b.data_ptr(), C=o.data_ptr(), alpha=0.0005, beta=0.02, q_min=-128, q_max=127,
                                  bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8)
            sets.append((d, W, b, o))
        us = min(time_graph(lambda i, sp: capi.check(L.tce_w8a8_matmul(C.byref(sets[i % nsets][0]), sp)), 32) for _ in range(3))
        rows.append(f"{M}x{N}x{K}: {us:.2f}")
        del sets
print(json.dumps({"lib": os.environ.get("TCE_LIB_PATH", "in-tree"), "modes": os.environ.get("W8A8_MODES", ""), "out": os.environ.get("W8A8_OUT", "int8"), "us": rows}), flush=True)
