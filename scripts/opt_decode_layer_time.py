#!/usr/bin/env python3
"""OPT-125M decode (M = 1): the LayerNormQ + q/k/v and LayerNormQ + fc1 steps as separate launches vs the fused launch
(tce_layernorm_q_w8a8_group).  us per step, hipGraph of 32 steps."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tune import dev, time_graph, capi
from tinychatengine_amd.linear import W8A8B8O8Linear
L = capi.lib()
E, F = 768, 3072
g = torch.Generator(device=dev).manual_seed(3)
ri = lambda *s: torch.randint(-127, 128, s, device=dev, generator=g, dtype=torch.int32).to(torch.int8)
x = torch.randn(1, E, device=dev); lw = torch.randn(E, device=dev); lb = torch.randn(E, device=dev)
for name, ns in (("ln + q,k,v (3 x 768x768)", (E, E, E)), ("ln + fc1 (3072x768, ReLU)", (F,))):
    lins = [W8A8B8O8Linear(ri(n, E), ri(n), 0.0005, 0.02, relu=len(ns) == 1) for n in ns]
    q8 = torch.empty(1, E, dtype=torch.int8, device=dev)
    outs = [torch.empty(1, n, dtype=torch.int8, device=dev) for n in ns]
    descs = [capi.W8A8Desc(M=1, N=n, K=E, batch=1, A=q8.data_ptr(), B=l.weight.data_ptr(), bias=l.bias.data_ptr(), C=o.data_ptr(), alpha=l.alpha, beta=l.beta,
                           q_min=l.q_min, q_max=127, bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8) for l, n, o in zip(lins, ns, outs)]
    arr = (capi.W8A8Desc * len(descs))(*descs)

    def separate(i, sp):
        capi.check(L.tce_layernorm_q(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), q8.data_ptr(), 1, E, sp))
        for d in descs:
            capi.check(L.tce_w8a8_matmul(C.byref(d), sp))

    def fused(i, sp):
        capi.check(L.tce_layernorm_q_w8a8_group(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1, E, arr, len(descs), None, sp))

    print(json.dumps({"step": name, "launches_separate": 1 + len(ns), "separate_us": round(time_graph(separate, 32), 2), "fused_us": round(time_graph(fused, 32), 2)}), flush=True)
