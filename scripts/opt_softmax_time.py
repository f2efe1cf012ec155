#!/usr/bin/env python3
"""tce_opt_softmax_q and tce_layernorm_q alone at prompt sizes; us per launch (hipGraph of 16)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tune import dev, time_graph, capi
L = capi.lib()
for heads, sq, tgz in ((12, 512, 512), (32, 512, 512), (12, 108, 108), (12, 1, 512), (32, 2048, 2048)):
    s = torch.randn(heads, sq, tgz, device=dev) * 2
    mask = torch.zeros(sq, tgz, device=dev).masked_fill_(torch.triu(torch.ones(sq, tgz, dtype=torch.bool, device=dev), diagonal=tgz - sq + 1), torch.finfo(torch.float32).min)
    ldp = (tgz + 15) // 16 * 16
    out = torch.empty(heads, sq, ldp, dtype=torch.int8, device=dev)
    fn = lambda i, sp: capi.check(L.tce_opt_softmax_q(C.c_void_p(s.data_ptr()), C.c_void_p(mask.data_ptr()), C.c_void_p(out.data_ptr()), heads, sq, tgz, ldp, sp))
    print(json.dumps({"softmax heads": heads, "rows_per_head": sq, "keys": tgz, "us": round(min(time_graph(fn, 16) for _ in range(3)), 2)}), flush=True)
