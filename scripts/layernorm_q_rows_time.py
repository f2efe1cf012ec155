#!/usr/bin/env python3
"""tce_layernorm_q alone on a few long rows (the decode token's row at the three OPT widths, and small batches): us per launch in a hipGraph of 32."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tune import dev, time_graph, capi
L = capi.lib()
for m, n in ((1, 768), (1, 2048), (1, 4096), (8, 4096), (64, 4096), (256, 768), (256, 4096), (108, 768), (512, 768), (512, 2048), (512, 4096), (1024, 4096), (2048, 4096), (2048, 2048), (4096, 4096)):
    x = torch.randn(m, n, device=dev); w = torch.randn(n, device=dev); b = torch.randn(n, device=dev)
    out = torch.empty(m, n, dtype=torch.int8, device=dev)
    fn = lambda i, sp: capi.check(L.tce_layernorm_q(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), m, n, sp))
    print(json.dumps({"rows": m, "columns": n, "us": round(min(time_graph(fn, 32) for _ in range(3)), 2)}), flush=True)
