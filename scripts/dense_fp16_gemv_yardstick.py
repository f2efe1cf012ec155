#!/usr/bin/env python3
"""The vendor's dense fp16 GEMV (rocBLAS / hipBLASLt through torch) streaming the SAME NUMBER OF BYTES per launch as the W4A16 decode launches,
issued the way bench.py's roofline leg issues them: one hipGraph of stream-ordered launches rotating over 32 different matrices
(more than the 256 MB memory-side cache).  A yardstick for "fraction of 8 TB/s per launch", not a product path.
   gpurun -- 'python scripts/dense_fp16_gemv_yardstick.py > gpurun_out/dense_fp16_gemv_yardstick.jsonl'"""
import json
import torch

dev = torch.device("cuda:0")
K = 4096
# rows of an fp16 [rows x 4096] matrix with the bytes of: o_proj (8.67 MB), qkv (26.0 MB), gate+up (46.6 MB), lm_head 32000 (67.7 MB)
for name, nbytes in (("o_proj-sized", 8667136), ("qkv-sized", 25985024), ("gate+up-sized", 46558208), ("lm_head-sized", 67656192)):
    rows = nbytes // (K * 2) // 16 * 16
    n_mats = 32
    ws = [torch.randn(rows, K, device=dev, dtype=torch.float16) * 0.02 for _ in range(n_mats)]
    x = torch.randn(K, device=dev, dtype=torch.float16)
    outs = [torch.empty(rows, device=dev, dtype=torch.float16) for _ in range(n_mats)]
    res = {"launch": name, "rows": rows, "bytes": rows * K * 2}
    for api in ("mv", "matmul_row"):
        def token():
            for w, o in zip(ws, outs):
                if api == "mv":
                    torch.mv(w, x, out=o)
                else:
                    torch.matmul(x.view(1, K), w.t(), out=o.view(1, rows))
        for _ in range(3):
            token()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            token()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000.0 / (10 * n_mats))
        res[api + "_us"] = round(best, 2)
        res[api + "_frac_of_8TBs"] = round(rows * K * 2 / best / 1e6 / 8.0, 3)
        del g
    print(json.dumps(res), flush=True)
    del ws, outs
