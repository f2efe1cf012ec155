#!/usr/bin/env python3
"""torch.nn.functional.scaled_dot_product_attention for ONE query row per head (32 heads x 128) over 128 .. 8192 keys, KV sets rotating
through more than 256 MB, one hipGraph -- the framework's attention next to tce_attention_decode_step_f16 (which also rotates q and
the new key and appends them).  A yardstick, not a product path.
   gpurun -- 'python scripts/sdpa_decode_yardstick.py > gpurun_out/sdpa_decode_yardstick.jsonl'"""
import json
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
H, D = 32, 128
for keys in (128, 512, 2048, 8192):
    bytes_ = 2 * H * keys * D * 2
    nsets = min(96, max(4, int(3.2e8 // bytes_) + 1))
    Ks = [torch.randn(1, H, keys, D, device=dev, dtype=torch.float16) for _ in range(nsets)]
    Vs = [torch.randn(1, H, keys, D, device=dev, dtype=torch.float16) for _ in range(nsets)]
    q = torch.randn(1, H, 1, D, device=dev, dtype=torch.float16)
    outs = [None] * nsets

    def token():
        for i in range(nsets):
            outs[i] = F.scaled_dot_product_attention(q, Ks[i], Vs[i])
    row = {"keys": keys, "kv_MB": round(bytes_ / 1e6, 1), "sets": nsets}
    try:
        for _ in range(3):
            token()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            token()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000.0 / (5 * nsets))
        row["sdpa_us"] = round(best, 2)
    except Exception as e:  # noqa: BLE001
        row["error"] = f"{type(e).__name__}: {e}"[:200]
    print(json.dumps(row), flush=True)
    del Ks, Vs
