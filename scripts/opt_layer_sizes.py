#!/usr/bin/env python3
"""One decode token (m = 1, 512 keys) through whole SmoothQuant OPT decoder stacks at the three sizes the reference ships (llm/include/model.h: OPT-125M /
1.3B / 6.7B), every layer with its own int8 weights, the stack captured in one hipGraph: us per layer, launches, and the rate at which the layer's int8
weights are streamed (the figure a roofline prices: at 1.3B / 6.7B a layer's weights are 50 / 201 MB and the stack is far beyond the memory-side cache).

    python scripts/opt_layer_sizes.py [--eager]     (--eager: no graph, a few tokens, for rocprofv3 --kernel-trace --stats)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinychatengine_amd.opt_layer import Int8OPTDecoderLayer  # noqa: E402

dev = torch.device("cuda:0")
eager = "--eager" in sys.argv
only = [a for a in sys.argv[1:] if not a.startswith("--")]
for a in sys.argv[1:]:
    if a.startswith("--mode="):  # a debug mode of the library (timing experiments), e.g. --mode=86
        from tinychatengine_amd import capi
        capi.check(capi.lib().tce_w4a16_set_debug_mode(int(a.split("=")[1])))
SIZES = {"OPT-125M": (768, 12, 3072, 12), "OPT-1.3B": (2048, 32, 8192, 24), "OPT-6.7B": (4096, 32, 16384, 32)}
for name, (E, H, F, NL) in SIZES.items():
    if only and name not in only:
        continue
    m, pos = (512, 0) if "--prefill" in sys.argv else (1, 511)  # --prefill: a 512-row prompt instead of a decode token
    if m > 1:
        NL = min(NL, 8)
    tgz = pos + m
    layers = [Int8OPTDecoderLayer(E, H, F, 512, m, dev, seed=7 + i) for i in range(NL)]
    hid0 = torch.randn(m, E, device=dev)
    hid = hid0.clone()
    mask = torch.zeros((m, tgz), device=dev)
    if m > 1:
        mask.masked_fill_(torch.triu(torch.ones(m, tgz, dtype=torch.bool, device=dev), diagonal=pos + 1), torch.finfo(torch.float32).min)

    def token():
        hid.copy_(hid0)
        for l in layers:
            l.step(hid, pos, mask)
    token()
    torch.cuda.synchronize()
    if eager:
        for _ in range(5):
            token()
        torch.cuda.synchronize()
        print(json.dumps({"model": name, "eager_tokens": 6}), flush=True)
        del layers
        torch.cuda.empty_cache()
        continue
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        token()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    wbytes = 4 * E * E + 2 * E * F  # int8 weights of a layer (q, k, v, out_proj, fc1, fc2)
    kv = 2 * H * tgz * (E // H)     # the cache rows a token reads
    print(json.dumps({"model": name, "embed": E, "heads": H, "head_dim": E // H, "ffn": F, "layers": NL, "keys": tgz, "launches_per_layer": layers[0].launches(m),
                      "us_per_layer": round(us / NL, 2), "us_per_token": round(us, 1), "tokens_per_s_stack": round(1e6 / us, 1),
                      "weight_MB_per_layer": round(wbytes / 1e6, 2), "weight_stream_GBps": round((wbytes + kv) * NL / us / 1e3, 1),
                      "frac_of_8TBps": round((wbytes + kv) * NL / us / 1e3 / 8000, 3), "finite": bool(torch.isfinite(hid).all().item())}), flush=True)
    del g, layers
    torch.cuda.empty_cache()
