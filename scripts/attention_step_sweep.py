#!/usr/bin/env python3
"""The fast decode attention step over context lengths and key-range cuts (tce_w4a16_set_debug_mode(3000 + workgroups)); the KV caches
of the graph's nodes rotate over enough sets to exceed the 256 MB memory-side cache, so the keys come from HBM as they do in a
model with 32 layers of cache."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from tune import dev, time_graph, capi
from tinychatengine_amd.attention_ops import DecodeAttention
L = capi.lib()
al = int(np.array([0.0884], np.float16).view(np.uint16)[0])
qkv = torch.randn(3 * 32 * 128, device=dev).half()
oo = torch.empty(32, 128, dtype=torch.float16, device=dev)
for t in (128, 256, 512, 1024, 2048, 4096, 8192):
    bytes_ = 2 * 32 * t * 128 * 2
    nsets = min(96, max(4, int(3.2e8 // bytes_) + 1))
    cos = torch.randn(t + 1, 128, device=dev).half(); sin = torch.randn(t + 1, 128, device=dev).half()
    atts = [DecodeAttention(32, 128, t, dev, cos, sin) for _ in range(nsets)]
    for a_ in atts:
        a_.k_cache.normal_(0, 0.8); a_.v_cache.normal_(0, 0.8)
    def step(i, sp):
        a_ = atts[i % nsets]
        capi.check(L.tce_attention_decode_step_f16(qkv.data_ptr(), a_.k_cache.data_ptr(), a_.v_cache.data_ptr(), cos.data_ptr(), sin.data_ptr(), None, oo.data_ptr(),
                                                   a_.workspace.data_ptr(), 32, 128, t, t - 1, al, sp))
    row = {"context": t, "kv_MB": round(bytes_ / 1e6, 1), "cache_sets": nsets}
    for nw in (4, 8, 16):
        capi.check(L.tce_w4a16_set_debug_mode(2900 + nw))
        for wgs in (32, 64, 128, 256, 512):
            capi.check(L.tce_w4a16_set_debug_mode(3000 + wgs))
            row[f"wgs{wgs}_waves{nw}_us"] = round(time_graph(step, max(32, nsets)), 2)
    capi.check(L.tce_w4a16_set_debug_mode(2900))
    capi.check(L.tce_w4a16_set_debug_mode(3000))  # back to the fitted rule
    row["rule_us"] = round(time_graph(step, max(32, nsets)), 2)
    print(json.dumps(row), flush=True)
    del atts
