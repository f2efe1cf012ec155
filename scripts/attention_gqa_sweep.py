#!/usr/bin/env python3
"""The grouped-query decode attention step (32 query heads over 8 key / value heads, Llama-3-8B) over context lengths and key-range cuts
(tce_w4a16_set_debug_mode(3000 + workgroups)); caches rotate over enough sets to exceed the 256 MB memory-side cache.  Beside it the
multi-head kernel on the same number of query heads (4x the cache bytes)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from tune import dev, time_graph, capi
from tinychatengine_amd.attention_ops import DecodeAttention
L = capi.lib()
al = int(np.array([0.0884], np.float16).view(np.uint16)[0])
H, KV = 32, 8
qkv = torch.randn((H + 2 * KV) * 128, device=dev).half()
qkv_mha = torch.randn(3 * H * 128, device=dev).half()
oo = torch.empty(H, 128, dtype=torch.float16, device=dev)
for t in (128, 256, 512, 1024, 2048, 4096, 8192):
    bytes_ = 2 * KV * t * 128 * 2
    nsets = min(128, max(4, int(3.2e8 // bytes_) + 1))
    cos = torch.randn(t + 1, 128, device=dev).half(); sin = torch.randn(t + 1, 128, device=dev).half()
    atts = [DecodeAttention(H, 128, t, dev, cos, sin, kv_heads=KV) for _ in range(nsets)]
    for a_ in atts:
        a_.k_cache.normal_(0, 0.8); a_.v_cache.normal_(0, 0.8)
    def step(i, sp):
        a_ = atts[i % nsets]
        capi.check(L.tce_attention_decode_step_gqa_f16(qkv.data_ptr(), a_.k_cache.data_ptr(), a_.v_cache.data_ptr(), cos.data_ptr(), sin.data_ptr(), None, oo.data_ptr(),
                                                       a_.workspace.data_ptr(), H, KV, 128, t, t - 1, al, sp))
    row = {"context": t, "query_heads": H, "kv_heads": KV, "kv_MB": round(bytes_ / 1e6, 2), "cache_sets": nsets}
    for fuse in (1, 2, 4):
        capi.check(L.tce_w4a16_set_debug_mode(2920 + fuse))
        for wgs in (0, 64, 128, 256, 512):
            capi.check(L.tce_w4a16_set_debug_mode(3000 + wgs))
            row[f"fuse{fuse}_wgs{wgs}_us"] = round(time_graph(step, max(32, nsets)), 2)
    capi.check(L.tce_w4a16_set_debug_mode(2920))
    capi.check(L.tce_w4a16_set_debug_mode(3000))  # back to the fitted rule
    row["rule_us"] = round(time_graph(step, max(32, nsets)), 2)
    row["rule"] = capi.describe_attention_step(H, t, KV)
    del atts
    n2 = min(96, max(4, int(3.2e8 // (4 * bytes_)) + 1))
    atts = [DecodeAttention(H, 128, t, dev, cos, sin) for _ in range(n2)]
    for a_ in atts:
        a_.k_cache.normal_(0, 0.8); a_.v_cache.normal_(0, 0.8)
    def step2(i, sp):
        a_ = atts[i % n2]
        capi.check(L.tce_attention_decode_step_f16(qkv_mha.data_ptr(), a_.k_cache.data_ptr(), a_.v_cache.data_ptr(), cos.data_ptr(), sin.data_ptr(), None, oo.data_ptr(),
                                                   a_.workspace.data_ptr(), H, 128, t, t - 1, al, sp))
    row["multi_head_32_us"] = round(time_graph(step2, max(32, n2)), 2)
    print(json.dumps(row), flush=True)
    del atts
