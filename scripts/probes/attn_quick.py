#!/usr/bin/env python3
"""The grouped-query decode attention step at a few contexts, the fitted rule only (a quick A/B while changing attention_fast.hip)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tune import dev, time_graph, capi
from tinychatengine_amd.attention_ops import DecodeAttention
L = capi.lib()
al = int(np.array([0.0884], np.float16).view(np.uint16)[0])
H, KV = 32, int(os.environ.get("ATTN_KV_HEADS", "8"))
qkv = torch.randn((H + 2 * KV) * 128, device=dev).half()
oo = torch.empty(H, 128, dtype=torch.float16, device=dev)
for mode in os.environ.get("ATTN_MODES", "0").split(","):  # e.g. "0,2916+3032": debug modes applied together (2900 + waves, 3000 + workgroups)
    for t in [int(x) for x in os.environ.get("ATTN_CONTEXTS", "128,256,512,1024,2048,4096").split(",")]:
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
        posw = torch.tensor([t - 1], dtype=torch.int32, device=dev)
        def step_dev(i, sp):  # the position in a device word (what a replayable token graph uses)
            a_ = atts[i % nsets]
            capi.check(L.tce_attention_decode_step_pos_f16(qkv.data_ptr(), a_.k_cache.data_ptr(), a_.v_cache.data_ptr(), cos.data_ptr(), sin.data_ptr(), None, oo.data_ptr(),
                                                           a_.workspace.data_ptr(), H, KV, 128, t, posw.data_ptr(), t - 1, al, sp))
        for m_ in mode.split("+"):
            if int(m_): capi.check(L.tce_w4a16_set_debug_mode(int(m_)))
        us = [round(time_graph(step, max(32, nsets)), 2) for _ in range(3)]
        us_dev = [round(time_graph(step_dev, max(32, nsets)), 2) for _ in range(3)]
        print(json.dumps({"mode": mode, "context": t, "us": us, "us_pos_on_device": us_dev, "rule": capi.describe_attention_step(H, t, KV)}), flush=True)
        del atts
