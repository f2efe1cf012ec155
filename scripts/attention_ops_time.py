#!/usr/bin/env python3
"""Decode-time attention operators (32 heads, head_dim 128) at a few context lengths: us per call, hipGraph of 32 calls."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from tune import dev, time_graph, capi
L = capi.lib()
for t in (128, 512, 2048):
    q = torch.randn(32, 1, 128, device=dev).half(); k = torch.randn(32, t, 128, device=dev).half()
    s = torch.empty(32, 1, t, dtype=torch.float16, device=dev); p = torch.empty_like(s)
    vt = torch.randn(32, 128, t, device=dev).half(); o = torch.empty(32, 1, 128, dtype=torch.float16, device=dev)
    one = int(np.array([1.0], np.float16).view(np.uint16)[0]); al = int(np.array([0.0884], np.float16).view(np.uint16)[0])
    row = {"context": t}
    row["qk_us"] = round(time_graph(lambda i, sp: capi.check(L.tce_bmm_f16t(q.data_ptr(), k.data_ptr(), s.data_ptr(), 32, 1, t, 128, al, sp)), 32), 2)
    row["softmax_us"] = round(time_graph(lambda i, sp: capi.check(L.tce_softmax_half(s.data_ptr(), p.data_ptr(), 32, t, sp)), 32), 2)
    row["pv_us"] = round(time_graph(lambda i, sp: capi.check(L.tce_bmm_f16t(p.data_ptr(), vt.data_ptr(), o.data_ptr(), 32, 1, 128, t, one, sp)), 32), 2)
    print(json.dumps(row), flush=True)
    row2 = {"context": t}
    kk = torch.randn(32, t, 128, device=dev).half(); qq = torch.randn(32, 128, device=dev).half(); oo = torch.empty(32, 128, dtype=torch.float16, device=dev)
    row2["fused_decode_us"] = round(time_graph(lambda i, sp: capi.check(L.tce_attention_decode_f16(qq.data_ptr(), kk.data_ptr(), vt.data_ptr(), None, oo.data_ptr(), 32, t, 128, al, sp)), 32), 2)
    print(json.dumps(row2), flush=True)
    # the one-launch fp32 step (RoPE + append + chunked online softmax): a fresh set of caches per graph node would not fit for long
    # contexts, so the 32 nodes rotate over 4 cache sets (4 x 2 x 32 x t x 128 halves: > the 32 MB of L2 from t = 512 up)
    from tinychatengine_amd.attention_ops import DecodeAttention
    cos = torch.randn(t + 1, 128, device=dev).half(); sin = torch.randn(t + 1, 128, device=dev).half()
    atts = [DecodeAttention(32, 128, t, dev, cos, sin) for _ in range(4)]
    for a_ in atts:
        a_.k_cache.normal_(0, 0.8); a_.v_cache.normal_(0, 0.8)
    qkv = torch.randn(3 * 32 * 128, device=dev).half()
    def step(i, sp):
        a_ = atts[i % 4]
        capi.check(L.tce_attention_decode_step_f16(qkv.data_ptr(), a_.k_cache.data_ptr(), a_.v_cache.data_ptr(), cos.data_ptr(), sin.data_ptr(), None, oo.data_ptr(),
                                                   a_.workspace.data_ptr(), 32, 128, t, t - 1, al, sp))
    us = time_graph(step, 32)
    bytes_ = 2 * 32 * t * 128 * 2
    print(json.dumps({"context": t, "decode_step_fp32_us": round(us, 2), "kv_bytes": bytes_, "GBs": round(bytes_ / us / 1e3, 1)}), flush=True)
