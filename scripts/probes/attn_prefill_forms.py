#!/usr/bin/env python3
"""tce_attention_prefill_f16: the rule's choice against the forced forms (tce_w4a16_set_debug_mode: 2954 / 2958 = 4 / 8 waves x one row tile, 2964 / 2968 = x two row tiles; 27xx / 28xx:
heavy/light block pairing on / off) for prompts and chunks on a context; us per call (rotation + append + attention), 32 query heads x 128, 8 or 32 kv heads."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.attention_ops import DecodeAttention
dev = torch.device("cuda:0"); L = capi.lib()
H, hd = 32, 128
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3 / reps)
    return min(ts)
for KV in (8, 32):
    for (m, pos) in ((512, 0), (2048, 0), (130, 0), (512, 1536), (64, 2000), (256, 256)):
        maxk = pos + m
        cos = torch.randn(maxk, hd, device=dev).half(); sin = torch.randn(maxk, hd, device=dev).half()
        att = DecodeAttention(H, hd, maxk, dev, cos, sin, kv_heads=KV)
        att.k_cache.normal_(0, 0.8); att.v_cache.normal_(0, 0.8)
        qkv = torch.randn(m, (H + 2 * KV) * hd, device=dev).half()
        out = torch.empty(m, H * hd, dtype=torch.float16, device=dev)
        row = {"kv_heads": KV, "rows": m, "cached_keys": pos}
        for name, modes in (("rule", (2950,)), ("w4", (2954,)), ("w8", (2958,)), ("w4x2", (2964,)), ("w8x2", (2968,)), ("w4_paired", (2704,)), ("w8_paired", (2708,)), ("w4_unpaired", (2804,)), ("w8_unpaired", (2808,))):
            for m_ in modes: capi.check(L.tce_w4a16_set_debug_mode(m_))
            try:
                row[name] = round(timed(lambda: att.prefill(qkv, pos, out=out)), 1)
            except Exception as e:  # noqa: BLE001
                row[name] = None; L.tce_reset_last_error()
            capi.check(L.tce_w4a16_set_debug_mode(2950))
        best = min((v, k) for k, v in row.items() if isinstance(v, float) and k != "rule")
        row["best"] = best[1]; row["rule_over_best"] = round(row["rule"] / best[0], 3)
        print(json.dumps(row), flush=True)
        del att
