"""tce_attention_prefill_f16 over prompt lengths, against (i) torch's scaled_dot_product_attention on the same q / K / V (no rotation, no append: the
vendor's fused attention as a yardstick) and (ii) the reference-arithmetic operators of the same block (tce_bmm_f16t -> tce_softmax_half -> tce_bmm_f16t:
binary16 chains, what the reference's CUDA kernels compute), one jsonl line per case.

    python scripts/attention_prefill_sweep.py [--out profiles/r3/attention_prefill_sweep.jsonl]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinychatengine_amd.attention_ops import BMM_F16T, DecodeAttention, softmax  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/attention_prefill_sweep.jsonl")
    ap.add_argument("--chain", action="store_true", help="also time the reference-arithmetic operators (slow)")
    ap.add_argument("--waves", type=int, default=0, help="force 4 / 8 waves per workgroup (0: the launch's rule; 14 / 18: two row tiles per wave)")
    ap.add_argument("--pair", type=int, default=0, help="heavy / light block pairing: 0 the launch's rule, 1 forced on, 2 forced off (with --waves 0 / 4 / 8)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hd = 128
    from tinychatengine_amd import capi
    capi.check(capi.lib().tce_w4a16_set_debug_mode((2950 if args.pair == 0 else (2700 if args.pair == 1 else 2800)) + args.waves))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        for heads, kv_heads, pos, m in [(32, 32, 0, 128), (32, 32, 0, 512), (32, 8, 0, 512), (32, 32, 0, 2048), (32, 8, 0, 2048), (32, 32, 1536, 512), (32, 8, 3584, 512), (32, 8, 0, 4096)]:
            max_keys = pos + m
            g = torch.Generator(device=dev).manual_seed(1)
            att = DecodeAttention(heads, hd, max_keys, dev, torch.rand((max_keys, hd), device=dev, generator=g).half(), torch.rand((max_keys, hd), device=dev, generator=g).half(),
                                  kv_heads=kv_heads)
            att.k_cache.normal_(0, 0.8, generator=g)
            att.v_cache.normal_(0, 0.8, generator=g)
            qkv = (torch.randn((m, (heads + 2 * kv_heads) * hd), device=dev, generator=g) * 0.9).half()
            out = torch.empty((m, heads * hd), dtype=torch.float16, device=dev)
            reps = 20 if m <= 2048 else 5
            us = timed(lambda: att.prefill(qkv, pos, out=out, causal=True), reps)
            flops = 4.0 * heads * hd * (m * pos + m * (m + 1) / 2)
            rec = {"waves_forced": args.waves, "pairing": ["rule", "on", "off"][args.pair], "query_heads": heads, "kv_heads": kv_heads, "cached_keys": pos, "new_rows": m, "us": round(us, 1), "TFLOPs_causal": round(flops / us / 1e6, 1),
                   "finite": bool(torch.isfinite(out).all().item())}
            # yardstick: torch SDPA on [1][heads][m][hd] x [1][heads][pos + m][hd], causal on the last m rows
            q = qkv[:, : heads * hd].reshape(m, heads, hd).transpose(0, 1).contiguous()[None]
            K = att.k_cache.repeat_interleave(heads // kv_heads, dim=0)[None].contiguous()
            V = att.v_cache.repeat_interleave(heads // kv_heads, dim=0)[None].contiguous()
            try:
                if pos == 0:
                    fn = lambda: torch.nn.functional.scaled_dot_product_attention(q, K, V, is_causal=True)
                else:
                    bias = torch.zeros((m, pos + m), dtype=torch.float16, device=dev)
                    bias.masked_fill_(torch.triu(torch.ones(m, pos + m, dtype=torch.bool, device=dev), diagonal=pos + 1), float("-inf"))
                    fn = lambda: torch.nn.functional.scaled_dot_product_attention(q, K, V, attn_mask=bias)
                rec["torch_sdpa_us"] = round(timed(fn, reps), 1)
                ref = fn()[0].transpose(0, 1).reshape(m, heads * hd).float()
                # (SDPA has no rotation: compare only when the tables are trivial -- they are not; the shape of the work is what is compared)
            except Exception as e:  # noqa: BLE001
                rec["torch_sdpa_error"] = f"{type(e).__name__}: {e}"[:200]
            if args.chain and m <= 512:
                # the reference's operator sequence on the rotated q and the caches (per head batch): scores = BMM_F16T(alpha), softmax, BMM_F16T(1) on V^T
                qh = q[0].contiguous()                                   # [heads][m][hd]
                Kh, Vt = K[0].contiguous(), V[0].transpose(1, 2).contiguous()  # [heads][keys][hd], [heads][hd][keys]
                sc = torch.empty((heads, m, pos + m), dtype=torch.float16, device=dev)
                oc = torch.empty((heads, m, hd), dtype=torch.float16, device=dev)
                qk, pv = BMM_F16T(1.0 / np.sqrt(hd)), BMM_F16T(1.0)

                def chain():
                    qk.forward(qh, Kh, sc)
                    softmax(sc, sc)
                    pv.forward(sc, Vt, oc)
                rec["reference_arithmetic_operators_us"] = round(timed(chain, 3), 1)
            f.write(json.dumps(rec) + "\n")
            f.flush()
            print(json.dumps(rec))
            del att
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
