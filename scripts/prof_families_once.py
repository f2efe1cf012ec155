#!/usr/bin/env python3
"""Round 5: a few eager launches of every kernel family, for the profiler (scripts/profile_r5.sh runs this under rocprofv3 once per counter pass).

    python scripts/prof_families_once.py [gemm] [w8a8] [attn] [token]      (default: all four)

  gemm  -- the W4A16 prefill GEMM on the packed copy at M = 512 / 2048 (the dispatcher's choice, and the forms forced by PROF_GEMM_FORMS)
  w8a8  -- tce_w8a8_matmul on the three OPT-125M shapes at M = 512 and the 128-row tile's 512 x 4096 x 4096
  attn  -- the fast decode attention step at 512 and 2048 keys (32 heads x 128; the grouped-query form 32 / 8)
  token -- one whole decode token through 32 decoder layers, eagerly (the fused-norm GEMV launches, the attention step, the residual epilogues)
  shard -- (round 6) rank 0's share of a token of the column-sharded model at 8 and 2 ranks (this GPU playing rank 0; no exchange): a block's seven shards as ONE launch
           (tce_w4a16_forward_independent) and as the four launches of rounds 1-5
Every family uses distinct shapes per case, so (kernel name, grid size) identifies a case in the profiler's tables (summarize_prof.py with TCE_PROF_KEY_GRID=1)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4

fams = [a for a in sys.argv[1:] if not a.startswith("-")] or ["gemm", "w8a8", "attn", "token"]
dev = torch.device("cuda:0")
L = capi.lib()
g = torch.Generator(device=dev).manual_seed(1)
sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
REPS = int(os.environ.get("PROF_REPS", "10"))

if "gemm" in fams:
    forced = [int(f) for f in os.environ.get("PROF_GEMM_FORMS", "").split(",") if f]
    for (M, N, K) in [(512, 4096, 4096), (512, 11008, 4096), (512, 4096, 11008), (2048, 4096, 4096), (2048, 11008, 4096), (2048, 4096, 11008)]:
        lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
        x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        for form in [0] + (forced if M == 2048 and N == 4096 and K == 4096 else []):
            capi.check(L.tce_w4a16_set_debug_mode(60 + form))
            for i in range(REPS):
                capi.check(capi.w4a16_forward(lins[i % 3].desc(x, out), sp()))  # (desc() hands the shared scratch area over for M > 128)
            torch.cuda.synchronize()
        capi.check(L.tce_w4a16_set_debug_mode(60))
        del lins
    torch.cuda.empty_cache()

if "w8a8" in fams:
    ri = lambda *s: torch.randint(-127, 128, s, device=dev, generator=g, dtype=torch.int32).to(torch.int8)
    for (M, N, K) in [(512, 768, 768), (512, 3072, 768), (512, 768, 3072), (512, 4096, 4096)]:
        A = ri(M, K)
        sets = []
        for _ in range(3):
            W, b, o = ri(N, K), ri(N), torch.empty(M, N, dtype=torch.int8, device=dev)
            d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=A.data_ptr(), B=W.data_ptr(), bias=b.data_ptr(), C=o.data_ptr(), alpha=0.0005, beta=0.02, q_min=-128, q_max=127,
                              bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8)
            sets.append((d, W, b, o))
        for i in range(REPS):
            capi.check(L.tce_w8a8_matmul(C.byref(sets[i % 3][0]), sp()))
        torch.cuda.synchronize()
        del sets

if "attn" in fams:
    from tinychatengine_amd.attention_ops import DecodeAttention
    al = int(np.array([0.0884], np.float16).view(np.uint16)[0])
    for kv_heads in (32, 8):
        for t in (512, 2048):
            cos = torch.randn(t + 1, 128, device=dev).half()
            sin = torch.randn(t + 1, 128, device=dev).half()
            atts = [DecodeAttention(32, 128, t, dev, cos, sin, kv_heads=kv_heads) for _ in range(4)]
            qkv = torch.randn((32 + 2 * kv_heads) * 128, device=dev).half()
            for a_ in atts:
                a_.k_cache.normal_(0, 0.8)
                a_.v_cache.normal_(0, 0.8)
            for i in range(REPS):
                atts[i % 4].step(qkv, t - 1)
            torch.cuda.synchronize()
            del atts

if "token" in fams:
    from tinychatengine_amd.decode import SHAPES, DecodeLinears
    from tinychatengine_amd.decoder_block import DecoderBlock
    ctx = 512
    shape = SHAPES["llama3-8b"]
    heads, hd, ctx_max = shape.hidden // 128, 128, 1024
    ang = np.random.default_rng(0).uniform(0, 2 * np.pi, (ctx_max, hd // 2))
    cos = torch.from_numpy(np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)).to(dev)
    sin = torch.from_numpy(np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)).to(dev)
    kv_heads = shape.qkv[1] // 128 if len(shape.qkv) == 3 else heads
    blocks = [DecoderBlock(shape.hidden, heads, shape.ffn, ctx_max, dev, cos, sin, seed=100 + i, kv_heads=kv_heads) for i in range(shape.layers)]
    for b in blocks:
        b.attention.k_cache.normal_(0, 0.8)
        b.attention.v_cache.normal_(0, 0.8)
    hid0 = torch.randn(1, shape.hidden, device=dev).to(torch.float16)
    hid = hid0.clone()
    for _ in range(max(2, REPS // 3)):
        hid.copy_(hid0)
        for b in blocks:
            b.step(hid, ctx - 1)
    torch.cuda.synchronize()
    print("token finite", bool(torch.isfinite(hid.float()).all().item()))
print("families done:", fams)


if "shard" in fams:
    from tinychatengine_amd.decode import SHAPES, DecodeLinears
    st_ = torch.cuda.current_stream().cuda_stream
    for P in (8, 2):
        dlp = DecodeLinears(SHAPES["llama3-8b"], device=dev, rank=0, world=P, prepack=True)
        for rep in range(max(1, REPS // 4)):
            for li in range(dlp.n_layers):   # one launch per block
                capi.w4a16_forward_independent([d for g_ in dlp.block_launches(li) for d in g_], st_)
            torch.cuda.synchronize()
            for li in range(dlp.n_layers):   # four launches per block
                dlp.run_block(li)
            torch.cuda.synchronize()
        del dlp
        torch.cuda.empty_cache()
