"""Every kernel family REPEATED under uneven load: a second stream keeps the memory system busy with large copies while the launches under test run, and every result is held
to the first one bit for bit (all of these kernels are deterministic by construction: fixed summation orders, partial results combined in run / chunk order).

Why this file exists (round 5): one instantiation of the packed prefill GEMM (two wave quartets per 128 x 128 tile, groups of 32) passed every parity test and every fuzz
run on an idle chip and still returned, about once in 700 launches under such load -- and on some boxes on its first execution --, one accumulator register's lanes 48-63
without the first quartet's history (profiles/r5/pk_form2_g32_first_launch.txt).  That instantiation is no longer offered by the dispatcher; this test is what would have
shown it, and it runs every other form and family the same way.  The first result of each case is also checked against the oracle (packed GEMM) -- the other families'
parity lives in their own test files.  (The file name sorts last on purpose: under `pytest -x` everything else has reported before these timing-dependent cases run.)"""
import ctypes as C
import os
import time

import numpy as np
import pytest

from conftest import w4a16_close

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

SECONDS = float(os.environ.get("TCE_UNDER_LOAD_SECONDS", "1.5"))  # per case (a soak: TCE_UNDER_LOAD_SECONDS=10 python -m pytest tests/test_zz_gpu_under_load.py -m gpu)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from tinychatengine_amd import capi
    capi.lib()
    capi.set_gemm_config()
    return torch.device("cuda:0")


class Load:
    """~1 ms of copies (3 x 256 MiB) and a few small launches on a side stream per round()."""

    def __init__(self, dev):
        self.side = torch.cuda.Stream()
        self.a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
        self.b = torch.empty_like(self.a)
        self.small = [torch.randn(1 << 16, device=dev) for _ in range(8)]

    def round(self):
        with torch.cuda.stream(self.side):
            for _ in range(3):
                self.b.copy_(self.a, non_blocking=True)
            for s_ in self.small:
                s_.mul_(1.0001)


def repeat_under_load(dev, launch, results, seconds=SECONDS):
    """launch() enqueues the work on the current stream; results() returns the tensors to compare.  Returns (rounds, rounds whose results differ from the first)."""
    load = Load(dev)
    launch()
    torch.cuda.synchronize()
    first = [r.clone() for r in results()]
    rounds = bad = 0
    t_end = time.time() + seconds
    while time.time() < t_end:
        load.round()
        launch()
        torch.cuda.synchronize()
        rounds += 1
        if not all(torch.equal(r, f) for r, f in zip(results(), first)):
            bad += 1
    torch.cuda.synchronize()
    print(f"[under load] {rounds} rounds x {len(first)} results, {bad} rounds differing from the first", flush=True)  # (pytest -s shows it; the soak's log is kept under profiles/)
    return rounds, bad, first


PK_CASES = [  # M, N, K, G, forced forms (60 = the dispatcher's choice)
    (192, 200, 512, 32, [61, 62, 63, 64, 60]), (700, 392, 3072, 32, [61, 64, 60]), (260, 300, 1408, 64, [61, 62, 63, 64, 60]),
    (513, 2100, 256, 128, [61, 62, 63, 64, 66, 67, 68, 2670, 2671, 2673, 2674, 2675, 60]), (384, 520, 2048, 128, [61, 62, 63, 64, 66, 67, 672, 68, 2669, 2670, 2671, 2672, 2683, 2673, 2674, 2675, 2676, 60]),
    (260, 300, 1408, 128, [2676, 64, 60]),  # round 6, form 16 (two quartets per tile AND the k range handed off between two workgroups) on an odd number of k-blocks
]


@pytest.mark.parametrize("M,N,K,G,modes", PK_CASES)
def test_packed_prefill_gemm_forms_under_load(dev, oracle, M, N, K, G, modes):
    from tinychatengine_amd import capi
    from test_gpu_w4a16_pk import _lin, _quant
    L = capi.lib()
    rng = np.random.default_rng(M + N + K)
    qw, sc, zp = _quant(oracle, N, K, G, seed=M * 3 + N + K, random_zeros=False, zero_scale_groups=0)
    a = rng.standard_normal((M, K)).astype(np.float16)
    ref32 = oracle.w4a16_gemv_q4_6_mt(a, qw, sc, zp, M, N, K, G)
    lin = _lin(dev, qw, sc, zp, G).prepack()
    x = torch.from_numpy(a).to(dev)
    outs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in modes]

    def launch():
        for mode, out in zip(modes, outs):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            out.fill_(float("nan"))
            capi.check(capi.w4a16_forward(lin.desc(x, out), torch.cuda.current_stream().cuda_stream))

    try:
        rounds, bad, first = repeat_under_load(dev, launch, lambda: outs)
    finally:
        L.tce_w4a16_set_debug_mode(60)
    for mode, f in zip(modes, first):
        ok, worst = w4a16_close(f.cpu().numpy(), ref32)
        assert ok, f"mode {mode} {M}x{N}x{K} g{G}: worst |err|/tol = {worst:.3f}"
    assert rounds > 50 and bad == 0, f"{bad} of {rounds} rounds differ from the first result ({M}x{N}x{K} g{G}, modes {modes})"


def test_k_cut_hand_off_and_last_arriver_under_load(dev):
    """The cross-workgroup exchanges of the prefill GEMM (two runs as a directed hand-off; three / four runs through the last arriver) at the full M = 512 sizes."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4, gemm_scratch
    L = capi.lib()
    g = torch.Generator(device=dev).manual_seed(3)
    M, N, K = 512, 4096, 4096
    lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack()
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    modes = [642, 643, 644]
    outs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in modes]

    def launch():
        for mode, out in zip(modes, outs):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            capi.check(capi.w4a16_forward(lin.desc(x, out), torch.cuda.current_stream().cuda_stream))

    try:
        rounds, bad, first = repeat_under_load(dev, launch, lambda: outs)
    finally:
        L.tce_w4a16_set_debug_mode(60)
    assert rounds > 50 and bad == 0, f"{bad} of {rounds} rounds differ"
    assert int(gemm_scratch(dev)[:4096].to(torch.int32).sum().item()) == 0  # the counters back at zero, no hand-off ran out of patience


def test_decode_token_under_load(dev):
    """Four decoder layers of a Llama-3-8B-shaped token, eagerly: fused-norm GEMVs on the packed copy, the fast attention step with its cross-workgroup combine, the residual epilogues."""
    from tinychatengine_amd.decode import SHAPES
    from tinychatengine_amd.decoder_block import DecoderBlock
    shape = SHAPES["llama3-8b"]
    heads, hd, ctx_max, ctx = shape.hidden // 128, 128, 1024, 700
    ang = np.random.default_rng(0).uniform(0, 2 * np.pi, (ctx_max, hd // 2))
    cos = torch.from_numpy(np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)).to(dev)
    sin = torch.from_numpy(np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)).to(dev)
    kv_heads = shape.qkv[1] // 128 if len(shape.qkv) == 3 else heads
    blocks = [DecoderBlock(shape.hidden, heads, shape.ffn, ctx_max, dev, cos, sin, seed=100 + i, kv_heads=kv_heads) for i in range(4)]
    for b in blocks:
        b.attention.k_cache.normal_(0, 0.8)
        b.attention.v_cache.normal_(0, 0.8)
    hid0 = torch.randn(1, shape.hidden, device=dev).to(torch.float16)
    hid = hid0.clone()

    def launch():
        hid.copy_(hid0)
        for b in blocks:
            b.step(hid, ctx - 1)

    rounds, bad, first = repeat_under_load(dev, launch, lambda: [hid])
    assert bool(torch.isfinite(first[0].float()).all().item())
    assert rounds > 50 and bad == 0, f"{bad} of {rounds} tokens differ from the first"


def test_w8a8_gemm_under_load(dev):
    """The int8 GEMM on the OPT-125M shapes (512 rows, and the 108-row launch whose k-steps are cut across workgroups through the scratch area)."""
    from tinychatengine_amd import capi
    L = capi.lib()
    g = torch.Generator(device=dev).manual_seed(5)
    ri = lambda *s: torch.randint(-127, 128, s, device=dev, generator=g, dtype=torch.int32).to(torch.int8)
    scratch = capi.w8a8_scratch(dev)
    sets = []
    for (M, N, K) in [(512, 768, 768), (512, 3072, 768), (512, 768, 3072), (108, 768, 3072)]:
        A, W, b, o = ri(M, K), ri(N, K), ri(N), torch.empty(M, N, dtype=torch.int8, device=dev)
        d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=A.data_ptr(), B=W.data_ptr(), bias=b.data_ptr(), C=o.data_ptr(), alpha=0.0005, beta=0.02, q_min=-128, q_max=127,
                          bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8)
        sets.append((d, A, W, b, o))

    def launch():
        for s_ in sets:
            capi.check(capi.w8a8_matmul_v2(s_[0], torch.cuda.current_stream().cuda_stream, scratch.data_ptr()))

    rounds, bad, _ = repeat_under_load(dev, launch, lambda: [s_[4] for s_ in sets])
    assert rounds > 50 and bad == 0, f"{bad} of {rounds} rounds differ"


def test_prompt_and_chunk_through_decoder_layers_under_load(dev):
    """512 prompt rows (the full-size prefill GEMMs -- q/k/v, o_proj + residual, gate/up with the pair epilogue, down_proj at K = 14336 --, the prefill attention, the norms) and an
    8-row chunk behind them (the small-batch kernels), through two Llama-3-8B-shaped layers."""
    from tinychatengine_amd.decode import SHAPES
    from tinychatengine_amd.decoder_block import DecoderBlock
    shape = SHAPES["llama3-8b"]
    heads, hd, ctx_max = shape.hidden // 128, 128, 1024
    ang = np.random.default_rng(0).uniform(0, 2 * np.pi, (ctx_max, hd // 2))
    cos = torch.from_numpy(np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)).to(dev)
    sin = torch.from_numpy(np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)).to(dev)
    kv_heads = shape.qkv[1] // 128 if len(shape.qkv) == 3 else heads
    blocks = [DecoderBlock(shape.hidden, heads, shape.ffn, ctx_max, dev, cos, sin, seed=300 + i, kv_heads=kv_heads).prepare_prefill() for i in range(2)]
    rows0 = (torch.randn(512, shape.hidden, device=dev) * 0.5).to(torch.float16)
    chunk0 = (torch.randn(8, shape.hidden, device=dev) * 0.5).to(torch.float16)
    rows, chunk = rows0.clone(), chunk0.clone()

    def launch():
        rows.copy_(rows0)
        chunk.copy_(chunk0)
        for b in blocks:
            b.prefill(rows, 0)
        for b in blocks:
            b.prefill(chunk, 512)

    rounds, bad, first = repeat_under_load(dev, launch, lambda: [rows, chunk], seconds=SECONDS + 1.0)
    assert all(bool(torch.isfinite(f.float()).all().item()) for f in first)
    assert rounds > 30 and bad == 0, f"{bad} of {rounds} rounds differ from the first"


def test_opt_int8_layer_under_load(dev):
    """An OPT-125M-shaped W8A8 decoder layer (LayerNormQ fused into the projections, the int8 attention, fc1 / fc2): a 108-row prompt and a decode step."""
    from tinychatengine_amd.opt_layer import Int8OPTDecoderLayer
    embed, heads, ffn, max_keys = 768, 12, 3072, 256
    layer = Int8OPTDecoderLayer(embed, heads, ffn, max_keys, 108, dev, seed=11)
    g = torch.Generator(device=dev).manual_seed(2)
    h0 = torch.empty(108, embed, device=dev).normal_(0, 2, generator=g)
    h1 = torch.empty(1, embed, device=dev).normal_(0, 2, generator=g)
    mask0 = torch.zeros(108, 108, device=dev)
    mask0.masked_fill_(torch.ones(108, 108, device=dev, dtype=torch.bool).triu(1), float(np.finfo(np.float32).min))
    mask1 = torch.zeros(1, 109, device=dev)
    a, b = h0.clone(), h1.clone()

    def launch():
        a.copy_(h0)
        b.copy_(h1)
        layer.step(a, 0, mask0)
        layer.step(b, 108, mask1)

    rounds, bad, first = repeat_under_load(dev, launch, lambda: [a, b])
    assert all(bool(torch.isfinite(f).all().item()) for f in first)
    assert rounds > 50 and bad == 0, f"{bad} of {rounds} rounds differ from the first"


def test_headline_plan_replay_under_load(dev):
    """bench.py's timed path -- the decode token's W4A16 launches on the packed copies as ONE graph replay (tce_plan) -- on four Llama-3-8B-shaped layers + lm_head."""
    from tinychatengine_amd.decode import SHAPES, DecodeLinears
    dl = DecodeLinears(SHAPES["llama3-8b"], device=dev, layers=4, prepack=True)
    plan = dl.make_plan()
    outs = [*dl.out_qkv, dl.out_o, dl.out_gate, dl.out_up, dl.out_down, dl.logits]
    try:
        rounds, bad, first = repeat_under_load(dev, lambda: plan.launch(torch.cuda.current_stream().cuda_stream), lambda: outs)
        plan.status()
    finally:
        plan.close()
    assert all(bool(torch.isfinite(f.float()).all().item()) for f in first)
    assert rounds > 50 and bad == 0, f"{bad} of {rounds} replays differ from the first"
