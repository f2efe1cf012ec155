"""GPU parity tests of the decode kernel on pre-packed copies (csrc/w4a16_gemv_i8.hip: M <= 4 rows as an exact int8 contraction on the
matrix pipe; round 4).  Reached through the C ABI exactly as a host reaches it: tce_w4a16_prepack once per weight tensor, then
tce_w4a16_forward / _forward_group / plans on descriptors that carry the copy.  The checker is oracle/ (the CPU restatement pinned to the
reference's own code); tolerance: conftest.w4a16_close (1e-3 relative, the north-star rule), unchanged.

What is particular to this kernel and therefore tested here: the digit-plane conversion of the activations (block exponent per wave:
outliers, tiny values, subnormals, zeros, inf / NaN), every template (rows per pass 1 / 2 / 4, group sizes 128 / 64 / 32, one and two
tiles per wave, 8 and 16 units per wave), ragged K (the last wave of a workgroup owns fewer units), ragged N (tile rows past N), real
zero points, the fused epilogues, grouped launches, and that a row's bits depend on neither N nor the launch geometry.
"""
import numpy as np
import pytest

from conftest import record_parity, w4a16_close, w4a16_report

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU (they must not silently pass without it)"
    from tinychatengine_amd import capi
    capi.lib()
    capi.set_gemv_config()
    capi.set_gemm_config()
    capi.set_gemv_i8()
    return torch.device("cuda:0")


def _lin(oracle, dev, N, K, G, seed, random_zeros=False, std=0.02):
    """A Linear_half_int4 with a packed copy + the numpy q4_6 arrays the oracle reads."""
    from tinychatengine_amd.linear import Linear_half_int4
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((N, K)) * std).astype(np.float32)
    qw, sc, zp, _, _ = oracle.quantize_q4_6(w, G)
    if random_zeros:
        nib = rng.integers(0, 16, (N, zp.shape[1] * 8), dtype=np.uint32)
        zp = (nib.reshape(N, -1, 8) << (np.arange(8, dtype=np.uint32) * 4)).sum(axis=2).astype(np.uint32)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    lin = Linear_half_int4(t(qw.view(np.int32)), t(sc.view(np.float16)), t(zp.view(np.int32)), G).prepack()
    assert lin.packed is not None
    return lin, (qw, sc, zp)


def _check(got, ref32, what):
    assert not np.isnan(got.astype(np.float32)).any(), f"{what}: output has NaN / unwritten elements"
    ok, worst = w4a16_close(got, ref32)
    assert ok, f"{what}: worst |err|/tol = {worst:.3f} (tol = 1e-3*max(|ref|, rms/64))"


def _fwd(lin, x, flags=0, out=None):
    from tinychatengine_amd import capi
    m = x.shape[0]
    n_out = lin.out_features // 2 if flags & capi.TCE_W4_SILU_MUL_PAIRS else lin.out_features
    if out is None:
        out = torch.full((m, n_out), float("nan"), dtype=torch.float16, device=x.device)
    d = lin.desc(x, out, flags=flags)
    assert capi.describe_dispatch(d).startswith("gemv-i8"), capi.describe_dispatch(d)
    capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out


# M, N, K, G: every template, K tails (K / 128 not a multiple of the units per wave), N tails (N % 16 != 0), K > 16384 (16 units per wave)
SHAPES = [
    (1, 256, 4096, 128), (1, 100, 1408, 128), (1, 4096, 4096, 128), (1, 520, 11008, 128), (1, 264, 14336, 128), (1, 1024, 5120, 128), (1, 48, 13824, 128),
    (1, 16, 128, 128), (1, 72, 1024, 128), (1, 40, 28672, 128), (1, 136, 16512, 128),
    (2, 132, 2560, 128), (2, 4096, 4096, 128), (3, 200, 11008, 128), (4, 96, 4096, 128), (4, 264, 14336, 128),
    (1, 128, 2048, 64), (2, 104, 4096, 64), (1, 96, 1024, 32), (1, 520, 4096, 32), (1, 64, 11008, 64),
]


@pytest.mark.parametrize("M,N,K,G", SHAPES)
def test_i8_gemv_matches_oracle(dev, oracle, M, N, K, G):
    lin, (qw, sc, zp) = _lin(oracle, dev, N, K, G, seed=M * 7 + N + K)
    rng = np.random.default_rng(N + K)
    a = rng.standard_normal((M, K)).astype(np.float16)
    ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, G)
    assert lin.zeros_are_8
    got = _fwd(lin, torch.from_numpy(a).to(dev)).cpu().numpy()
    _check(got, ref32, f"i8 gemv {M}x{N}x{K} g{G}")
    # without the caller's promise the zero points are read (and the extra MFMA pair runs): same values, within the same rule
    lin.zeros_are_8 = False
    got2 = _fwd(lin, torch.from_numpy(a).to(dev)).cpu().numpy()
    _check(got2, ref32, f"i8 gemv {M}x{N}x{K} g{G}, zero points read")
    assert np.array_equal(got.view(np.uint16), got2.view(np.uint16)), "(8 - z) * sum X is exactly zero for z = 8: the two forms must agree bit for bit"


@pytest.mark.parametrize("M,N,K,G", [(1, 192, 4096, 128), (2, 72, 1408, 128), (4, 64, 11008, 128), (2, 64, 1024, 64), (1, 40, 512, 32), (1, 264, 28672, 128)])
def test_i8_gemv_reads_real_zero_points(dev, oracle, M, N, K, G):
    lin, (qw, sc, zp) = _lin(oracle, dev, N, K, G, seed=99 + N, random_zeros=True)
    assert not lin.zeros_are_8
    a = np.random.default_rng(5).standard_normal((M, K)).astype(np.float16)
    ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, G)
    _check(_fwd(lin, torch.from_numpy(a).to(dev)).cpu().numpy(), ref32, f"i8 gemv zeros {M}x{N}x{K} g{G}")


def test_i8_activation_ranges(dev, oracle):
    """The conversion's block exponent: one outlier 2^12 times the rest, tiny activations, subnormals, exact zeros, a whole block of zeros, values
    near the binary16 maximum -- against the oracle at the unchanged tolerance (an element 2^19 below its block's maximum is truncated: far inside it)."""
    N, K, G = 264, 4096, 128
    lin, (qw, sc, zp) = _lin(oracle, dev, N, K, G, seed=3)
    rng = np.random.default_rng(11)
    base = rng.standard_normal((1, K)).astype(np.float32)
    cases = {}
    x = base.copy(); x[0, 7] = 4096.0; cases["outlier x4096"] = x
    x = base.copy() * 2.0 ** -6; cases["small"] = x
    x = base.copy(); x[0, ::2] = 0.0; cases["half zeros"] = x
    x = base.copy(); x[0, 1024:2048] = 0.0; cases["a wave's whole block zero"] = x
    x = base.copy() * 9000.0; np.clip(x, -60000, 60000, out=x); cases["near the binary16 maximum"] = x
    x = np.zeros((1, K), np.float32); x[0, 5] = 1.0; cases["one non-zero"] = x
    x = base.copy(); x[0, :512] *= 1e-4; x[0, 512:1024] *= 300.0; cases["mixed magnitudes across blocks"] = x
    for name, xv in cases.items():
        a = xv.astype(np.float16)
        ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, 1, N, K, G)
        got = _fwd(lin, torch.from_numpy(a).to(dev)).cpu().numpy()
        if np.isinf(ref32).any() or np.abs(ref32).max() > 60000:  # (the large case may overflow binary16 on both sides)
            fin = np.abs(ref32) < 60000
            _check(got[fin], ref32[fin], f"i8 ranges: {name}")
        else:
            _check(got, ref32, f"i8 ranges: {name}")
    # binary16 SUBNORMAL activations (|x| ~ 3e-6 < 6.1e-5), against weights large enough that the outputs are normal halves (the output grid must not be the error)
    big, (qw2, sc2, zp2) = _lin(oracle, dev, N, K, G, seed=4, std=200.0)
    for name, xv in {"subnormal activations": base * 3e-6, "subnormals and one normal": np.where(np.arange(K) == 9, 0.5, base * 3e-6).astype(np.float32)}.items():
        a = xv.astype(np.float16)
        assert (np.abs(a[a != 0]) < 6.2e-5).mean() > 0.99
        ref32, _ = oracle.w4a16_gemv_q4_6(a, qw2, sc2, zp2, 1, N, K, G)
        _check(_fwd(big, torch.from_numpy(a).to(dev)).cpu().numpy(), ref32, f"i8 ranges: {name}")
    # all-zero input: exact zeros
    got = _fwd(lin, torch.zeros((1, K), dtype=torch.float16, device=dev))
    assert torch.count_nonzero(got).item() == 0


def test_i8_nonfinite_activations_poison_the_row(dev, oracle):
    """The reference's sum holds NaN (or +-inf) wherever a non-finite activation meets a row; this kernel: NaN for every output of that activation row, and only that row."""
    N, K = 64, 4096
    lin, _ = _lin(oracle, dev, N, K, 128, seed=8)
    x = torch.randn((2, K), device=dev).to(torch.float16)
    x[1, 100] = float("inf")
    y = _fwd(lin, x)
    assert torch.isfinite(y[0]).all() and torch.isnan(y[1]).all()
    x[1, 100] = float("nan")
    y = _fwd(lin, x)
    assert torch.isfinite(y[0]).all() and torch.isnan(y[1]).all()


def test_i8_bits_do_not_depend_on_n_or_geometry(dev, oracle):
    """Column shards (rows [r N/P, (r+1) N/P), packed on their own), grouped launches and both tiles-per-wave forms reproduce the plain launch bit for bit."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import forward_group
    for (N, K) in [(4096, 4096), (2048, 11008)]:
        lin, _ = _lin(oracle, dev, N, K, 128, seed=N + K)
        x = torch.randn((1, K), device=dev).to(torch.float16)
        full = _fwd(lin, x)
        parts = [_fwd(lin.shard(r, 8).prepack(), x) for r in range(8)]
        assert torch.equal(torch.cat(parts, dim=1), full)
        try:
            for rows in (1, 2):
                capi.set_gemv_i8(0, rows)
                assert torch.equal(_fwd(lin, x), full), f"tiles per wave {rows}"
        finally:
            capi.set_gemv_i8()
        # grouped with two other linears that share x
        l2, _ = _lin(oracle, dev, 264, K, 128, seed=1)
        l3, _ = _lin(oracle, dev, 1024, K, 128, seed=2)
        outs = [torch.full((1, l.out_features), float("nan"), dtype=torch.float16, device=dev) for l in (l2, lin, l3)]
        forward_group([l2, lin, l3], x, outs)
        torch.cuda.synchronize()
        assert torch.equal(outs[1], full) and torch.equal(outs[0], _fwd(l2, x)) and torch.equal(outs[2], _fwd(l3, x))
        # deterministic run to run
        assert torch.equal(_fwd(lin, x), full)


def test_i8_against_the_fp16_gemv_and_the_switch(dev, oracle):
    """tce_w4a16_set_gemv_i8(1) hands the same descriptors to the fp16 GEMV kernels (the q4_6 arrays): both families inside the tolerance, and close to each other."""
    from tinychatengine_amd import capi
    N, K = 4096, 4096
    lin, (qw, sc, zp) = _lin(oracle, dev, N, K, 128, seed=77)
    x = torch.randn((1, K), device=dev).to(torch.float16)
    ref32, _ = oracle.w4a16_gemv_q4_6(x.cpu().numpy(), qw, sc, zp, 1, N, K, 128)
    y_i8 = _fwd(lin, x)
    try:
        capi.set_gemv_i8(1)
        out = torch.full((1, N), float("nan"), dtype=torch.float16, device=dev)
        d = lin.desc(x, out)
        assert capi.describe_dispatch(d) == "gemv passes=1 kernel=row-block"
        capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    finally:
        capi.set_gemv_i8()
    _check(y_i8.cpu().numpy(), ref32, "i8")
    _check(out.cpu().numpy(), ref32, "fp16 gemv")
    rep_i8, rep_f16 = w4a16_report(y_i8.cpu().numpy(), ref32), w4a16_report(out.cpu().numpy(), ref32)
    record_parity("decode M=1 4096x4096 (gemv-i8)", rep_i8)
    # the integer part of this kernel is exact: its noise must not exceed the fp16-unpack kernel's (whose 1024 bias costs accumulation bits)
    assert rep_i8["err_rms_over_rms"] <= rep_f16["err_rms_over_rms"] * 1.05, (rep_i8, rep_f16)


@pytest.mark.parametrize("M,H,K", [(1, 11008, 4096), (1, 264, 1408), (2, 520, 4096), (4, 136, 2048)])
def test_i8_gate_up_silu_mul_fused(dev, oracle, M, H, K):
    """TCE_W4_SILU_MUL_PAIRS in this kernel's epilogue: binary16 arithmetic of SiLuMul_half on the rounded projections (Int4llamaDecoderLayer.cu:20-30)."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    gate, (gq, gs, gz) = _lin(oracle, dev, H, K, 128, seed=H)
    up, (uq, us, uz) = _lin(oracle, dev, H, K, 128, seed=H + 1)
    il = Linear_half_int4.interleave(gate, up).prepack()
    x = torch.randn((M, K), device=dev).to(torch.float16)
    got = _fwd(il, x, flags=capi.TCE_W4_SILU_MUL_PAIRS)
    # the kernel's own unfused projections (bit-identical rows: a row's bits do not depend on where it sits) -> the oracle's fp16 SiLuMul
    g16, u16 = _fwd(gate, x).cpu().numpy(), _fwd(up, x).cpu().numpy()
    want = oracle.silu_mul_half(g16, u16)
    def key(h):  # binary16 bits -> a monotone integer (distance = half steps)
        b = h.view(np.uint16).astype(np.int32)
        return np.where(b & 0x8000, -(b & 0x7FFF), b)
    d = np.abs(key(got.cpu().numpy()) - key(want))
    assert d.max() <= 1, f"{int((d > 1).sum())} outputs differ by more than one half step (the exp caveat allows one)"
    assert (d > 0).mean() < 2e-3


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (1, 4096, 11008), (3, 264, 2048), (1, 100, 1408)])
def test_i8_projection_plus_residual_fused(dev, oracle, M, N, K):
    from tinychatengine_amd import capi
    lin, _ = _lin(oracle, dev, N, K, 128, seed=N)
    x = torch.randn((M, K), device=dev).to(torch.float16)
    res = torch.randn((M, N), device=dev).to(torch.float16)
    y = _fwd(lin, x)
    got = _fwd(lin, x, flags=capi.TCE_W4_ADD_TO_C, out=res.clone())
    assert torch.equal(got, res + y)  # hadd of two halves: torch's fp16 add rounds the same way


@pytest.mark.parametrize("Ns,K,pairs", [([4096, 1024, 1024], 4096, False), ([520, 264], 11008, False), ([72], 1408, False), ([2752], 4096, True), ([1024], 14336, False),
                                          ([11008, 11008], 4096, False), ([22016], 4096, True), ([16400], 4096, False)])  # (>= 1024 tiles: two tiles per wave, an odd tile count)
def test_i8_rmsnorm_prologue_fused(dev, oracle, Ns, K, pairs):
    """The fused RMSNorm prologue in this kernel (input_layernorm + q/k/v, post_attention_layernorm + gate/up, Int4llamaDecoderLayer.cu:78, 92-99): the same rs bits
    as tce_rmsnorm_half (the shape-independent order of tce_common.hpp), the same normalised halves, hence outputs bit-identical to the two-launch form -- on
    grouped launches, on K > 8192 (two pieces per slot of the order) and together with the pair epilogue; and the normalisation itself against the oracle."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import forward_group_rmsnorm, rmsnorm_half
    lins = [_lin(oracle, dev, n, K, 128, seed=n + K)[0] for n in Ns]
    x = (torch.randn((1, K), device=dev) * 3.0).to(torch.float16)
    gamma = (1.0 + 0.1 * torch.randn(K, device=dev)).float()
    eps = 1e-6
    xn = rmsnorm_half(x, gamma, eps)
    # (tce_rmsnorm_half against the oracle: tests/test_gpu_epilogues.py::test_rmsnorm_kernel_matches_oracle)
    flags = capi.TCE_W4_SILU_MUL_PAIRS if pairs else 0
    outs = []
    for l in lins:
        o = torch.full((1, l.out_features // 2 if pairs else l.out_features), float("nan"), dtype=torch.float16, device=dev)
        d = l.desc(x, o, flags=flags, gamma=gamma, eps=eps)
        assert capi.describe_dispatch(d).startswith("gemv-i8"), capi.describe_dispatch(d)
        outs.append(o)
    if len(lins) > 1:
        forward_group_rmsnorm(lins, x, outs, gamma, eps)
    else:
        capi.check(capi.w4a16_forward(lins[0].desc(x, outs[0], flags=flags, gamma=gamma, eps=eps), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for l, o in zip(lins, outs):
        assert torch.equal(o, _fwd(l, xn, flags=flags)), "fused prologue != rmsnorm launch + plain launch"


@pytest.mark.parametrize("N,K,zeros", [(4096, 4096, False), (4096, 11008, False), (264, 1408, False), (5120, 13824, False), (11008, 4096, False), (4096, 4096, True), (1024, 28672, False), (1024, 28672, True)])
def test_i8_residual_plus_next_rmsnorm_fused(dev, oracle, N, K, zeros):
    """tce_w4a16_forward_residual_rmsnorm (o_proj / down_proj + residual add + the RMSNorm that follows, Int4llamaDecoderLayer.cu:86-99, 107-108): the residual row as
    TCE_W4_ADD_TO_C writes it and the normalised row as tce_rmsnorm_half forms it from that row -- bit for bit, launch after launch on one workspace (the counter
    returns to zero), with a ragged last tile (N % 16 = 8), more than 1024 pieces (two per slot of the order), 16 units per wave, real zero points, and replayed from a graph.
    (K > 16384 with general zero points has no one-launch form: the entry point serves it as the two launches it is defined by -- round 6)"""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import rmsnorm_half
    lin, _ = _lin(oracle, dev, N, K, 128, seed=N + K, random_zeros=zeros)
    gamma = (1.0 + 0.1 * torch.randn(N, device=dev)).float()
    eps = 1e-6
    ws = torch.zeros(int(capi.lib().tce_w4a16_residual_rmsnorm_workspace_bytes()), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for it in range(3):
        x = torch.randn((1, K), device=dev).to(torch.float16)
        res = torch.randn((1, N), device=dev).to(torch.float16)
        want_c = _fwd(lin, x, flags=capi.TCE_W4_ADD_TO_C, out=res.clone())
        want_xn = rmsnorm_half(want_c, gamma, eps)
        c = res.clone()
        xn = torch.full((1, N), float("nan"), dtype=torch.float16, device=dev)
        capi.check(capi.w4a16_forward_residual_rmsnorm(lin.desc(x, c, flags=capi.TCE_W4_ADD_TO_C), gamma.data_ptr(), eps, xn.data_ptr(), ws.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(c, want_c), f"residual row, launch {it}"
        assert torch.equal(xn, want_xn), f"normalised row, launch {it}: {(xn != want_xn).sum().item()} of {N} differ"
        assert int(ws.view(torch.int32)[2048].item()) == 0
    # graph replay: the same launch three times per replay (the residual keeps accumulating: compare against the eager sequence)
    c_g, c_e = res.clone(), res.clone()
    xn_g, xn_e = torch.zeros_like(xn), torch.zeros_like(xn)
    for _ in range(3):
        capi.check(capi.w4a16_forward_residual_rmsnorm(lin.desc(x, c_e, flags=capi.TCE_W4_ADD_TO_C), gamma.data_ptr(), eps, xn_e.data_ptr(), ws.data_ptr(), st))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g, stream=s2):
            for _ in range(3):
                capi.check(capi.w4a16_forward_residual_rmsnorm(lin.desc(x, c_g, flags=capi.TCE_W4_ADD_TO_C), gamma.data_ptr(), eps, xn_g.data_ptr(), ws.data_ptr(), s2.cuda_stream))
    c_g.copy_(res)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(c_g, c_e) and torch.equal(xn_g, xn_e)
    # argument checks: no packed copy / no residual flag -> refused
    assert capi.w4a16_forward_residual_rmsnorm(lin.desc(x, c), gamma.data_ptr(), eps, xn.data_ptr(), ws.data_ptr(), st) == capi.TCE_ERR_BAD_ARG


def test_decoder_block_with_the_norms_on_the_producer_side(dev):
    """DecoderBlock.step_chained (q/k/v and gate/up as plain launches on rows normalised by the PREVIOUS launch's residual epilogue) against DecoderBlock.step (the fused
    prologues): the same residual stream, bit for bit, over three layers and two tokens."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.decoder_block import DecoderBlock
    from tinychatengine_amd.linear import rmsnorm_half
    hidden, heads, ffn, ctx = 1024, 8, 2816, 64
    ang = np.random.default_rng(0).uniform(0, 2 * np.pi, (ctx, 64))
    cos = torch.from_numpy(np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)).to(dev)
    sin = torch.from_numpy(np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)).to(dev)
    mk = lambda: [DecoderBlock(hidden, heads, ffn, ctx, dev, cos, sin, seed=50 + i, kv_heads=4) for i in range(3)]
    a, b = mk(), mk()
    final_gamma = torch.ones(hidden, device=dev)
    ws = torch.zeros(int(capi.lib().tce_w4a16_residual_rmsnorm_workspace_bytes()), dtype=torch.uint8, device=dev)
    h0 = torch.randn(1, hidden, device=dev).to(torch.float16)
    for pos in range(2):
        ha, hb = h0.clone() * (pos + 1), h0.clone() * (pos + 1)
        for blk in a:
            blk.step(ha, pos)
        xn = [rmsnorm_half(hb, b[0].gamma1, b[0].eps), torch.empty_like(hb)]
        for i, blk in enumerate(b):
            nxt = b[i + 1].gamma1 if i + 1 < len(b) else final_gamma
            blk.step_chained(hb, xn[i % 2], pos, nxt, xn[(i + 1) % 2], ws)
        torch.cuda.synchronize()
        assert torch.isfinite(ha.float()).all()
        assert torch.equal(ha, hb), f"token {pos}: {(ha != hb).sum().item()} of {hidden} residual values differ"
        assert torch.equal(xn[len(b) % 2], rmsnorm_half(hb, final_gamma, b[0].eps))


FULL = [(4096, 4096), (11008, 4096), (4096, 11008), (14336, 4096), (4096, 14336), (1024, 4096), (12288, 4096), (32000, 4096), (128256, 4096),
        (15360, 5120), (5120, 5120), (13824, 5120), (5120, 13824)]


@pytest.mark.parametrize("N,K", FULL)
def test_i8_full_size_decode(dev, oracle, N, K):
    """configs[1] and [4] at M = 1 on the automatic dispatch WITH a packed copy: every output against the oracle, exact homogeneity under x -> 2x, and 8-way
    column shards bit-identical to the whole."""
    from tinychatengine_amd.linear import Linear_half_int4
    G = 128
    g = torch.Generator(device=dev).manual_seed(1234 + N)
    lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), G).prepack()
    x = torch.empty(1, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    y = _fwd(lin, x)
    ref32 = oracle.w4a16_gemv_q4_6_mt(x.cpu().numpy(), lin.weight.cpu().numpy().view(np.uint32), lin.scale.cpu().numpy(),
                                      lin.zero_point.cpu().numpy().view(np.uint32), 1, N, K, G)
    _check(y.cpu().numpy(), ref32, f"full {N}x{K}")
    rep = w4a16_report(y.cpu().numpy(), ref32)
    record_parity(f"decode M=1 {N}x{K} (gemv-i8)", rep)
    assert rep["frac_fail"] == 0.0 and rep["frac_over_plain"] < 0.01 and rep["worst_plain"] <= 1.0, rep
    y2 = _fwd(lin, x * 2)
    normal = y.abs() >= 2.0 ** -13
    assert torch.equal(y2[normal], (y * 2)[normal]), "x -> 2x must double every (normal) output exactly"
    parts = [_fwd(lin.shard(r, 8).prepack(), x) for r in range(8)]
    assert torch.equal(torch.cat(parts, dim=1), y)


def test_i8_in_plans(dev, oracle):
    """A decode token's launch list on packed copies, captured as a hipGraph plan: replays reproduce the eager launches bit for bit."""
    from tinychatengine_amd.decode import SHAPES as MODEL_SHAPES, DecodeLinears
    dl = DecodeLinears(MODEL_SHAPES["tiny"], device=dev, prepack=True)
    for g in dl.token_launches():
        DecodeLinears._hip_launch(g)
    torch.cuda.synchronize()
    eager = [t.clone() for t in (dl.logits, dl.out_down, dl.out_up, dl.out_qkv[2])]
    for t in (dl.logits, dl.out_down, dl.out_up, dl.out_qkv[2]):
        t.fill_(float("nan"))
    plan = dl.make_plan()
    for _ in range(3):
        plan.launch(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for a, b in zip(eager, (dl.logits, dl.out_down, dl.out_up, dl.out_qkv[2])):
        assert torch.equal(a, b)
