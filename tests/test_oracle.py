"""CPU: pin oracle/tce_oracle.c (the restatement) against (a) the committed golden vectors, which were produced by the
reference's own code (tests/golden/make_golden.py), and (b) the reference build itself when oracle/_ref is present."""
import numpy as np
import pytest


def _seq(codes):
    return (codes[:, 0::2] | (codes[:, 1::2] << 4)).astype(np.uint8)


# ---------------- golden vectors (always run) ----------------
def test_quantizer_matches_reference_python_quantizer(oracle, golden):
    M, N, K, G = golden["w4_dims"]
    qw, sc, zp, codes, d = oracle.quantize_q4_6(golden["w4_w"], int(G))
    assert np.array_equal(qw, golden["w4_qweight"])
    assert np.array_equal(sc.view(np.uint16), golden["w4_scales"].view(np.uint16))
    assert np.array_equal(zp, golden["w4_zeros"])
    assert sc.shape[1] == 16 and zp.shape[1] == 2  # K=1408: 11 groups padded to zeros_width 2 (quantize_methods.py:9-21)
    M, N, K, G = golden["awq_dims"]
    codes, d = oracle.group_quantize(golden["awq_w"], int(G))
    q5, s5, z5 = oracle.pack_q4_5(codes, d, int(N), int(K), int(G))
    assert np.array_equal(q5, golden["awq_qweight"])
    assert np.array_equal(s5.view(np.uint16), golden["awq_scales"].view(np.uint16))
    assert np.array_equal(z5, golden["awq_zeros"])


def test_w4_naive_and_gemv_oracles_match_golden(oracle, golden):
    M, N, K, G = (int(v) for v in golden["w4_dims"])
    codes = oracle.unpack_q4_6(golden["w4_qweight"], N, K)
    s32 = golden["w4_scales"][:, : K // G].astype(np.float32)
    a = golden["w4_a"]
    out = oracle.naive_mat_mul_int4(a.astype(np.float32), _seq(codes), s32, 8.0, M, N, K, G)
    assert np.array_equal(out, golden["w4_expected_f32"])
    c32, c16 = oracle.w4a16_gemv_q4_6(a, golden["w4_qweight"], golden["w4_scales"], golden["w4_zeros"], M, N, K, G)
    assert np.array_equal(c32, golden["w4_expected_f32"])  # zero point 8 everywhere -> identical to the naive oracle
    assert np.array_equal(c16.view(np.uint16), golden["w4_expected_f32"].astype(np.float16).view(np.uint16))


def test_awq_fp16_oracle_matches_golden(oracle, golden):
    M, N, K, G = (int(v) for v in golden["awq_dims"])
    out = oracle.naive_mat_mul_fp16_int4(golden["awq_a"], golden["awq_qweight"], golden["awq_scales"], M, N, K, G)
    assert np.array_equal(out.view(np.uint16), golden["awq_expected_f16"].view(np.uint16))


def test_int8_oracles_match_golden(oracle, golden):
    M, N, K = (int(v) for v in golden["i8_dims"])
    al, be = (float(v) for v in golden["i8_alpha_beta"])
    A, B, b8, bf, Bb = golden["i8_A"], golden["i8_B"], golden["i8_bias8"], golden["i8_biasf"], golden["i8_Bb"]
    assert np.array_equal(oracle.int8_matmul_bias_i8(A, B, b8, al, be, -128, 127, M, N, K), golden["i8_bias_i8"])
    assert np.array_equal(oracle.int8_matmul_bias_i8(A, B, b8, al, be, 0, 127, M, N, K), golden["i8_bias_i8_relu"])
    assert np.array_equal(golden["i8_bias_i8"], golden["i8_bias_i8_over_column"])
    assert np.array_equal(oracle.int8_matmul_nobias_i8(A, B, al, -128, 127, M, N, K), golden["i8_nobias_i8"])
    assert np.array_equal(oracle.int8_matmul_nobias_i8(A, Bb, al, -128, 127, M, N, K, batch=True), golden["i8_nobias_batch_i8"])
    assert np.array_equal(oracle.int8_matmul_bias_f32(A, B, bf, al, M, N, K), golden["i8_bias_f32"])
    assert np.array_equal(oracle.int8_matmul_nobias_f32(A, B, al, M, N, K), golden["i8_nobias_f32"])
    assert np.array_equal(oracle.int8_matmul_nobias_f32(A, Bb, al, M, N, K, batch=True), golden["i8_nobias_batch_f32"])
    assert np.array_equal(oracle.int8_matmul_nobias_i8(golden["i8_tie_A"], golden["i8_tie_B"], 0.5, -128, 127, 16, 4, 1),
                          golden["i8_tie_nobias_i8"])
    assert golden["i8_tie_nobias_i8"][9, 0] == 1 and golden["i8_tie_nobias_i8"][7, 0] == -1  # +-0.5 round away from zero
    assert np.array_equal(oracle.naive_mat_mul_int8(A, np.ascontiguousarray(B.T), 3, -2, 0.02, 0.01, 0.05, -128, 127, M, N, K),
                          golden["i8_naive"])
    assert golden["i8_bias_i8"].min() == -128 and golden["i8_bias_i8"].max() == 127  # the clamp is exercised


def test_fp32_oracle_matches_golden(oracle, golden):
    out = oracle.fp32_matmul_transposed(golden["f32_A"], golden["f32_B"], None, 5, 7, 64)
    assert np.array_equal(out, golden["f32_expected"]) and np.array_equal(out, golden["f32_expected_imp"])


def test_f16_conversions_match_numpy(oracle):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(4000) * 10.0 ** rng.integers(-9, 6, 4000), [0.0, -0.0, 65504.0, 65520.0, 1e-8, 6e-8, 2 ** -24, 2 ** -25]])
    for v in x.astype(np.float32):
        assert oracle.lib.orc_f32_to_f16(float(v)) == int(np.float32(v).astype(np.float16).view(np.uint16))
    for v in x:
        assert oracle.lib.orc_f64_to_f16(float(v)) == int(np.float64(v).astype(np.float16).view(np.uint16))
    for bits in list(range(0, 0x7C00, 37)) + [0x8001, 0x83FF, 0xFBFF]:
        assert oracle.lib.orc_f16_to_f32(bits) == float(np.uint16(bits).view(np.float16))


# ---------------- live cross-check against the reference build (when present) ----------------
@pytest.mark.parametrize("M,N,K,G", [(1, 64, 512, 128), (3, 40, 256, 32), (2, 24, 384, 64), (1, 16, 1408, 128)])
def test_w4_against_reference_build(oracle, reference, M, N, K, G):
    rng = np.random.default_rng(M * 1000 + N + K + G)
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    qw, sc, zp, codes, d = oracle.quantize_q4_6(w, G)
    a = rng.standard_normal((M, K)).astype(np.float16)
    s32 = sc[:, : K // G].astype(np.float32)
    a32 = a.astype(np.float32)
    assert np.array_equal(oracle.naive_mat_mul_int4(a32, _seq(codes), s32, 8.0, M, N, K, G),
                          reference.naive_mat_mul_int4(a32, _seq(codes), s32, 8.0, M, N, K, G))
    off = (rng.standard_normal(s32.shape) * 0.01).astype(np.float32)
    assert np.array_equal(oracle.naive_mat_mul_int4_with_offset(a32, _seq(codes), s32, off, 8.0, M, N, K, G),
                          reference.naive_mat_mul_int4_with_offset(a32, _seq(codes), s32, off, 8.0, M, N, K, G))
    if G == 32:
        assert np.array_equal(oracle.ref_int4_fast(a32, _seq(codes), s32, off, M, N, K), reference.ref_int4_fast(a32, _seq(codes), s32, off, M, N, K))
    if N % 8 == 0:
        q5, s5, _ = oracle.pack_q4_5(codes, d, N, K, G)
        assert np.array_equal(oracle.naive_mat_mul_fp16_int4(a, q5, s5, M, N, K, G).view(np.uint16),
                              reference.naive_mat_mul_fp16_int4(a, q5, s5, M, N, K, G).view(np.uint16))


@pytest.mark.parametrize("M,N,K", [(1, 32, 64), (17, 48, 96), (5, 8, 33)])
def test_int8_against_reference_build(oracle, reference, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = rng.integers(-128, 128, (M, K), dtype=np.int8)
    B = rng.integers(-128, 128, (N, K), dtype=np.int8)
    Bb = rng.integers(-128, 128, (M, N, K), dtype=np.int8)
    b8 = rng.integers(-128, 128, N, dtype=np.int8)
    bf = rng.standard_normal(N).astype(np.float32)
    for al, be in [(0.0005035400390625, 0.02130126953125), (0.0071, 0.13), (1.0, 1.0)]:
        for qmin in (-128, 0):
            assert np.array_equal(oracle.int8_matmul_bias_i8(A, B, b8, al, be, qmin, 127, M, N, K), reference.int8_matmul_bias_i8(A, B, b8, al, be, qmin, 127, M, N, K))
            assert np.array_equal(oracle.int8_matmul_nobias_i8(A, B, al, qmin, 127, M, N, K), reference.int8_matmul_nobias_i8(A, B, al, qmin, 127, M, N, K))
        assert np.array_equal(oracle.int8_matmul_nobias_i8(A, Bb, al, -128, 127, M, N, K, batch=True), reference.int8_matmul_nobias_i8(A, Bb, al, -128, 127, M, N, K, batch=True))
        assert np.array_equal(oracle.int8_matmul_bias_f32(A, B, bf, al, M, N, K), reference.int8_matmul_bias_f32(A, B, bf, al, M, N, K))
        assert np.array_equal(oracle.int8_matmul_nobias_f32(A, B, al, M, N, K), reference.int8_matmul_nobias_f32(A, B, al, M, N, K))
        assert np.array_equal(oracle.int8_matmul_nobias_f32(A, Bb, al, M, N, K, batch=True), reference.int8_matmul_nobias_f32(A, Bb, al, M, N, K, batch=True))


def test_config1_reference_shape_runs_on_cpu(oracle, reference):
    """BASELINE config #1: kernels/ref-class naive_mat_mul_int4 on CPU, M=1, N=K=4096, group 128 (plumbing, no GPU)."""
    rng = np.random.default_rng(1234)
    N = K = 4096
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    qw, sc, zp, codes, d = oracle.quantize_q4_6(w, 128)
    a = rng.standard_normal((1, K)).astype(np.float16)
    ref = reference.naive_mat_mul_int4(a.astype(np.float32), _seq(codes), sc[:, :32].astype(np.float32), 8.0, 1, N, K, 128)
    c32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, 1, N, K, 128)
    assert np.array_equal(c32, ref)


def test_reference_struct_size_is_what_the_adapter_assumes(golden):
    assert int(golden["sizeof_matmul_params"][0]) == 416


def test_x86_interleave_branch_and_avx_baseline_sanity(oracle):
    """kernels/matmul_int4.cc:78-104 (QM_x86 nibble order) restated and pinned; and the AVX W4A8 fast path that
    bench.py times as the CPU baseline agrees with it to the reference's own tolerance (test_ops.cc:648-653, MSE 7e-4),
    which also proves the q4_3 packing used for the timing is the one the AVX kernel expects."""
    import ctypes as C
    import os
    from oracle import oracle as O
    if not os.path.exists(O.REF_X86NAIVE_SO):
        pytest.skip("oracle/_ref/libtce_ref_x86naive.so not built")
    ref = C.CDLL(O.REF_X86NAIVE_SO)
    rng = np.random.default_rng(7)
    M, N, K = 2, 48, 512
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    codes, d = oracle.group_quantize(w, 32)
    B = O.ReferenceAVX.pack_q4_3(codes)
    A = rng.standard_normal((M, K)).astype(np.float32)
    mine = O.naive_mat_mul_int4_x86(oracle.lib, "orc_", A, B, d, M, N, K)
    theirs = O.naive_mat_mul_int4_x86(ref, "ref_", A, B, d, M, N, K)
    assert np.array_equal(mine, theirs)
    generic = oracle.naive_mat_mul_int4(A, oracle.pack_sequential(codes, N, K), d.reshape(N, -1), 8.0, M, N, K, 32)
    assert np.abs(mine - generic).max() < 1e-5  # same math, different summation order
    if O.have_ref_avx():
        avx = O.ReferenceAVX(num_thread=4).w4a8(A[:1], B, d.reshape(N, -1), 1, N, K)
        assert float(np.mean((avx - mine[:1]) ** 2)) <= 7e-4


@pytest.mark.parametrize("isa,G", [("arm", 32), ("arm", 128), ("metal", 32), ("metal", 128)])
def test_arm_and_metal_nibble_orders_restated_and_pinned(oracle, isa, G):
    """SURVEY 8a row a7 (round 4): the QM_ARM (kernels/matmul_int4.cc:50-76) and QM_METAL (:16-49) flavours of naive_mat_mul_int4 -- two more nibble orders of the
    same weights, CPU-ISA storage layouts with no caller on the GPU path -- restated in oracle/tce_oracle.c and held to the reference's own function built under that
    flavour (oracle/_ref/libtce_ref_{arm,metal}naive.so), bit for bit, on random bytes (the branches are pure byte arithmetic; G = 128 exercises the ARM branch's
    overlapping runs as written).  Against the generic branch on the bytes re-ordered accordingly: the same products, another summation order."""
    import ctypes as C
    import os
    from oracle import oracle as O
    path = O.REF_ARMNAIVE_SO if isa == "arm" else O.REF_METALNAIVE_SO
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not built")
    ref = C.CDLL(path)
    rng = np.random.default_rng(17 + G)
    M, N, K = 3, 40, 512
    B = rng.integers(0, 256, (N, K // 2), dtype=np.uint8)
    d = (rng.random((N, K // G)).astype(np.float32) * 0.02 + 0.001)
    A = rng.standard_normal((M, K)).astype(np.float32)
    mine = O.naive_mat_mul_int4_isa(oracle.lib, "orc_", isa, A, B, d, M, N, K, G)
    theirs = O.naive_mat_mul_int4_isa(ref, "ref_", isa, A, B, d, M, N, K, G)
    assert np.array_equal(mine, theirs)
    if isa == "metal" or G == 32:  # the layouts as orders of codes along k (ARM with G > 32 overlaps its runs: not a permutation)
        lo, hi = (B & 0x0F).astype(np.int64), (B >> 4).astype(np.int64)
        codes = np.empty((N, K), np.int64)
        if isa == "metal":
            codes.reshape(N, K // 8, 8)[:, :, :4] = lo.reshape(N, K // 8, 4)
            codes.reshape(N, K // 8, 8)[:, :, 4:] = hi.reshape(N, K // 8, 4)
        else:
            codes.reshape(N, K // 32, 32)[:, :, :16] = lo.reshape(N, K // 32, 16)
            codes.reshape(N, K // 32, 32)[:, :, 16:] = hi.reshape(N, K // 32, 16)
        want = ((codes - 8) * np.repeat(d.astype(np.float64), G, axis=1)) @ A.astype(np.float64).T
        assert np.abs(mine - want.T).max() < 1e-4


def test_glue_ops_are_binary16_arithmetic(oracle):
    """orc_add_half / orc_silu_mul_half (Int4llamaDecoderLayer.cu:12-30) against numpy's float16 arithmetic, which rounds
    every operation to binary16 like __hadd / __hmul / __hdiv; exp is the float exponential rounded to half."""
    rng = np.random.default_rng(8)
    a = (rng.standard_normal(20000) * 4).astype(np.float16)
    b = (rng.standard_normal(20000) * 4).astype(np.float16)
    a[:8] = np.array([0.0, -0.0, 65504, -65504, 6e-8, -6e-8, 11.09, -17.0], np.float16)
    assert np.array_equal(oracle.add_half(a, b).view(np.uint16), (a + b).view(np.uint16))
    one = np.float16(1)
    with np.errstate(over="ignore"):
        e = np.exp((-a).astype(np.float32)).astype(np.float16)
        want = (a * (one / (one + e))) * b
    got = oracle.silu_mul_half(a, b)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_rmsnorm_restatement_against_float64(oracle):
    """orc_rmsnorm_half (generalT5LayerNorm, LlamaRMSNorm.cu:68-93): the reference's summation order in fp32 stays within
    a few fp32 ulps of the exact sum, so at most a handful of outputs may sit one binary16 step away from the float64
    evaluation; the clamp keeps huge products finite."""
    rng = np.random.default_rng(9)
    for n in (4096, 11008, 520, 100):
        x = (rng.standard_normal((3, n)) * 3).astype(np.float16)
        g = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
        got = oracle.rmsnorm_half(x, g, 1e-6)
        xd = x.astype(np.float64)
        rs = 1.0 / np.sqrt((xd * xd).mean(axis=1, keepdims=True) + 1e-6)
        want = (xd * rs * g).astype(np.float16)
        bad = got.view(np.uint16) != want.view(np.uint16)
        assert bad.mean() < 5e-3
        assert np.allclose(got.astype(np.float32), want.astype(np.float32), rtol=2e-3, atol=1e-6)
    big = np.full((1, 64), 60000.0, np.float16); big[0, 1:] = 0
    out = oracle.rmsnorm_half(big, np.full(64, 1e4, np.float32), 1e-6)
    assert np.isfinite(out.astype(np.float32)).all() and float(out[0, 0]) == float(np.float16(64504.0))


def test_layernorm_q_restatement(oracle):
    """orc_layernorm_q (LayerNormQ.cc:12-52) against an independent numpy float32 evaluation with sequential sums."""
    rng = np.random.default_rng(12)
    x = (rng.standard_normal((5, 96)) * 2).astype(np.float32); w = (6 + rng.standard_normal(96)).astype(np.float32); b = rng.standard_normal(96).astype(np.float32)
    got = oracle.layernorm_q(x, w, b)
    want = np.empty_like(got)
    for r in range(5):
        mean = np.float32(0)
        for v in x[r]: mean = np.float32(mean + v)
        mean = np.float32(mean / np.float32(96))
        sq = np.float32(0)
        for v in x[r]:
            d = np.float32(v - mean); sq = np.float32(sq + np.float32(d * d))
        std = np.float32(np.sqrt(np.float32(np.float32(sq / np.float32(96)) + np.float32(0.00001))))
        f = ((x[r] - mean) / std * w + b).astype(np.float32)
        want[r] = np.where(f >= 0, np.floor(f + np.float32(0.5)), np.ceil(f - np.float32(0.5))).astype(np.int8)
    assert np.array_equal(got, want)


def _round_fraction_to_half_bits(v):
    """Independent restatement: exact rational -> binary16 bits, round to nearest even (finite range)."""
    from fractions import Fraction
    if v == 0:
        return 0
    sign = 0x8000 if v < 0 else 0
    a = abs(v)
    e = 0
    while a >= 2:
        a /= 2; e += 1
    while a < 1:
        a *= 2; e -= 1
    if e < -14:  # subnormal: multiples of 2^-24
        q = abs(v) / Fraction(1, 2 ** 24)
        k = int(q)
        r = q - k
        if r > Fraction(1, 2) or (r == Fraction(1, 2) and (k & 1)):
            k += 1
        return sign | k
    q = a * 1024  # in [1024, 2048)
    k = int(q)
    r = q - k
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and (k & 1)):
        k += 1
    if k == 2048:
        k = 1024; e += 1
    if e > 15:
        return sign | 0x7C00
    return sign | ((e + 15) << 10) | (k - 1024)


def test_hfma_is_one_exact_rounding(oracle):
    """orc_hfma (the primitive of the attention BMM restatement) against exact rational arithmetic with an independent
    rounding routine: random finite halves, subnormals, ties, cancellations."""
    from fractions import Fraction
    rng = np.random.default_rng(7)
    def val(bits):
        return Fraction(float(np.array([bits], np.uint16).view(np.float16)[0]))
    cases = []
    for _ in range(3000):
        a, b, c = (int(x) for x in rng.integers(0, 0x7C00, 3))
        s = [int(x) << 15 for x in rng.integers(0, 2, 3)]
        cases.append((a | s[0], b | s[1], c | s[2]))
    cases += [(0x0001, 0x0001, 0x0001), (0x3C00, 0x3C00, 0xBC00), (0x3C01, 0x3C01, 0xBC02), (0x7BFF, 0x3C00, 0x0001), (0x0400, 0x3800, 0x8200),
              (0x3555, 0x4200, 0x8001), (0x0001, 0x3C00, 0x8001)]
    for a, b, c in cases:
        exact = val(a) * val(b) + val(c)
        want = _round_fraction_to_half_bits(exact)
        got = oracle.hfma(a, b, c)
        if exact == 0:
            assert got in (0, 0x8000)
        else:
            assert got == want, (hex(a), hex(b), hex(c), hex(got), hex(want))


def test_attention_ops_against_float64(oracle):
    """The restated BMM and softmax stay within binary16 accumulation error of a float64 evaluation (sanity of the loops, not parity)."""
    rng = np.random.default_rng(11)
    A = (rng.standard_normal((3, 5, 128)) * 0.5).astype(np.float16)
    B = (rng.standard_normal((3, 40, 128)) * 0.5).astype(np.float16)
    alpha = np.float16(0.08838)
    got = oracle.bmm_f16t(A, B, alpha).astype(np.float64)
    ref = np.einsum("bmk,bnk->bmn", A.astype(np.float64), B.astype(np.float64)) * float(alpha)
    assert np.abs(got - ref).max() < 0.03
    x = (rng.standard_normal((6, 333)) * 3).astype(np.float16)
    p = oracle.softmax_half(x).astype(np.float64)
    e = np.exp(x.astype(np.float64) - x.astype(np.float64).max(axis=1, keepdims=True))
    assert np.abs(p - e / e.sum(axis=1, keepdims=True)).max() < 8e-3 and np.abs(p.sum(axis=1) - 1).max() < 0.03  # the sum is a 333-term binary16 chain
