"""GPU parity tests of the W8A8 (SmoothQuant) path: BIT-EXACT against the oracle, which is pinned bit-exact to
kernels/ref/matmul_ref_int8.cc (the reference's own tests demand check_two_exact_equal for int8 outputs,
llm/tests/non_cuda/test_ops.cc:204,238,340,373).  Shapes and scalar constants are the reference's test shapes
(test_ops.cc:177-209, 245-276, 311-345, 380-410, 444-473) plus the edge cases it exercises implicitly."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ALPHA, BETA = 0.0005035400390625, 0.02130126953125  # test_ops.cc:179


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from tinychatengine_amd import capi
    capi.lib()
    return torch.device("cuda:0")


def _t(dev, x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _params(dev, A, B, out_dtype, bias=None, alpha=ALPHA, beta=BETA, qmin=-128, qmax=127):
    from tinychatengine_amd.matmul import matmul_params, matrix
    M, K = A.shape
    N = B.shape[-2]
    out = torch.zeros((M, N), dtype=out_dtype, device=dev)
    p = matmul_params(A=matrix(M, K, _t(dev, A)), B=matrix(K, N, _t(dev, B)), C=matrix(M, N, out), alpha=alpha, beta=beta)
    if bias is not None:
        p.bias = matrix(1, N, _t(dev, bias))
    p.C.qparams.q_min, p.C.qparams.q_max = qmin, qmax
    return p, out


def _data(M, N, K, seed, corner=True):
    rng = np.random.default_rng(seed)
    A = rng.integers(-127, 128, (M, K), dtype=np.int8)
    B = rng.integers(-127, 128, (N, K), dtype=np.int8)
    if corner:  # the -128 corner set of SURVEY §8d
        A[0, :] = -128
        B[0, :] = -128
        A[-1, ::3] = -128
    return A, B, rng.integers(-128, 128, N, dtype=np.int8), rng.standard_normal(N).astype(np.float32)


SHAPES = [(108, 3072, 768), (108, 768, 768), (512, 768, 768), (1, 768, 768), (1, 3072, 768), (512, 768, 3072), (108, 2048, 2048),
          (512, 3072, 768),                      # BASELINE config 4, the fc1 shape bench.py times
          (108, 8192, 2048), (108, 2048, 8192),  # OPT-1.3B fc1 / fc2 (test_ops.cc)
          (7, 40, 80), (65, 130, 208), (3, 5, 33), (2, 3, 16), (64, 64, 64),
          (5, 17, 96), (70, 33, 112), (130, 66, 176), (33, 65, 128), (16, 16, 192)]  # K % 64 in {32, 48, 0}: the transposed fragment tails


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_all_linear_variants_bit_exact(dev, oracle, M, N, K):
    from tinychatengine_amd.matmul import MatmulOperator
    op = MatmulOperator()
    A, B, b8, bf = _data(M, N, K, seed=M + N + K)
    for alpha, beta in [(ALPHA, BETA), (0.0071, 0.13)]:
        for qmin in (-128, 0):  # 0 = W8A8B8O8LinearReLU
            p, out = _params(dev, A, B, torch.int8, b8, alpha, beta, qmin)
            op.mat_mul_accelerator_int8_fast_2x2_32unroll(p)
            exp = oracle.int8_matmul_bias_i8(A, B, b8, alpha, beta, qmin, 127, M, N, K)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), exp), f"bias_i8 qmin={qmin} mismatches: {(out.cpu().numpy() != exp).sum()}"
            p, out = _params(dev, A, B, torch.int8, None, alpha, beta, qmin)
            op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias(p)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), oracle.int8_matmul_nobias_i8(A, B, alpha, qmin, 127, M, N, K))
        p, out = _params(dev, A, B, torch.int8, b8, alpha, beta)
        op.mat_mul_accelerator_int8_fast_32unroll_over_column(p)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), oracle.int8_matmul_bias_i8(A, B, b8, alpha, beta, -128, 127, M, N, K))
        p, out = _params(dev, A, B, torch.float32, bf, alpha)
        op.mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32(p)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.int8_matmul_bias_f32(A, B, bf, alpha, M, N, K).view(np.uint32))
        p, out = _params(dev, A, B, torch.float32, bf, alpha)
        op.mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32_over_column(p)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.int8_matmul_bias_f32(A, B, bf, alpha, M, N, K).view(np.uint32))
        p, out = _params(dev, A, B, torch.float32, None, alpha)
        op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32(p)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.int8_matmul_nobias_f32(A, B, alpha, M, N, K).view(np.uint32))


@pytest.mark.parametrize("M,N,K", [(12, 512, 64), (12, 64, 512), (5, 9, 33), (1, 16, 64)])
def test_batch_variants_bit_exact(dev, oracle, M, N, K):
    """*_nobias_batch / *_nobias_ofp32_batch: row i of A has its own B_i (decode-time attention BMMs)."""
    from tinychatengine_amd.matmul import MatmulOperator
    rng = np.random.default_rng(M * N + K)
    A = rng.integers(-128, 128, (M, K), dtype=np.int8)
    Bb = rng.integers(-128, 128, (M, N, K), dtype=np.int8)
    op = MatmulOperator()
    p, out = _params(dev, A, Bb, torch.int8, None, 0.0031)
    op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_batch(p)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oracle.int8_matmul_nobias_i8(A, Bb, 0.0031, -128, 127, M, N, K, batch=True))
    p, out = _params(dev, A, Bb, torch.float32, None, 0.0031)
    op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32_batch(p)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.int8_matmul_nobias_f32(A, Bb, 0.0031, M, N, K, batch=True).view(np.uint32))


def test_bmm_head_loop_as_one_launch(dev, oracle):
    """BMM_S8T_S8N_F32T b=12,(512,512,64) and BMM_S8T_S8N_S8T b=12,(512,64,512) (test_ops.cc:380-410, 444-473)."""
    from tinychatengine_amd.linear import bmm_s8t_s8n
    rng = np.random.default_rng(12)
    for (b, m, n, k, fp32) in [(12, 512, 512, 64, True), (12, 512, 64, 512, False), (3, 7, 5, 48, False)]:
        a = rng.integers(-128, 128, (b, m, k), dtype=np.int8)
        w = rng.integers(-128, 128, (b, n, k), dtype=np.int8)
        out = bmm_s8t_s8n(_t(dev, a), _t(dev, w), 0.0013, fp32).cpu().numpy()
        for h in range(b):
            exp = oracle.int8_matmul_nobias_f32(a[h], w[h], 0.0013, m, n, k) if fp32 else oracle.int8_matmul_nobias_i8(a[h], w[h], 0.0013, -128, 127, m, n, k)
            assert np.array_equal(out[h], exp), f"head {h}"


def test_golden_vectors(dev, golden):
    """The committed outputs of the reference's own code."""
    from tinychatengine_amd.matmul import MatmulOperator
    op = MatmulOperator()
    M, N, K = (int(v) for v in golden["i8_dims"])
    al, be = (float(v) for v in golden["i8_alpha_beta"])
    A, B, b8, bf, Bb = golden["i8_A"], golden["i8_B"], golden["i8_bias8"], golden["i8_biasf"], golden["i8_Bb"]
    runs = [
        ("i8_bias_i8", op.mat_mul_accelerator_int8_fast_2x2_32unroll, B, torch.int8, b8, -128),
        ("i8_bias_i8_relu", op.mat_mul_accelerator_int8_fast_2x2_32unroll, B, torch.int8, b8, 0),
        ("i8_bias_i8_over_column", op.mat_mul_accelerator_int8_fast_32unroll_over_column, B, torch.int8, b8, -128),
        ("i8_nobias_i8", op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias, B, torch.int8, None, -128),
        ("i8_nobias_batch_i8", op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_batch, Bb, torch.int8, None, -128),
        ("i8_bias_f32", op.mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32, B, torch.float32, bf, -128),
        ("i8_nobias_f32", op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32, B, torch.float32, None, -128),
        ("i8_nobias_batch_f32", op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32_batch, Bb, torch.float32, None, -128),
    ]
    for name, fn, Bm, dt, bias, qmin in runs:
        p, out = _params(dev, A, Bm, dt, bias, al, be, qmin)
        fn(p)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint8), golden[name].view(np.uint8)), name
    p, out = _params(dev, golden["i8_tie_A"], golden["i8_tie_B"], torch.int8, None, 0.5)
    op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias(p)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), golden["i8_tie_nobias_i8"]), "round-half-away-from-zero"


def test_linear_wrappers_and_size_independent_properties(dev, oracle):
    """L2 wrappers + properties that hold at any size: ReLU variant == max(non-ReLU, 0); permuting the rows of A
    permutes the rows of C; alpha = 1, beta = 0 on small integers returns the exact integer dot products."""
    from tinychatengine_amd.linear import W8A8B8O8Linear, W8A8BFP32OFP32Linear
    M, N, K = 108, 3072, 768
    A, B, b8, bf = _data(M, N, K, seed=1)
    x, w = _t(dev, A), _t(dev, B)
    y = W8A8B8O8Linear(w, _t(dev, b8), ALPHA, BETA)(x)
    yr = W8A8B8O8Linear(w, _t(dev, b8), ALPHA, BETA, relu=True)(x)
    assert torch.equal(yr, torch.clamp(y, min=0))
    perm = torch.randperm(M, device=dev)
    assert torch.equal(W8A8B8O8Linear(w, _t(dev, b8), ALPHA, BETA)(x[perm].contiguous()), y[perm])
    yf = W8A8BFP32OFP32Linear(w, _t(dev, bf), ALPHA)(x)
    assert np.array_equal(yf.cpu().numpy().view(np.uint32), oracle.int8_matmul_bias_f32(A, B, bf, ALPHA, M, N, K).view(np.uint32))
    small_a = torch.randint(-3, 4, (64, 256), dtype=torch.int8, device=dev)
    small_b = torch.randint(-3, 4, (96, 256), dtype=torch.int8, device=dev)
    exact = W8A8BFP32OFP32Linear(small_b, torch.zeros(96, device=dev), 1.0)(small_a)
    assert torch.equal(exact, (small_a.float() @ small_b.float().t()))
    y1 = W8A8B8O8Linear(w, _t(dev, b8), ALPHA, BETA)(x[:1].contiguous())  # m == 1 -> over_column entry point
    assert torch.equal(y1, y[:1])


def test_unsupported_kind_is_rejected(dev):
    from tinychatengine_amd import capi
    a = torch.zeros(4, 64, dtype=torch.int8, device=dev)
    d = capi.W8A8Desc(M=4, N=4, K=64, batch=1, A=a.data_ptr(), B=a.data_ptr(), bias=a.data_ptr(), C=a.data_ptr(), alpha=1.0, beta=1.0,
                      q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_FP32, out_kind=capi.TCE_OUT_INT8)
    assert capi.w8a8_matmul(d, 0) == capi.TCE_ERR_UNSUPPORTED_KIND
    # the *_batch form (a B per row of A) is a single problem in the reference: with batch > 1 one stride word would have to be both the batch and the row
    # stride of B -- refused, not mis-addressed
    b = torch.zeros(2 * 4 * 4 * 64, dtype=torch.int8, device=dev)
    c = torch.zeros(2 * 4 * 4, dtype=torch.int8, device=dev)
    d = capi.W8A8Desc(M=4, N=4, K=64, batch=2, A=a.data_ptr(), B=b.data_ptr(), C=c.data_ptr(), strideA=0, strideB=4 * 4 * 64, strideC=16, alpha=1.0, beta=0.0,
                      q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE, out_kind=capi.TCE_OUT_INT8, b_per_row=1)
    assert capi.w8a8_matmul(d, 0) == capi.TCE_ERR_UNSUPPORTED_KIND


@pytest.mark.parametrize("m,n", [(108, 768), (1, 768), (512, 768), (65, 1024), (3, 2048), (7, 20), (5, 772), (2, 36), (3, 60), (1, 8192), (4, 4),
                                 (1, 4096), (256, 1024), (257, 1024), (6, 4096), (2, 260), (1, 2080), (300, 256)])
def test_layernorm_q_bit_exact(dev, oracle, m, n):
    """tce_layernorm_q against the oracle's restatement of LayerNormQ::forward (LayerNormQ.cc:12-52): bit for bit -- the wave-per-row form (many rows, short
    rows) and the forms that walk a row's sums with 4 or 16 waves at once (up to 256 rows of 256 / 1024 columns or more; sequential_sum_speculated), with rows
    whose running sums return to zero or cancel (the speculation's misses) among them."""
    import ctypes as C
    from tinychatengine_amd import capi
    rng = np.random.default_rng(m * 1000 + n)
    x = (rng.standard_normal((m, n)) * 3 + rng.standard_normal((m, 1))).astype(np.float32)
    if n >= 256:
        x[0] = rng.standard_normal(n).astype(np.float32)              # zero-mean noise
        r = x[m - 1]
        for nw in (4, 16):                                            # every segment of either form sums to (almost) nothing
            seg = ((n + nw * 32 - 1) // (nw * 32)) * 32
            for s0 in range(0, n, seg):
                e = min(s0 + seg, n)
                r[e - 1] = np.float32(-np.sum(r[s0:e - 1], dtype=np.float64))
        if m > 2:
            x[1] *= np.float32(1e-3)
            x[1, 3], x[1, n // 2 + 5] = 1.0e7, -1.0e7                 # cancellation of huge values
    w = (8 + 4 * rng.standard_normal(n)).astype(np.float32)
    b = (rng.standard_normal(n) * 2).astype(np.float32)
    out = torch.zeros((m, n), dtype=torch.int8, device=dev)
    tx, tw, tb = _t(dev, x), _t(dev, w), _t(dev, b)
    capi.check(capi.lib().tce_layernorm_q(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), out.data_ptr(), m, n, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    want = oracle.layernorm_q(x, w, b)
    got = out.cpu().numpy()
    assert np.array_equal(got, want), f"{(got != want).sum()} of {got.size} int8 outputs differ"


@pytest.mark.parametrize("M,N,K", [(65, 130, 208), (33, 65, 128), (512, 768, 768), (70, 33, 1136), (16, 16, 64)])
def test_wave_quartets_per_tile_are_bit_exact(dev, oracle, M, N, K):
    """The K range of a 64x64 tile cut between 1 / 2 / 4 wave quartets (debug modes 71 / 72 / 74) and the automatic choice: int32 partial
    tiles are exact, so every setting must give the oracle's bytes -- odd step counts, a K tail (K % 64 != 0), fewer steps than quartets."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.matmul import MatmulOperator
    op = MatmulOperator()
    L = capi.lib()
    A, B, b8, _ = _data(M, N, K, seed=K + M)
    exp = oracle.int8_matmul_bias_i8(A, B, b8, ALPHA, BETA, -128, 127, M, N, K)
    try:
        for mode in (71, 72, 74, 70):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            p, out = _params(dev, A, B, torch.int8, b8)
            op.mat_mul_accelerator_int8_fast_2x2_32unroll(p)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), exp), f"mode {mode}: {(out.cpu().numpy() != exp).sum()} mismatches"
    finally:
        L.tce_w4a16_set_debug_mode(70)


@pytest.mark.parametrize("M,N,K", [(512, 768, 3072), (108, 768, 3072), (65, 130, 208), (33, 65, 128), (70, 33, 1136), (40, 16, 64), (300, 768, 1552)])
def test_32_row_tiles_are_bit_exact(dev, oracle, M, N, K):
    """Round 6: the 32 x 64 tile (w8a8_mfma_kernel<KS, false, 1>: each wave 16 rows x 32 columns) that the rule takes where 64 x 64 tiles would leave most CUs without a
    workgroup (512 x 768 x 3072: 96 -> 192 workgroups) -- forced on (debug mode 191) with 1 / 2 / 4 quartets per tile and under the rule, int8 and fp32 outputs, ragged last
    row / column tiles, a K tail: the oracle's bytes every time."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.matmul import MatmulOperator
    op = MatmulOperator()
    L = capi.lib()
    A, B, b8, bf = _data(M, N, K, seed=K + M + 7)
    exp = oracle.int8_matmul_bias_i8(A, B, b8, ALPHA, BETA, -128, 127, M, N, K)
    exp32 = oracle.int8_matmul_bias_f32(A, B, bf, ALPHA, M, N, K)
    try:
        capi.check(L.tce_w4a16_set_debug_mode(191))
        for mode in (71, 72, 74, 70):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            p, out = _params(dev, A, B, torch.int8, b8)
            op.mat_mul_accelerator_int8_fast_2x2_32unroll(p)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), exp), f"mode {mode}: {(out.cpu().numpy() != exp).sum()} mismatches"
            p, out = _params(dev, A, B, torch.float32, bf)
            op.mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32(p)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint32), exp32.view(np.uint32)), f"mode {mode}, fp32 output"
        capi.check(L.tce_w4a16_set_debug_mode(190))  # the rule
        p, out = _params(dev, A, B, torch.int8, b8)
        op.mat_mul_accelerator_int8_fast_2x2_32unroll(p)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), exp)
    finally:
        L.tce_w4a16_set_debug_mode(70)
        L.tce_w4a16_set_debug_mode(190)


KSLICE_FORMS = (304, 404, 904)


@pytest.mark.parametrize("M,N,K", [(512, 768, 3072), (108, 768, 3072), (65, 130, 208), (33, 65, 128), (70, 33, 1136), (40, 16, 64), (300, 768, 1552), (512, 768, 768), (97, 200, 16)])
def test_whole_tile_per_wave_is_bit_exact(dev, oracle, M, N, K):
    """Round 6: w8a8_kslice_kernel -- every wave of the workgroup contracts the WHOLE (32 | 64) x (48 | 64) tile on its own run of k-steps (each operand byte through the CU's
    L1 once per workgroup), the waves' int32 tiles added through LDS -- every form forced (debug mode 19000 + form) and the rule: int8 and fp32 outputs, ragged last row /
    column tiles, a K tail, fewer k-steps than waves (waves without a step), K < 64 (the tail only): the oracle's bytes every time."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.matmul import MatmulOperator
    op = MatmulOperator()
    L = capi.lib()
    A, B, b8, bf = _data(M, N, K, seed=K + M + 11)
    exp = oracle.int8_matmul_bias_i8(A, B, b8, ALPHA, BETA, -128, 127, M, N, K)
    exp32 = oracle.int8_matmul_bias_f32(A, B, bf, ALPHA, M, N, K)
    try:
        for form in KSLICE_FORMS + (0,):
            capi.check(L.tce_w4a16_set_debug_mode(19000 + form))
            p, out = _params(dev, A, B, torch.int8, b8)
            op.mat_mul_accelerator_int8_fast_2x2_32unroll(p)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), exp), f"form {form}: {(out.cpu().numpy() != exp).sum()} mismatches"
            p, out = _params(dev, A, B, torch.float32, bf)
            op.mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32(p)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint32), exp32.view(np.uint32)), f"form {form}, fp32 output"
    finally:
        L.tce_w4a16_set_debug_mode(19000)


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (65, 100, 1024), (512, 768, 3072), (108, 2048, 8192), (16, 136, 1600), (9, 64, 192), (130, 70, 4160)])
def test_deep_pipeline_tile_is_bit_exact(dev, oracle, M, N, K):
    """The 64x64 tile with eight k-steps in flight (w8a8_mfma_deep_kernel; the rule takes it for chains of 64+ steps on few tiles) forced with 1 / 2 / 4 wave quartets
    (tce_w8a8_set_tuning(0, 0, 1 / 2 / 4)): cooperative panels through a double-buffered LDS stage, requests past the chain clamped -- fewer steps than the ring (1, 3), step
    counts that are no multiple of the ring or of the quartets (25, 65), ragged M and N; the oracle's bytes every time."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.matmul import MatmulOperator
    op = MatmulOperator()
    L = capi.lib()
    A, B, b8, _ = _data(M, N, K, seed=3 * K + M)
    exp = oracle.int8_matmul_bias_i8(A, B, b8, ALPHA, BETA, -128, 127, M, N, K)
    try:
        for quartets in (1, 2, 4, 0):
            capi.w8a8_set_tuning(deep_pipeline=quartets)
            p, out = _params(dev, A, B, torch.int8, b8)
            op.mat_mul_accelerator_int8_fast_2x2_32unroll(p)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), exp), f"{quartets} quartets: {(out.cpu().numpy() != exp).sum()} mismatches"
    finally:
        capi.w8a8_set_tuning()


@pytest.mark.parametrize("m,k,ns", [(1, 768, (768, 768, 768)), (1, 768, (3072,)), (3, 2048, (2048, 512)), (8, 96, (40, 24, 16, 8)), (1, 8192, (64,)),
                                     (1, 4096, (16384,)), (2, 1024, (1000, 24, 3000)), (1, 4096, (4096, 4096, 4096))])  # k >= 1024: the weights-resident form (OPT-6.7B: fc1, q / k / v)
def test_layernorm_q_fused_with_its_linears_is_bit_exact(dev, oracle, m, k, ns):
    """tce_layernorm_q_w8a8_group: LayerNormQ::forward (LayerNormQ.cc:12-52) + q/k/v (Int8OPTAttention.cc:186-201) or
    fc1 as ONE launch, against the oracle's LayerNormQ followed by int8_ref_matmul, and against the library's separate launches."""
    from tinychatengine_amd.linear import W8A8B8O8Linear, W8A8BFP32OFP32Linear, layernorm_q_linears
    rng = np.random.default_rng(m * 7 + k + len(ns))
    x = (rng.standard_normal((m, k)) * 3.0 + 0.5).astype(np.float32)
    x[0, :8] = [0.5, 1.5, 2.5, -0.5, -1.5, 100.0, -100.0, 0.0]  # ties of the final rounding and saturating values
    lw = (rng.standard_normal(k) * 20.0).astype(np.float32)
    lb = (rng.standard_normal(k) * 5.0).astype(np.float32)
    q_ref = oracle.layernorm_q(x, lw, lb)
    lins, exps = [], []
    for i, n in enumerate(ns):
        W = rng.integers(-128, 128, (n, k), dtype=np.int8)
        if i == 1:  # one fp32-out linear in the group (the out_proj / fc2 kind), the rest int8-out, the last one with ReLU clamping
            b = rng.standard_normal(n).astype(np.float32)
            lins.append(W8A8BFP32OFP32Linear(_t(dev, W), _t(dev, b), 0.0031))
            exps.append(oracle.int8_matmul_bias_f32(q_ref, W, b, 0.0031, m, n, k))
        else:
            b = rng.integers(-128, 128, n, dtype=np.int8)
            relu = i == len(ns) - 1 and len(ns) > 1
            lins.append(W8A8B8O8Linear(_t(dev, W), _t(dev, b), ALPHA, BETA, relu=relu))
            exps.append(oracle.int8_matmul_bias_i8(q_ref, W, b, ALPHA, BETA, 0 if relu else -128, 127, m, n, k))
    xt, lwt, lbt = _t(dev, x), _t(dev, lw), _t(dev, lb)
    ln_out = torch.zeros((m, k), dtype=torch.int8, device=dev)
    outs = layernorm_q_linears(xt, lwt, lbt, lins, ln_out=ln_out)
    torch.cuda.synchronize()
    assert np.array_equal(ln_out.cpu().numpy(), q_ref), "fused LayerNormQ output differs from the reference's"
    for o, e in zip(outs, exps):
        got = o.cpu().numpy()
        assert np.array_equal(got.view(np.uint8 if got.dtype == np.int8 else np.uint32), e.view(np.uint8 if e.dtype == np.int8 else np.uint32))
    # and the unfused path of this library
    from tinychatengine_amd import capi
    q_sep = torch.zeros((m, k), dtype=torch.int8, device=dev)
    capi.check(capi.lib().tce_layernorm_q(xt.data_ptr(), lwt.data_ptr(), lbt.data_ptr(), q_sep.data_ptr(), m, k, torch.cuda.current_stream().cuda_stream))
    for lin, o in zip(lins, outs):
        assert torch.equal(lin(q_sep), o)


@pytest.mark.parametrize("m,k", [(1, 4096), (1, 2048), (2, 1040), (1, 2064), (3, 1024), (1, 8192)])
def test_layernorm_q_long_rows_when_the_speculated_sums_miss(dev, oracle, m, k):
    """The wide form of tce_layernorm_q_w8a8_group (k >= 1024) walks a row's two sequential fp32 sums with 16 waves at once, each starting its segment from 64
    candidate running values around the exactly rounded prefix (sequential_sum_speculated, tce_common.hpp); where no candidate is the true running value the
    segment is re-added.  Rows built to miss: running sums that return to (almost) zero at the segment boundaries, cancellation of huge values (the running value
    loses its low bits, then regains them), zero-mean noise, a constant row (all deviations zero), tiny values after large ones -- against the oracle, bit for bit."""
    from tinychatengine_amd.linear import W8A8B8O8Linear, layernorm_q_linears
    rng = np.random.default_rng(m * 13 + k)
    seg = ((k + 16 * 32 - 1) // (16 * 32)) * 32
    rows = []
    base = rng.standard_normal(k).astype(np.float32)
    r0 = base.copy()                                      # zero-mean noise
    r1 = base.copy()                                      # every segment sums to (almost) nothing: the running value sits near zero at each boundary
    for b in range(0, k, seg):
        e = min(b + seg, k)
        r1[e - 1] = np.float32(-np.sum(r1[b:e - 1], dtype=np.float64))
    r2 = base.copy() * np.float32(1e-3)                   # cancellation: +1e7 early, -1e7 late, small values between and after
    r2[3], r2[k // 2 + 5] = 1.0e7, -1.0e7
    r3 = np.full(k, 0.7, np.float32)                      # a constant row: every squared deviation is the same tiny number (or zero)
    r4 = np.concatenate([base[: k // 2] * np.float32(3e4), base[k // 2:] * np.float32(1e-4)]).astype(np.float32)  # large, then tiny
    r5 = np.abs(base) + np.float32(0.25)                  # all positive (a steadily growing sum)
    pool = [r0, r1, r2, r3, r4, r5]
    lw = (rng.standard_normal(k) * 20.0).astype(np.float32)
    lb = (rng.standard_normal(k) * 5.0).astype(np.float32)
    W = rng.integers(-128, 128, (48, k), dtype=np.int8)
    b8 = rng.integers(-128, 128, 48, dtype=np.int8)
    lin = W8A8B8O8Linear(_t(dev, W), _t(dev, b8), ALPHA, BETA)
    lwt, lbt = _t(dev, lw), _t(dev, lb)
    for start in range(0, len(pool), m):
        x = np.stack([pool[(start + i) % len(pool)] for i in range(m)]).astype(np.float32)
        q_ref = oracle.layernorm_q(x, lw, lb)
        xt = _t(dev, x)
        ln_out = torch.zeros((m, k), dtype=torch.int8, device=dev)
        (out,) = layernorm_q_linears(xt, lwt, lbt, [lin], ln_out=ln_out)
        torch.cuda.synchronize()
        assert np.array_equal(ln_out.cpu().numpy(), q_ref), f"rows {start}..: {int((ln_out.cpu().numpy() != q_ref).sum())} normalised values differ"
        assert np.array_equal(out.cpu().numpy(), oracle.int8_matmul_bias_i8(q_ref, W, b8, ALPHA, BETA, -128, 127, m, 48, k))


def test_layernorm_q_fused_argument_checks(dev):
    from tinychatengine_amd import capi
    x = torch.zeros(9, 64, device=dev)
    w = torch.zeros(64, device=dev)
    B = torch.zeros(16, 64, dtype=torch.int8, device=dev)
    o = torch.zeros(9, 16, dtype=torch.int8, device=dev)
    d = capi.W8A8Desc(M=9, N=16, K=64, batch=1, B=B.data_ptr(), C=o.data_ptr(), alpha=1.0, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE, out_kind=capi.TCE_OUT_INT8)
    arr = (capi.W8A8Desc * 1)(d)
    L = capi.lib()
    assert L.tce_layernorm_q_w8a8_group(x.data_ptr(), w.data_ptr(), w.data_ptr(), 9, 64, arr, 1, None, None) == capi.TCE_ERR_UNSUPPORTED_SHAPE  # m > 8
    arr[0].M = 2
    assert L.tce_layernorm_q_w8a8_group(x.data_ptr(), w.data_ptr(), w.data_ptr(), 1, 64, arr, 1, None, None) == capi.TCE_ERR_BAD_ARG  # M mismatch
    assert L.tce_layernorm_q_w8a8_group(x.data_ptr(), w.data_ptr(), w.data_ptr(), 1, 40, arr, 1, None, None) == capi.TCE_ERR_UNSUPPORTED_SHAPE  # k % 16


@pytest.mark.parametrize("embed,heads,ffn,steps", [(256, 4, 512, [(0, 5), (5, 1), (6, 1), (7, 3)]),      # a prompt of 5, two decode tokens, a batch of 3
                                                   (768, 12, 3072, [(0, 40), (40, 1)]),                   # OPT-125M's sizes (model.h:70): prefill 40, one decode token
                                                   (256, 4, 512, [(0, 1), (1, 1), (2, 20)]),
                                                   (512, 4, 1024, [(0, 7), (7, 1), (8, 2), (10, 1)])])                    # heads of 128 (OPT-6.7B's head dimension)
def test_opt_decoder_layer_against_the_oracle_composition(dev, oracle, embed, heads, ffn, steps):
    """A whole SmoothQuant OPT decoder layer on this library's launches (tinychatengine_amd/opt_layer.py: 8 per decode step, 12 per prefill) against
    the ORACLE'S composition of Int8OPTDecoderLayer::forward (Int8OPTDecoderLayer.cc:24-59, Int8OPTAttention.cc:183-284): LayerNormQ, the three
    projections, the KV append, the per-head qk BMM, batch_Add + softmax + int8 conversion, the per-head pv BMM, out_proj + add, LayerNormQ,
    fc1 (ReLU), fc2 + add -- token after token with a growing cache.  Every stage is compared BIT FOR BIT (q / k / v, the scores, the attention
    rows, fc1's output, the residual stream); the int8 probabilities may differ where the device's expf and the host's disagree in the last bit
    (counted: < 2e-4 of them, by one step), and the stages behind them are then checked against the oracle fed with the device's probabilities."""
    from tinychatengine_amd.opt_layer import Int8OPTDecoderLayer
    hd, max_keys, max_rows = embed // heads, 64, 40
    layer = Int8OPTDecoderLayer(embed, heads, ffn, max_keys, max_rows, dev, seed=embed + ffn)
    layer.fused_attention = False  # the separately issued steps are the ones held to the oracle stage by stage ...
    fused_layer = Int8OPTDecoderLayer(embed, heads, ffn, max_keys, max_rows, dev, seed=embed + ffn)  # ... and tce_opt_attention_decode to THEM, bit for bit
    assert fused_layer.fused_attention
    P = {k: getattr(layer, k).cpu().numpy() for k in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "Wq", "Wk", "Wv", "bq", "bk", "bv", "Wo", "bo", "W1", "b1", "W2", "b2")}
    rng = np.random.default_rng(embed + len(steps))
    Kc = np.zeros((heads, 0, hd), np.int8)
    Vc = np.zeros((heads, 0, hd), np.int8)
    for (pos, m) in steps:
        tgz = pos + m
        hidden0 = (rng.standard_normal((m, embed)) * 2).astype(np.float32)
        mask = np.zeros((m, tgz), np.float32)
        for j in range(m):  # causal: row j sees keys 0 .. pos + j
            mask[j, pos + j + 1:] = np.finfo(np.float32).min
        h_gpu = torch.from_numpy(hidden0.copy()).to(dev)
        t_mask = torch.from_numpy(mask).to(dev)
        layer.step(h_gpu, pos, t_mask)
        h_fused = torch.from_numpy(hidden0.copy()).to(dev)
        fused_layer.step(h_fused, pos, t_mask)
        torch.cuda.synchronize()
        assert torch.equal(fused_layer.attn[:m], layer.attn[:m]) and torch.equal(fused_layer.k_cache, layer.k_cache) and torch.equal(fused_layer.vt_cache[:, :, :tgz], layer.vt_cache[:, :, :tgz]), \
            f"step {(pos, m)}: the one-launch attention differs from the four launches"
        assert torch.equal(h_fused.view(torch.int32), h_gpu.view(torch.int32)), f"step {(pos, m)}: residual stream, fused attention"
        assert fused_layer.launches(m) == (5 if m <= 8 else 12) and layer.launches(m) == (8 if m <= 8 else 12)
        # ---- the oracle's composition ----
        ln = oracle.layernorm_q(hidden0, P["ln1_w"], P["ln1_b"]).reshape(m, embed)
        q, k, v = (oracle.int8_matmul_bias_i8(ln, P["W" + n], P["b" + n], layer.a_qkv, layer.b_qkv, -128, 127, m, embed, embed) for n in "qkv")
        for name, want in (("q", q), ("k", k), ("v", v)):
            assert np.array_equal(getattr(layer, name)[:m].cpu().numpy(), want), f"step {(pos, m)}: {name} projection"
        Kc = np.concatenate([Kc, k.reshape(m, heads, hd).transpose(1, 0, 2)], axis=1)
        Vc = np.concatenate([Vc, v.reshape(m, heads, hd).transpose(1, 0, 2)], axis=1)
        assert np.array_equal(layer.k_cache[:, :tgz].cpu().numpy(), Kc) and np.array_equal(layer.vt_cache[:, :, :tgz].cpu().numpy(), Vc.transpose(0, 2, 1)), "KV append"
        qh = q.reshape(m, heads, hd).transpose(1, 0, 2)
        scores = np.stack([oracle.int8_matmul_nobias_f32(qh[h], Kc[h], layer.a_qk, m, tgz, hd) for h in range(heads)])
        got_scores = layer.scores.view(-1)[: heads * m * tgz].view(heads, m, tgz).cpu().numpy()
        assert np.array_equal(got_scores.view(np.uint32), scores.view(np.uint32)), f"step {(pos, m)}: qk BMM"
        probs = oracle.opt_softmax_q(scores, mask)
        ldp = (tgz + 15) // 16 * 16
        got_probs = layer.probs.view(-1)[: heads * m * ldp].view(heads, m, ldp)[:, :, :tgz].cpu().numpy()
        diff = np.abs(got_probs.astype(np.int32) - probs.astype(np.int32))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-4, f"step {(pos, m)}: int8 probabilities: max step {diff.max()}, {float((diff > 0).mean()):.2e} differ"
        attn = np.concatenate([oracle.int8_matmul_nobias_i8(got_probs[h], np.ascontiguousarray(Vc[h].T), layer.a_pv, -128, 127, m, hd, tgz) for h in range(heads)], axis=1)
        assert np.array_equal(layer.attn[:m].cpu().numpy(), attn), f"step {(pos, m)}: pv BMM / unshape"
        h1 = hidden0 + oracle.int8_matmul_bias_f32(attn, P["Wo"], P["bo"], layer.a_o, m, embed, embed)  # fp32 add: one rounding (Int8OPTDecoderLayer.cc:14-22)
        ln2 = oracle.layernorm_q(h1, P["ln2_w"], P["ln2_b"]).reshape(m, embed)
        f1 = oracle.int8_matmul_bias_i8(ln2, P["W1"], P["b1"], layer.a_1, layer.b_1, 0, 127, m, ffn, embed)
        assert np.array_equal(layer.fc1[:m].cpu().numpy(), f1), f"step {(pos, m)}: final_layer_norm + fc1"
        want = h1 + oracle.int8_matmul_bias_f32(f1, P["W2"], P["b2"], layer.a_2, m, embed, ffn)
        assert np.array_equal(h_gpu.cpu().numpy().view(np.uint32), want.view(np.uint32)), f"step {(pos, m)}: the residual stream"


@pytest.mark.parametrize("M,N,K,mode,lda,ldb,ldc,batch", [(512, 4096, 1024, 75, 0, 0, 0, 1),      # the rule: 128 x 128 tiles (4 x 32 x ... >= 224 is not met: 128 x 64)
                                                          (1024, 4096, 512, 75, 0, 0, 0, 1),      # the rule: 128 x 128 tiles
                                                          (300, 2100, 1024, 76, 0, 0, 0, 1),      # ragged in M and N, 128 columns forced
                                                          (300, 2100, 1024, 77, 0, 0, 0, 1),      # the same with 64 columns
                                                          (129, 65, 256, 76, 0, 0, 0, 1), (129, 65, 256, 77, 0, 0, 0, 1),
                                                          (128, 128, 4096, 77, 0, 0, 0, 1),       # one tile, 64 k-steps
                                                          (200, 260, 320, 76, 384, 448, 300, 2),  # leading dimensions, a batch of two
                                                          (200, 260, 320, 77, 384, 448, 300, 2),
                                                          # two wave quartets per tile, each on half of K (176 / 177: 128 / 64 columns), odd numbers of k-steps
                                                          (129, 65, 256, 176, 0, 0, 0, 1), (300, 2100, 1024, 177, 0, 0, 0, 1), (200, 260, 320, 176, 384, 448, 300, 2),
                                                          (512, 4096, 4096, 75, 0, 0, 0, 1), (130, 200, 576, 177, 0, 0, 0, 1),
                                                          # round 6, the whole tile in every wave (19304 / 19404 / 19904) and the staged int8 stores under leading dimensions, a
                                                          # batch, `accumulate`: ldc a multiple of 16 (row pieces of 16 bytes) and not (byte stores), N ragged against the 48 / 64-wide tiles
                                                          (200, 260, 832, 19304, 896, 960, 304, 2), (200, 260, 832, 19404, 896, 960, 300, 2), (200, 260, 832, 19904, 896, 960, 304, 3),
                                                          (70, 192, 1024, 19304, 0, 0, 208, 1), (70, 192, 1024, 19404, 0, 0, 0, 2), (40, 64, 1024, 19000, 1040, 0, 768, 12),
                                                          # a row pitch no 16-byte piece fits (301: both output kinds take the one-element stores), every family
                                                          (200, 260, 832, 19304, 896, 960, 301, 2), (200, 260, 832, 19001, 896, 960, 301, 2), (300, 2100, 1024, 76, 0, 0, 2101, 1),
                                                          # the 64 x 64 quartet kernel's staged stores (19001: the k-slice forms off)
                                                          (200, 260, 832, 19001, 896, 960, 304, 2), (512, 768, 768, 19001, 0, 0, 0, 1)])
def test_w8a8_large_tiles_bit_exact(dev, oracle, M, N, K, mode, lda, ldb, ldc, batch):
    """The prefill-sized int8 kernel (w8a8_mfma_big_kernel: 128 x 128 / 128 x 64 tiles, operand panels through a three-stage LDS ring) against the oracle: the
    int8-out form with an int8 bias (clamp at 0: the ReLU linear) and the fp32-out form with an fp32 bias accumulating into C -- every element, ragged edges,
    leading dimensions and batches, the -128 corner rows; and the form with two wave quartets per tile (each on half of K, int32 tiles added through LDS)."""
    from tinychatengine_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(M + N + K + mode)
    lda_, ldb_, ldc_ = lda or K, ldb or K, ldc or N
    A = rng.integers(-128, 128, (batch, M, lda_), dtype=np.int8)
    B = rng.integers(-128, 128, (batch, N, ldb_), dtype=np.int8)
    A[:, 0, :] = -128
    B[:, 0, :] = -128
    b8 = rng.integers(-128, 128, N, dtype=np.int8)
    bf = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((batch, M, ldc_)).astype(np.float32)
    tA, tB, tb8, tbf = _t(dev, A), _t(dev, B), _t(dev, b8), _t(dev, bf)
    st = torch.cuda.current_stream().cuda_stream
    capi.check(L.tce_w4a16_set_debug_mode(mode))
    try:
        out8 = torch.full((batch, M, ldc_), 77, dtype=torch.int8, device=dev)
        d = capi.W8A8Desc(M=M, N=N, K=K, batch=batch, A=tA.data_ptr(), B=tB.data_ptr(), bias=tb8.data_ptr(), C=out8.data_ptr(), alpha=ALPHA, beta=BETA, q_min=0, q_max=127,
                          bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8, lda=lda, ldb=ldb, ldc=ldc, strideA=M * lda_, strideB=N * ldb_, strideC=M * ldc_)
        capi.check(capi.w8a8_matmul(d, st))
        tC = _t(dev, C0.copy())
        d = capi.W8A8Desc(M=M, N=N, K=K, batch=batch, A=tA.data_ptr(), B=tB.data_ptr(), bias=tbf.data_ptr(), C=tC.data_ptr(), alpha=0.0071, q_min=-128, q_max=127,
                          bias_kind=capi.TCE_BIAS_FP32, out_kind=capi.TCE_OUT_FP32, accumulate=1, lda=lda, ldb=ldb, ldc=ldc, strideA=M * lda_, strideB=N * ldb_, strideC=M * ldc_)
        capi.check(capi.w8a8_matmul(d, st))
        torch.cuda.synchronize()
    finally:
        capi.check(L.tce_w4a16_set_debug_mode(75))
        capi.check(L.tce_w4a16_set_debug_mode(19000))
    got8, gotf = out8.cpu().numpy(), tC.cpu().numpy()
    for bi in range(batch):
        Ab, Bb = np.ascontiguousarray(A[bi, :, :K]), np.ascontiguousarray(B[bi, :, :K])
        assert np.array_equal(got8[bi, :, :N], oracle.int8_matmul_bias_i8(Ab, Bb, b8, ALPHA, BETA, 0, 127, M, N, K)), f"int8 out, batch {bi}"
        assert (got8[bi, :, N:] == 77).all(), "columns behind N were written"
        want = C0[bi, :, :N] + oracle.int8_matmul_bias_f32(Ab, Bb, bf, 0.0071, M, N, K)
        assert np.array_equal(gotf[bi, :, :N].view(np.uint32), want.view(np.uint32)), f"fp32 out + accumulate, batch {bi}"
        assert np.array_equal(gotf[bi, :, N:], C0[bi, :, N:])


def test_w8a8_leading_dimensions_and_accumulate(dev, oracle):
    """tce_w8a8_desc.lda / ldb / ldc (a head's 64 columns of a wider matrix as an operand as it lies) and `accumulate` (the fp32 residual add in
    the launch), MFMA and generic kernels, against the oracle on the gathered operands."""
    from tinychatengine_amd import capi
    import ctypes as C
    rng = np.random.default_rng(8)
    for (M, N, K, lda, ldb, ldc) in ((70, 96, 64, 320, 128, 200), (5, 40, 48, 64, 48, 40), (130, 64, 192, 192, 256, 64)):
        A = rng.integers(-128, 128, (M, lda), dtype=np.int8)
        B = rng.integers(-128, 128, (N, ldb), dtype=np.int8)
        bias = rng.standard_normal(N).astype(np.float32)
        C0 = rng.standard_normal((M, ldc)).astype(np.float32)
        tA, tB, tb, tC = (torch.from_numpy(x).to(dev) for x in (A, B, bias, C0.copy()))
        d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=tA.data_ptr(), B=tB.data_ptr(), bias=tb.data_ptr(), C=tC.data_ptr(), alpha=0.001, q_min=-128, q_max=127,
                          bias_kind=capi.TCE_BIAS_FP32, out_kind=capi.TCE_OUT_FP32, accumulate=1, lda=lda, ldb=ldb, ldc=ldc)
        capi.check(capi.w8a8_matmul(d, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        want = C0.copy()
        want[:, :N] = C0[:, :N] + oracle.int8_matmul_bias_f32(np.ascontiguousarray(A[:, :K]), np.ascontiguousarray(B[:, :K]), bias, 0.001, M, N, K)
        assert np.array_equal(tC.cpu().numpy().view(np.uint32), want.view(np.uint32)), (M, N, K)
    bad = capi.W8A8Desc(M=4, N=4, K=16, batch=1, A=tA.data_ptr(), B=tB.data_ptr(), C=tC.data_ptr(), alpha=1.0, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE,
                        out_kind=capi.TCE_OUT_INT8, accumulate=1)
    assert capi.w8a8_matmul(bad, None) == capi.TCE_ERR_UNSUPPORTED_KIND
    bad = capi.W8A8Desc(M=4, N=4, K=16, batch=1, A=tA.data_ptr(), B=tB.data_ptr(), C=tC.data_ptr(), alpha=1.0, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE,
                        out_kind=capi.TCE_OUT_FP32, lda=8)
    assert capi.w8a8_matmul(bad, None) == capi.TCE_ERR_BAD_ARG


@pytest.mark.parametrize("M,N,K,lda,ldc,batch", [(1, 768, 3072, 0, 0, 1), (1, 768, 768, 0, 0, 1), (8, 100, 8192, 0, 0, 1), (2, 19, 64, 128, 40, 1), (3, 130, 272, 288, 0, 3),
                                                 (8, 64, 8208, 0, 0, 1), (5, 3, 80, 96, 8, 2)])
def test_decode_sized_wave_per_column_kernel_bit_exact(dev, oracle, M, N, K, lda, ldc, batch):
    """M <= 8 rows: one wave per output column (w8a8_rowdot_kernel<0>), against the oracle and against the MFMA / generic kernels it replaces (debug mode
    73), every epilogue: int8 with bias and ReLU floor, fp32 with bias and `accumulate`; leading dimensions, a batch, an N that is not a multiple of the
    workgroup's four columns, M * K on both sides of the 64 KiB LDS bound (8 x 8208 stays on the MFMA kernel)."""
    from tinychatengine_amd import capi
    rng = np.random.default_rng(M * 1000 + N + K)
    la, lc = lda or K, ldc or N
    A = rng.integers(-128, 128, (batch, M, la), dtype=np.int8)
    B = rng.integers(-128, 128, (batch, N, K), dtype=np.int8)
    b8 = rng.integers(-128, 128, N, dtype=np.int8)
    bf = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((batch, M, lc)).astype(np.float32)
    tA, tB, tb8, tbf = _t(dev, A), _t(dev, B), _t(dev, b8), _t(dev, bf)
    st = torch.cuda.current_stream().cuda_stream
    alpha = 0.0007 if K > 1000 else 0.004
    for mode in (70, 73):
        capi.check(capi.lib().tce_w4a16_set_debug_mode(mode))
        try:
            o8 = torch.zeros((batch, M, lc), dtype=torch.int8, device=dev)
            d = capi.W8A8Desc(M=M, N=N, K=K, batch=batch, A=tA.data_ptr(), B=tB.data_ptr(), bias=tb8.data_ptr(), C=o8.data_ptr(), strideA=M * la, strideB=N * K, strideC=M * lc,
                              alpha=alpha, beta=0.02, q_min=0, q_max=127, bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8, lda=lda, ldc=ldc)
            capi.check(capi.w8a8_matmul(d, st))
            of = _t(dev, C0.copy())
            d = capi.W8A8Desc(M=M, N=N, K=K, batch=batch, A=tA.data_ptr(), B=tB.data_ptr(), bias=tbf.data_ptr(), C=of.data_ptr(), strideA=M * la, strideB=N * K, strideC=M * lc,
                              alpha=alpha, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_FP32, out_kind=capi.TCE_OUT_FP32, lda=lda, ldc=ldc, accumulate=1)
            capi.check(capi.w8a8_matmul(d, st))
            torch.cuda.synchronize()
        finally:
            capi.lib().tce_w4a16_set_debug_mode(70)
        for h in range(batch):
            Ah = np.ascontiguousarray(A[h, :, :K])
            want8 = oracle.int8_matmul_bias_i8(Ah, B[h], b8, alpha, 0.02, 0, 127, M, N, K)
            assert np.array_equal(o8[h, :, :N].cpu().numpy(), want8), (mode, h, "int8")
            wantf = C0[h, :, :N] + oracle.int8_matmul_bias_f32(Ah, B[h], bf, alpha, M, N, K)
            assert np.array_equal(of[h, :, :N].cpu().numpy().view(np.uint32), wantf.view(np.uint32)), (mode, h, "fp32 accumulate")
            if lc > N:  # nothing outside the N columns was touched
                assert np.array_equal(of[h, :, N:].cpu().numpy(), C0[h, :, N:]) and not o8[h, :, N:].any()


@pytest.mark.parametrize("M,N,K,lda,ldb", [(12, 64, 512, 512, 512), (12, 64, 400, 400, 2048), (3, 5, 272, 288, 320), (32, 128, 4096, 0, 0)])
def test_per_row_operand_with_long_rows_bit_exact(dev, oracle, M, N, K, lda, ldb):
    """The *_batch form (row m of A meets its own B_m; BMM_S8T_S8N_S8T.cc:45-52) with K >= 256: one wave per output element (w8a8_rowdot_kernel<1>) --
    the probabilities x V^T product of a decode step, V^T rows `max_keys` apart -- against the oracle and the one-thread-per-output kernel (mode 73)."""
    from tinychatengine_amd import capi
    rng = np.random.default_rng(M + N + K)
    la, lb = lda or K, ldb or K
    A = rng.integers(-128, 128, (M, la), dtype=np.int8)
    B = rng.integers(-128, 128, (M, N, lb), dtype=np.int8)
    tA, tB = _t(dev, A), _t(dev, B)
    st = torch.cuda.current_stream().cuda_stream
    Bc = np.ascontiguousarray(B[:, :, :K])
    want8 = oracle.int8_matmul_nobias_i8(np.ascontiguousarray(A[:, :K]), Bc, 0.0009, -128, 127, M, N, K, batch=True)
    wantf = oracle.int8_matmul_nobias_f32(np.ascontiguousarray(A[:, :K]), Bc, 0.0009, M, N, K, batch=True)
    for mode in (70, 73):
        capi.check(capi.lib().tce_w4a16_set_debug_mode(mode))
        try:
            o8 = torch.zeros((M, N), dtype=torch.int8, device=dev)
            d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=tA.data_ptr(), B=tB.data_ptr(), C=o8.data_ptr(), alpha=0.0009, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE,
                              out_kind=capi.TCE_OUT_INT8, b_per_row=1, strideB=N * lb, lda=lda, ldb=ldb)
            capi.check(capi.w8a8_matmul(d, st))
            of = torch.zeros((M, N), dtype=torch.float32, device=dev)
            d.C, d.out_kind = of.data_ptr(), capi.TCE_OUT_FP32
            capi.check(capi.w8a8_matmul(d, st))
            torch.cuda.synchronize()
        finally:
            capi.lib().tce_w4a16_set_debug_mode(70)
        assert np.array_equal(o8.cpu().numpy(), want8), mode
        assert np.array_equal(of.cpu().numpy().view(np.uint32), wantf.view(np.uint32)), mode


@pytest.mark.parametrize("heads,sq,tgz,scale", [(12, 1, 512, 3.0), (12, 1, 512, 0.05), (4, 7, 33, 2.0), (2, 3, 1, 1.0), (3, 2, 5, 0.2), (2, 5, 100, 0.3), (1, 4, 511, 4.0), (2, 2, 513, 0.1),
                                                (1, 1, 4000, 1.0),
                                                # prompt sizes (over 1638 rows: the kernel form whose wave 0 walks the four rows' sums of a workgroup at once)
                                                (12, 512, 512, 3.0), (12, 512, 512, 0.05), (32, 64, 513, 1.0), (4, 500, 100, 0.3), (7, 300, 37, 2.0), (3, 700, 130, 0.02)])
def test_opt_softmax_q_against_the_oracle(dev, oracle, heads, sq, tgz, scale):
    """tce_opt_softmax_q on its own against orc_opt_softmax_q (pinned to the reference's softmax.cc run in place: tests/test_oracle_glue.py): rows whose own
    maximum is >= 1 (they do not wait for row (0, 0)'s first probability) and rows below it (small `scale`: they start their maximum from that probability),
    lengths that are not multiples of 4 / 16 / 32, a causal mask, padded output rows.  The device's expf may differ from libm's in the last bit: at most one
    int8 step on a few elements."""
    from tinychatengine_amd import capi
    rng = np.random.default_rng(heads * 100 + tgz)
    scores = (rng.standard_normal((heads, sq, tgz)) * scale).astype(np.float32)
    if heads * sq > 1638:  # rows on both sides of the quirk's threshold inside the same workgroups: every third row scaled the other way
        scores[:, ::3, :] *= np.float32(0.02 if scale >= 1.0 else 40.0)
    mask = np.zeros((sq, tgz), np.float32)
    for j in range(sq):
        if tgz - sq + j + 1 < tgz:
            mask[j, max(1, tgz - sq + j + 1):] = np.finfo(np.float32).min
    ldp = (tgz + 15) // 16 * 16
    probs = torch.full((heads, sq, ldp), 77, dtype=torch.int8, device=dev)
    ts, tm = _t(dev, scores), _t(dev, mask)
    capi.check(capi.lib().tce_opt_softmax_q(C.c_void_p(ts.data_ptr()), C.c_void_p(tm.data_ptr()), C.c_void_p(probs.data_ptr()), heads, sq, tgz, ldp,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    want = oracle.opt_softmax_q(scores, mask)
    got = probs.cpu().numpy()
    diff = np.abs(got[:, :, :tgz].astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() <= max(2e-4, 1.5 / diff.size), f"max step {diff.max()}, {int((diff > 0).sum())} of {diff.size} differ"
    assert (got[:, :, tgz:] == 77).all()


@pytest.mark.parametrize("heads,hd,max_keys,pos,m,a_qk", [(12, 64, 512, 511, 1, 2.0e-3), (12, 64, 512, 500, 1, 2.0e-6), (4, 64, 1040, 1000, 8, 1.0e-3), (3, 64, 64, 0, 5, 5.0e-4),
                                                          (2, 64, 4096, 4000, 3, 1.0e-5), (12, 64, 2048, 17, 2, 1.0e-3),
                                                          # head dimension 128 (OPT-6.7B: 32 heads of 128, model.h): the same cases' corners
                                                          (32, 128, 512, 511, 1, 1.0e-3), (4, 128, 1040, 1000, 8, 1.0e-6), (3, 128, 64, 0, 5, 3.0e-4), (2, 128, 2048, 1937, 3, 1.0e-3)])
def test_opt_attention_decode_equals_the_four_launches(dev, oracle, heads, hd, max_keys, pos, m, a_qk):
    """tce_opt_attention_decode (KV append + qk BMM + mask / softmax / int8 + pv BMM, one launch) against tce_opt_kv_append -> tce_w8a8_matmul -> tce_opt_softmax_q ->
    tce_w8a8_matmul on the same inputs: the attention rows and both caches bit for bit -- long contexts, several new rows, a position that is not a multiple of 16
    (the cached value rows are read in 16-byte pieces), score scales that put the rows' maxima above 1 (rows independent of row (0, 0)) and far below it (every row
    starts its maximum from row (0, 0)'s first probability), garbage in the caches behind the position."""
    from tinychatengine_amd import capi
    L = capi.lib()
    E, tgz = heads * hd, pos + m
    rng = np.random.default_rng(heads + pos + m + hd)
    kc0 = rng.integers(-128, 128, (heads, max_keys, hd), dtype=np.int8)
    vt0 = rng.integers(-128, 128, (heads, hd, max_keys), dtype=np.int8)
    q, k, v = (rng.integers(-128, 128, (m, E), dtype=np.int8) for _ in range(3))
    mask = np.zeros((m, tgz), np.float32)
    for r in range(m):
        mask[r, pos + r + 1:] = np.finfo(np.float32).min
    tq, tk, tv, tm = _t(dev, q), _t(dev, k), _t(dev, v), _t(dev, mask)
    st = torch.cuda.current_stream().cuda_stream
    vp = lambda t: C.c_void_p(t.data_ptr())
    a_pv = 1.0 / 127.0
    # the four launches
    kc_a, vt_a = _t(dev, kc0), _t(dev, vt0)
    capi.check(L.tce_opt_kv_append(vp(tk), vp(tv), vp(kc_a), vp(vt_a), heads, hd, m, pos, max_keys, C.c_void_p(st)))
    scores = torch.zeros((heads, m, tgz), dtype=torch.float32, device=dev)
    d = capi.W8A8Desc(M=m, N=tgz, K=hd, batch=heads, A=tq.data_ptr(), B=kc_a.data_ptr(), C=scores.data_ptr(), strideA=hd, strideB=max_keys * hd, strideC=m * tgz, lda=E, alpha=a_qk,
                      q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE, out_kind=capi.TCE_OUT_FP32)
    capi.check(capi.w8a8_matmul(d, st))
    ldp = (tgz + 15) // 16 * 16
    probs = torch.zeros((heads, m, ldp), dtype=torch.int8, device=dev)
    capi.check(L.tce_opt_softmax_q(vp(scores), vp(tm), vp(probs), heads, m, tgz, ldp, C.c_void_p(st)))
    out_a = torch.zeros((m, E), dtype=torch.int8, device=dev)
    d = capi.W8A8Desc(M=m, N=hd, K=tgz, batch=heads, A=probs.data_ptr(), B=vt_a.data_ptr(), C=out_a.data_ptr(), strideA=m * ldp, strideB=hd * max_keys, strideC=hd, lda=ldp, ldb=max_keys,
                      ldc=E, alpha=a_pv, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE, out_kind=capi.TCE_OUT_INT8)
    capi.check(capi.w8a8_matmul(d, st))
    # the one launch
    kc_b, vt_b = _t(dev, kc0), _t(dev, vt0)
    out_b = torch.full((m, E), 99, dtype=torch.int8, device=dev)
    capi.check(L.tce_opt_attention_decode(vp(tq), vp(tk), vp(tv), vp(kc_b), vp(vt_b), vp(tm), vp(out_b), heads, hd, m, pos, max_keys, 0, a_qk, a_pv, C.c_void_p(st)))
    torch.cuda.synchronize()
    assert torch.equal(kc_a, kc_b) and torch.equal(vt_a, vt_b), "caches"
    assert torch.equal(out_a, out_b), f"{int((out_a != out_b).sum())} attention outputs differ"
    assert a_qk < 1e-3 or int(probs.abs().sum()) > 0
    rmax = (scores.cpu().numpy() + mask[None]).max(axis=-1)
    if a_qk <= 2e-6 and hd == 64:
        assert (rmax < 1).all()   # every row (but (0, 0)) took the dependent path
    assert L.tce_opt_attention_decode(vp(tq), vp(tk), vp(tv), vp(kc_b), vp(vt_b), vp(tm), vp(out_b), heads, hd, 9, 0, max_keys, 0, a_qk, a_pv, None) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert L.tce_opt_attention_decode(vp(tq), vp(tk), vp(tv), vp(kc_b), vp(vt_b), vp(tm), vp(out_b), heads, 96, 1, 0, max_keys, 0, a_qk, a_pv, None) == capi.TCE_ERR_UNSUPPORTED_SHAPE  # head_dim


@pytest.mark.parametrize("M,N,K", [(512, 768, 3072), (108, 768, 3072), (300, 520, 2048 + 48), (64, 64, 1536), (1000, 200, 4096)])
def test_k_steps_cut_across_workgroups_stay_bit_exact(M, N, K):
    """Round 5 (tce_w8a8_desc_v2.scratch): few 64 x 64 tiles with a long k chain have their k-steps cut into runs on several workgroups; the int32 partial tiles are exact,
    so every cut gives the bits of the uncut launch (itself held to the oracle above) -- int8 output with bias and fp32 output accumulating into C, the rule's cut and
    every forced one, ragged M / N / K, repeated calls on one scratch area (its counters are back at zero after every call)."""
    from tinychatengine_amd import capi
    dev = torch.device("cuda:0")
    L = capi.lib()
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    ri = lambda *s: torch.randint(-128, 128, s, device=dev, generator=g, dtype=torch.int32).to(torch.int8)
    A, W, b8 = ri(M, K), ri(N, K), ri(N)
    bf = torch.empty(N, device=dev).normal_(0, 1, generator=g)
    c0 = torch.empty(M, N, device=dev).normal_(0, 1, generator=g)
    scratch = capi.w8a8_scratch(dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(kind, scratch_ptr, v2):
        if kind == "int8":
            out = torch.zeros(M, N, dtype=torch.int8, device=dev)
            d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=A.data_ptr(), B=W.data_ptr(), bias=b8.data_ptr(), C=out.data_ptr(), alpha=0.0005035400390625, beta=0.02130126953125,
                              q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8)
        else:
            out = c0.clone()
            d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=A.data_ptr(), B=W.data_ptr(), bias=bf.data_ptr(), C=out.data_ptr(), alpha=0.0007, q_min=-128, q_max=127,
                              bias_kind=capi.TCE_BIAS_FP32, out_kind=capi.TCE_OUT_FP32, accumulate=1)
        capi.check(capi.w8a8_matmul_v2(d, st, scratch_ptr) if v2 else capi.w8a8_matmul(d, st))
        torch.cuda.synchronize()
        return out

    try:
        for kind in ("int8", "fp32"):
            want = run(kind, None, False)
            assert bool(want.float().abs().sum() > 0)
            assert torch.equal(run(kind, None, True), want)  # the size-prefixed descriptor without a scratch area: the same launch
            for mode in (180, 182, 183, 184, 186, 188, 180):
                capi.check(L.tce_w4a16_set_debug_mode(mode))
                for rep in range(2):
                    got = run(kind, scratch.data_ptr(), True)
                    assert torch.equal(got, want), f"{kind} {M}x{N}x{K} cut mode {mode} call {rep}"
                assert int(scratch[:4096].to(torch.int32).sum().item()) == 0
    finally:
        L.tce_w4a16_set_debug_mode(180)
