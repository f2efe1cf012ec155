"""GPU parity tests of the W4A16 path (run with -m gpu on an MI355X).  Every kernel is reached through the C ABI
(libtce_hip.so) via tinychatengine_amd.capi / .matmul; the checker is oracle/ (the CPU restatement pinned to the
reference's own code) and the committed golden vectors.

Tolerance (north-star): W4A16 within 1e-3 relative -- see conftest.w4a16_close for the exact per-element rule; the
reference's own GPU-vs-CPU comparator is far looser (MSE <= 7e-4, llm/tests/cuda/test_ops.cu:656), also asserted.
The binary16-arithmetic AWQ entry point is bit-exact.
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
    capi.lib()  # raises if libtce_hip.so is missing: no fallback
    capi.set_gemv_config()  # automatic
    capi.set_gemm_config()
    return torch.device("cuda:0")


def _make(oracle, M, N, K, G, seed, random_zeros=False):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    qw, sc, zp, codes, d = oracle.quantize_q4_6(w, G)
    if random_zeros:  # general AWQ checkpoints carry real zero points; the kernel must read them (gemv_cuda.cu:159,166)
        nib = rng.integers(0, 16, (N, zp.shape[1] * 8), dtype=np.uint32)
        zp = (nib.reshape(N, -1, 8) << (np.arange(8, dtype=np.uint32) * 4)).sum(axis=2).astype(np.uint32)
    a = rng.standard_normal((M, K)).astype(np.float16)
    return qw, sc, zp, a


def _run(dev, qw, sc, zp, a, G, flags=0):
    from tinychatengine_amd import capi
    M, K = a.shape
    N = qw.shape[0]
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    tq, ts, tz, ta = t(qw.view(np.int32)), t(sc.view(np.float16)), t(zp.view(np.int32)), t(a)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
    d = capi.W4A16Desc(M=M, N=N, K=K, group_size=G, A=ta.data_ptr(), qweight=tq.data_ptr(), scales=ts.data_ptr(),
                       zeros=tz.data_ptr(), C=out.data_ptr(), flags=flags)
    capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _check(got, ref32, what):
    assert not np.isnan(got.astype(np.float32)).any(), f"{what}: output has NaN / unwritten elements"
    ok, worst = w4a16_close(got, ref32)
    mse = float(np.mean((got.astype(np.float64) - ref32.astype(np.float64)) ** 2))
    assert ok, f"{what}: worst |err|/tol = {worst:.3f} (tol = 1e-3*max(|ref|, rms/64)), mse = {mse:.3e}"
    assert mse <= 7e-4, f"{what}: mse {mse}"  # the reference's own threshold (tests/cuda/test_ops.cu:656)


GEMV_SHAPES = [
    # M, N, K, G
    (1, 256, 4096, 128), (1, 64, 1408, 128), (1, 100, 4096, 128), (2, 128, 2048, 64), (4, 96, 1024, 32),
    (3, 132, 2560, 128), (1, 4096, 4096, 128), (1, 512, 11008, 128), (1, 256, 14336, 128), (1, 1024, 5120, 128),
    (8, 64, 4096, 128), (5, 48, 1408, 128), (1, 16, 32, 32), (1, 24, 13824, 128),
]


@pytest.mark.parametrize("M,N,K,G", GEMV_SHAPES)
def test_gemv_matches_oracle(dev, oracle, M, N, K, G):
    from tinychatengine_amd import capi
    qw, sc, zp, a = _make(oracle, M, N, K, G, seed=M * 7 + N + K)
    ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, G)
    _check(_run(dev, qw, sc, zp, a, G, flags=capi.TCE_W4_FORCE_GEMV), ref32, f"gemv {M}x{N}x{K} g{G}")


@pytest.mark.parametrize("M,N,K,G", [(1, 192, 4096, 128), (2, 64, 1408, 128), (4, 64, 1024, 64), (1, 40, 512, 32)])
def test_gemv_reads_real_zero_points(dev, oracle, M, N, K, G):
    from tinychatengine_amd import capi
    qw, sc, zp, a = _make(oracle, M, N, K, G, seed=99 + N, random_zeros=True)
    ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, G)
    _check(_run(dev, qw, sc, zp, a, G, flags=capi.TCE_W4_FORCE_GEMV), ref32, f"gemv zeros {M}x{N}x{K} g{G}")


def test_every_gemv_variant(dev, oracle):
    """Each compiled launch geometry (rows/wave, waves, K split, pipeline depth) on shapes with K tails and N tails."""
    from tinychatengine_amd import capi
    cases = [(1, 264, 8192, 128), (2, 72, 11008, 128), (4, 40, 6144 + 1408, 64)]
    data = []
    for (M, N, K, G) in cases:
        qw, sc, zp, a = _make(oracle, M, N, K, G, seed=N, random_zeros=True)
        data.append((M, N, K, G, qw, sc, zp, a, oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, G)[0]))
    try:
        for v in capi.gemv_variants():
            capi.set_gemv_config(*v)
            for (M, N, K, G, qw, sc, zp, a, ref32) in data:
                _check(_run(dev, qw, sc, zp, a, G, flags=capi.TCE_W4_FORCE_GEMV), ref32, f"variant {v} on {M}x{N}x{K}")
    finally:
        capi.set_gemv_config()


STREAM_CFGS = [(2, 16, 2), (2, 15, 3), (1, 16, 3), (1, 9, 2), (2, 4, 2), (1, 1, 2)]


@pytest.mark.parametrize("M", [1, 2])
def test_stream_kernel_every_config(dev, oracle, M):
    """The persistent kernel: rows per row group x waves per workgroup x ring depth, on grouped launches with odd N
    (row-group tails), K tails (K % 2048 != 0), more row groups than waves and fewer row groups than waves.  M = 2 is
    not a shape it takes: the forced configuration must then fall through to the row-block kernel, not fail."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4, forward_group
    g = torch.Generator(device=dev).manual_seed(31 + M)
    for (K, Ns) in [(4096, [9001, 520, 72]), (1408, [24]), (11008, [4104])]:
        x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
        lins = [Linear_half_int4.from_float(torch.empty(n, K, device=dev).normal_(0, 0.02, generator=g), 128) for n in Ns]
        refs = [oracle.w4a16_gemv_q4_6(x.cpu().numpy(), l.weight.cpu().numpy().view(np.uint32), l.scale.cpu().numpy(),
                                       l.zero_point.cpu().numpy().view(np.uint32), M, l.out_features, K, 128)[0] for l in lins]
        try:
            for cfg in STREAM_CFGS:
                capi.set_gemv_config(cfg[0], cfg[1], 0, cfg[2])
                outs = [torch.full((M, n), float("nan"), dtype=torch.float16, device=dev) for n in Ns]
                forward_group(lins, x, outs)
                torch.cuda.synchronize()
                for o, r, n in zip(outs, refs, Ns):
                    _check(o.cpu().numpy(), r, f"stream cfg {cfg} K={K} N={n} M={M}")
        finally:
            capi.set_gemv_config()


def test_zero_point_8_fast_path_is_bit_identical(dev, oracle):
    """TCE_W4_ZERO_POINT_IS_8 skips the zeros stream; with all-8 zero points the output bits must not change."""
    from tinychatengine_amd import capi
    for (M, N, K) in [(1, 4096, 4096), (1, 520, 11008), (2, 264, 4096)]:
        qw, sc, zp, a = _make(oracle, M, N, K, 128, seed=N)
        assert capi.lib().tce_w4a16_check_zero_point_8(torch.from_numpy(zp.view(np.int32)).to(dev).data_ptr(), zp.size) == 1
        plain = _run(dev, qw, sc, zp, a, 128, flags=capi.TCE_W4_FORCE_GEMV)
        fast = _run(dev, qw, sc, zp, a, 128, flags=capi.TCE_W4_FORCE_GEMV | capi.TCE_W4_ZERO_POINT_IS_8)
        assert np.array_equal(plain.view(np.uint16), fast.view(np.uint16))
        ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, 128)
        _check(fast, ref32, f"zero8 fast path {M}x{N}x{K}")
    zp2 = zp.copy()
    zp2[3, 0] ^= 0x10
    assert capi.lib().tce_w4a16_check_zero_point_8(torch.from_numpy(zp2.view(np.int32)).to(dev).data_ptr(), zp2.size) == 0


def test_gemv_golden_vector(dev, golden):
    """The committed vector produced by the reference's quantizer + naive_mat_mul_int4 (K=1408: padded scale rows)."""
    from tinychatengine_amd import capi
    M, N, K, G = (int(v) for v in golden["w4_dims"])
    got = _run(dev, golden["w4_qweight"], golden["w4_scales"], golden["w4_zeros"], golden["w4_a"], G, flags=capi.TCE_W4_FORCE_GEMV)
    _check(got, golden["w4_expected_f32"], "golden gemv")
    got = _run(dev, golden["w4_qweight"], golden["w4_scales"], golden["w4_zeros"], golden["w4_a"], G, flags=capi.TCE_W4_FORCE_GEMM)
    _check(got, golden["w4_expected_f32"], "golden gemm")


GEMM_SHAPES = [(64, 256, 512, 128), (33, 200, 1024, 128), (512, 384, 4096, 128), (128, 128, 1408 + 128 * 5, 64), (17, 72, 256, 32), (9, 16, 128, 128),
               # groups of 64 / 32 on the LDS-DMA GEMM (two / four groups per k-block; one and two quartets), and with a K tail (GEMV fallback)
               (130, 200, 1024, 64), (70, 136, 512, 32), (300, 2100, 384, 64), (260, 520, 256, 32), (40, 64, 192, 64)]


@pytest.mark.parametrize("M,N,K,G", GEMM_SHAPES)
def test_gemm_matches_oracle(dev, oracle, M, N, K, G):
    from tinychatengine_amd import capi
    qw, sc, zp, a = _make(oracle, M, N, K, G, seed=M + N * 3 + K, random_zeros=True)
    ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, G)
    try:
        for v in [None] + capi.gemm_variants():
            capi.set_gemm_config(*(v or (0, 0)))
            _check(_run(dev, qw, sc, zp, a, G, flags=capi.TCE_W4_FORCE_GEMM), ref32, f"gemm {v} {M}x{N}x{K} g{G}")
    finally:
        capi.set_gemm_config()


@pytest.mark.parametrize("M,N,K", [(64, 256, 512), (33, 200, 1024), (130, 132, 256)])
def test_gemm_add_to_c_every_variant(dev, oracle, M, N, K):
    """TCE_W4_ADD_TO_C on the GEMM kernels: C = half(C + half(gemm)), bit for bit against the same variant without the flag, for row tails
    (M % 16 != 0), column tails that end inside a 16-byte piece (N % 8 != 0) and an unaligned leading dimension (the scalar store path)."""
    from tinychatengine_amd import capi
    qw, sc, zp, a = _make(oracle, M, N, K, 128, seed=M + N + K, random_zeros=True)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    tq, ts, tz, ta = t(qw.view(np.int32)), t(sc.view(np.float16)), t(zp.view(np.int32)), t(a)
    c0 = (torch.randn(M, N + 3, device=dev) * 0.5).to(torch.float16)
    try:
        for v in [None] + capi.gemm_variants():
            capi.set_gemm_config(*(v or (0, 0)))
            for ldc in (N, N + 3):
                plain = torch.zeros(M, ldc, dtype=torch.float16, device=dev)
                acc = c0[:, :ldc].clone().contiguous()
                for out, flags in ((plain, capi.TCE_W4_FORCE_GEMM), (acc, capi.TCE_W4_FORCE_GEMM | capi.TCE_W4_ADD_TO_C)):
                    d = capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=ta.data_ptr(), qweight=tq.data_ptr(), scales=ts.data_ptr(),
                                       zeros=tz.data_ptr(), C=out.data_ptr(), ldc=ldc, flags=flags)
                    capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                want = (c0[:, :N] + plain[:, :N])
                assert torch.equal(acc[:, :N], want), f"variant {v} ldc {ldc}: add-to-C differs"
                if ldc > N:
                    assert torch.equal(acc[:, N:], c0[:, N:ldc]) and not plain[:, N:].any(), f"variant {v}: wrote past column N"
    finally:
        capi.set_gemm_config()


@pytest.mark.parametrize("pad", [8, 24, 4])
def test_gemm_strided_activations(dev, oracle, pad):
    """lda != K (rows are 16-byte pieces for the LDS-DMA loads; the ABI requires lda % 8 == 0 and refuses anything else)."""
    from tinychatengine_amd import capi
    M, N, K = 200, 264, 1024
    qw, sc, zp, a = _make(oracle, M, N, K, 128, seed=pad, random_zeros=True)
    ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, 128)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    tq, ts, tz = t(qw.view(np.int32)), t(sc.view(np.float16)), t(zp.view(np.int32))
    buf = torch.full((M, K + pad), float("nan"), dtype=torch.float16, device=dev)
    buf[:, :K] = t(a)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
    d = capi.W4A16Desc(M=M, N=N, K=K, group_size=128, A=buf.data_ptr(), lda=K + pad, qweight=tq.data_ptr(), scales=ts.data_ptr(),
                       zeros=tz.data_ptr(), C=out.data_ptr(), flags=capi.TCE_W4_FORCE_GEMM)
    rc = capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream)
    if pad % 8:
        assert rc == capi.TCE_ERR_BAD_ARG or rc < 0, "an lda that is not a multiple of 8 halves must be refused"
        return
    capi.check(rc)
    torch.cuda.synchronize()
    _check(out.cpu().numpy(), ref32, f"gemm lda = K + {pad}")


def test_gemm_is_transpose_detecting(dev, oracle):
    """Asymmetric data: A = one-hot rows selects single k, so a swapped fragment index shows up as a wrong column."""
    from tinychatengine_amd import capi
    M, N, K, G = 48, 80, 256, 128
    qw, sc, zp, _ = _make(oracle, M, N, K, G, seed=5)
    a = np.zeros((M, K), np.float16)
    for m in range(M):
        a[m, (m * 37 + 11) % K] = 1.0 + m / 64.0
    ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, G)
    _check(_run(dev, qw, sc, zp, a, G, flags=capi.TCE_W4_FORCE_GEMM), ref32, "one-hot gemm")
    _check(_run(dev, qw, sc, zp, a[:4].copy(), G, flags=capi.TCE_W4_FORCE_GEMV), ref32[:4], "one-hot gemv")


def test_dispatch_threshold_and_default_path(dev, oracle):
    """M <= 8 -> GEMV kernels, M > 8 -> MFMA GEMM; both must agree with the oracle at the boundary."""
    for M in (8, 9):
        qw, sc, zp, a = _make(oracle, M, 144, 1024, 128, seed=M)
        ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, 144, 1024, 128)
        _check(_run(dev, qw, sc, zp, a, 128), ref32, f"default path M={M}")


def test_grouped_launch_equals_separate_launches(dev, oracle):
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4, forward_group
    K, G = 4096, 128
    x = torch.randn(1, K, device=dev).to(torch.float16)
    g = torch.Generator(device=dev).manual_seed(7)
    lins = [Linear_half_int4.from_float(torch.empty(n, K, device=dev).normal_(0, 0.02, generator=g), G) for n in (512, 128, 144)]
    sep = [l.forward(x) for l in lins]
    outs = [torch.full_like(s, float("nan")) for s in sep]
    forward_group(lins, x, outs)
    torch.cuda.synchronize()
    for s, o, l in zip(sep, outs, lins):
        assert not torch.isnan(o).any()
        ref32, _ = oracle.w4a16_gemv_q4_6(x.cpu().numpy(), l.weight.cpu().numpy().view(np.uint32), l.scale.cpu().numpy(),
                                          l.zero_point.cpu().numpy().view(np.uint32), 1, l.out_features, K, G)
        _check(o.cpu().numpy(), ref32, "grouped launch")
        _check(s.cpu().numpy(), ref32, "separate launch")


def test_plan_replay_matches_eager(dev):
    from tinychatengine_amd.decode import DecodeLinears, SHAPES
    dl = DecodeLinears(SHAPES["tiny"], device=dev)
    for li in range(dl.n_layers):
        dl.run_block(li)
    dl.run_lm_head()
    torch.cuda.synchronize()
    eager = [dl.logits.clone(), dl.out_down.clone(), dl.out_up.clone(), dl.out_qkv[1].clone()]
    for t in (dl.logits, dl.out_down, dl.out_up, dl.out_qkv[1]):
        t.fill_(float("nan"))
    plan = dl.make_plan()
    s = torch.cuda.current_stream().cuda_stream
    plan.launch(s)
    plan.launch(s)
    torch.cuda.synchronize()
    for e, t in zip(eager, (dl.logits, dl.out_down, dl.out_up, dl.out_qkv[1])):
        assert torch.equal(e, t)
    assert plan.n_launches == dl.n_layers * 4 + 1
    plan.close()


def test_awq_fp16acc_is_bit_exact(dev, oracle, golden):
    """naive_mat_mul_fp16_int4: every op rounded to binary16 -- the GPU result must equal the reference bit for bit."""
    from tinychatengine_amd.matmul import MatmulOperator, matmul_params, matrix
    cases = [tuple(int(v) for v in golden["awq_dims"]) + (golden["awq_qweight"], golden["awq_scales"], golden["awq_a"], golden["awq_expected_f16"])]
    rng = np.random.default_rng(11)
    M, N, K, G = 3, 72, 512, 128
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    codes, d = oracle.group_quantize(w, G)
    q5, s5, _ = oracle.pack_q4_5(codes, d, N, K, G)
    a = rng.standard_normal((M, K)).astype(np.float16)
    cases.append((M, N, K, G, q5, s5, a, oracle.naive_mat_mul_fp16_int4(a, q5, s5, M, N, K, G)))
    for (M, N, K, G, q5, s5, a, exp) in cases:
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        out = torch.zeros((M, N), dtype=torch.float16, device=dev)
        p = matmul_params(A=matrix(M, K, t(a)), B=matrix(K, N // 8, t(q5.view(np.int32))), C=matrix(M, N, out),
                          fp16_scales=t(s5.view(np.float16)), block_size=G)
        MatmulOperator().naive_mat_mul_fp16_int4(p)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint16), exp.view(np.uint16))


@pytest.mark.parametrize("M", [1, 4, 64])
def test_awq_layout_gemm(dev, oracle, M):
    """gemm_forward_cuda surface: q4_5 in, result must match the q4_6 oracle on the same codes (fp32 accumulate)."""
    from tinychatengine_amd.matmul import MatmulOperator, matmul_params, matrix
    rng = np.random.default_rng(M)
    N, K, G = 136, 1024, 128
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    qw, sc, zp, codes, d = oracle.quantize_q4_6(w, G)
    q5, s5, _ = oracle.pack_q4_5(codes, d.reshape(-1), N, K, G)
    a = rng.standard_normal((M, K)).astype(np.float16)
    ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, G)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
    p = matmul_params(A=matrix(M, K, t(a)), B=matrix(K, N // 8, t(q5.view(np.int32))), C=matrix(M, N, out),
                      half_scales=t(s5.view(np.float16)), block_size=G)
    ws = MatmulOperator().gemm_forward_cuda(p, 8)
    torch.cuda.synchronize()
    _check(out.cpu().numpy(), ref32, f"awq gemm M={M}")
    out.fill_(float("nan"))
    MatmulOperator().gemm_forward_cuda(p, 8, workspace=ws, repack=False)  # cached re-layout
    torch.cuda.synchronize()
    _check(out.cpu().numpy(), ref32, f"awq gemm (cached) M={M}")


def test_error_codes_match_reference_behaviour(dev):
    """Unsupported group size: the reference prints and exits (gemv_cuda.cu:254-256); here a distinct error code."""
    from tinychatengine_amd import capi
    x = torch.zeros(1, 256, dtype=torch.float16, device=dev)
    q = torch.zeros(16, 32, dtype=torch.int32, device=dev)
    s = torch.zeros(16, 8, dtype=torch.float16, device=dev)
    z = torch.zeros(16, 1, dtype=torch.int32, device=dev)
    o = torch.zeros(1, 16, dtype=torch.float16, device=dev)
    d = capi.W4A16Desc(M=1, N=16, K=256, group_size=96, A=x.data_ptr(), qweight=q.data_ptr(), scales=s.data_ptr(), zeros=z.data_ptr(), C=o.data_ptr())
    assert capi.w4a16_forward(d, 0) == capi.TCE_ERR_UNSUPPORTED_GROUP and "Unsupported group size" in capi.last_error()
    d.group_size = 128
    d.K = 250
    assert capi.w4a16_forward(d, 0) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    d.K = 256
    d.A = None
    assert capi.w4a16_forward(d, 0) == capi.TCE_ERR_BAD_ARG


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json full sizes
# ---------------------------------------------------------------------------------------------------------
FULL = [(4096, 4096), (11008, 4096), (4096, 11008), (14336, 4096), (4096, 14336), (1024, 4096), (12288, 4096),
        # the two lm_head shapes (Llama-2 / Llama-3 vocabularies); since round 2's re-tuning both run on the row-block kernel
        # (four rows per wave, four waves per workgroup); the persistent kernel at this size: test_persistent_kernel_at_lm_head_size
        (32000, 4096), (128256, 4096),
        # BASELINE config 5 (cfgD): Llama-2-13B's linears, llm/include/model.h:72 -- fused qkv, o, gate / up, down
        (15360, 5120), (5120, 5120), (13824, 5120), (5120, 13824)]


def _floor_gate(got, ref32, what, limit=0.01):
    """how much of the pass is owed to the rms/64 floor of w4a16_close?  Recorded for every full-size case
    (gpurun_out/parity_report.jsonl -> DESIGN.md section 4) and bounded: at most `limit` of the elements may exceed the plain
    1e-3 * |ref| rule, and none whose |ref| is at least rms/64."""
    rep = w4a16_report(got, ref32)
    record_parity(what, rep)
    assert rep["frac_fail"] == 0.0, (what, rep)
    assert rep["frac_over_plain"] < limit, (what, rep)
    assert rep["worst_plain"] <= 1.0, (what, rep)
    return rep


@pytest.mark.parametrize("N,K", FULL)
def test_full_size_decode_gemv(dev, oracle, N, K):
    """configs[1] and [4]: M=1 at the Llama shapes on the AUTOMATIC dispatch, every output against the oracle (threaded over
    weight rows on the host), plus two size-independent properties: exact homogeneity under x -> 2x and bit-identical results
    for 8-way row shards (= the column shards of SURVEY 8e)."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    G = 128
    g = torch.Generator(device=dev).manual_seed(1234 + N)
    lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), G)
    x = torch.empty(1, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    y = torch.full((1, N), float("nan"), dtype=torch.float16, device=dev)
    want_kernel = "row-block"
    assert capi.describe_dispatch(lin.desc(x, y)) == f"gemv passes=1 kernel={want_kernel}"
    y = lin.forward(x)
    torch.cuda.synchronize()
    ref32 = oracle.w4a16_gemv_q4_6_mt(x.cpu().numpy(), lin.weight.cpu().numpy().view(np.uint32), lin.scale.cpu().numpy(),
                                      lin.zero_point.cpu().numpy().view(np.uint32), 1, N, K, G)
    _check(y.cpu().numpy(), ref32, f"full {N}x{K}")
    _floor_gate(y.cpu().numpy(), ref32, f"decode M=1 {N}x{K} ({want_kernel})")
    if N * K >= 200_000_000:  # test_persistent_kernel_at_lm_head_size: the persistent kernel (no longer the automatic choice here) at
        try:                  # full size, by name -- a row's arithmetic does not depend on the kernel, so the outputs are bit-identical
            capi.set_gemv_config(2, 16, 0, 2)
            yp = torch.full((1, N), float("nan"), dtype=torch.float16, device=dev)
            assert capi.describe_dispatch(lin.desc(x, yp)) == "gemv passes=1 kernel=persistent"
            lin.forward(x, yp)
            torch.cuda.synchronize()
            assert torch.equal(yp, y)
        finally:
            capi.set_gemv_config()
    y2 = lin.forward(x * 2)
    torch.cuda.synchronize()
    # power-of-two scaling is exact in fp16/fp32 -- except where the fp16 OUTPUT is subnormal (|y| < 2^-14), where
    # fp16(2v) and 2*fp16(v) may differ by one subnormal ulp (2^-24)
    normal = y.abs() >= 2.0 ** -13
    assert torch.equal(y2[normal], (y * 2)[normal]), "x -> 2x must double every (normal) output exactly"
    assert (y2.float() - 2 * y.float()).abs().max().item() <= 2.0 ** -23
    # column sharding: each shard's rows must reproduce the full result bit for bit under one geometry
    try:
        capi.set_gemv_config(2, 4, 1, 2)
        full = lin.forward(x)
        parts = [lin.shard(r, 8).forward(x) for r in range(8)]
        torch.cuda.synchronize()
        assert torch.equal(torch.cat(parts, dim=1), full)
    finally:
        capi.set_gemv_config()


@pytest.mark.parametrize("N,K", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_full_size_prefill_gemm(dev, oracle, N, K):
    """configs[2]: M=512 on the MFMA path.  128 of the 512 rows (every 4th: all 16-row MFMA tiles, all four row positions mod 4
    over the batch's 64-row blocks) are checked against the oracle, threaded over weight rows on the host (5.8e9 scalar MACs for
    the widest shape); the other rows through two properties: rows 256.. repeat rows 0..255 and must give identical outputs
    (row tiles 4..7 against 0..3), and the GEMV kernel (validated above) must agree on 4 more rows."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    G, M = 128, 512
    g = torch.Generator(device=dev).manual_seed(77 + N)
    lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), G)
    xh = torch.empty(M // 2, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    x = torch.cat([xh, xh], dim=0).contiguous()
    y = lin.forward(x)
    torch.cuda.synchronize()
    assert torch.equal(y[: M // 2], y[M // 2:]), "duplicate rows must produce identical outputs"
    rows = [r for r in range(0, 512) if (r % 4) == ((r // 64) % 4)]
    assert len(rows) == 128
    xs = x[rows].cpu().numpy()
    ref32 = oracle.w4a16_gemv_q4_6_mt(xs, lin.weight.cpu().numpy().view(np.uint32), lin.scale.cpu().numpy(),
                                      lin.zero_point.cpu().numpy().view(np.uint32), len(rows), N, K, G)
    _check(y[rows].cpu().numpy(), ref32, f"prefill {N}x{K}")
    _floor_gate(y[rows].cpu().numpy(), ref32, f"prefill M=512 {N}x{K} (128 rows)")
    if N == 4096 and K == 4096:
        # and ALL 512 rows of the 4096 x 4096 shape against the threaded oracle (8.6e9 scalar MACs: tens of seconds on the box's host
        # cores) -- on 512 DISTINCT rows, both weight forms (q4_6 as loaded: the 64-row LDS-DMA tiles; pre-packed: the 128-row tiles)
        xa = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
        ref_all = oracle.w4a16_gemv_q4_6_mt(xa.cpu().numpy(), lin.weight.cpu().numpy().view(np.uint32), lin.scale.cpu().numpy(),
                                            lin.zero_point.cpu().numpy().view(np.uint32), M, N, K, G)
        _check(lin.forward(xa).cpu().numpy(), ref_all, "prefill 4096 x 4096, all 512 rows, q4_6 as loaded")
        lin.prepack()
        ya = lin.forward(xa)
        torch.cuda.synchronize()
        assert "gemm-pk" in capi.describe_dispatch(lin.desc(xa, ya))
        _check(ya.cpu().numpy(), ref_all, "prefill 4096 x 4096, all 512 rows, pre-packed")
        lin.packed = None
    yv = torch.empty(4, N, dtype=torch.float16, device=dev)
    d = lin.desc(x[3:7].contiguous(), yv)
    d.flags = capi.TCE_W4_FORCE_GEMV
    capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ok, worst = w4a16_close(y[3:7].cpu().numpy(), yv.float().cpu().numpy(), rel=2e-3)  # two fp16-rounded results
    assert ok, worst


def test_randomized_shapes_match_oracle(dev, oracle):
    """Seeded random (M, N, K, G, zero points): every dispatch path (row-block GEMV with and without a K split, M-row
    batches, the GEMV fallback of non-128 groups at M > 8, the MFMA GEMM) against the oracle."""
    from tinychatengine_amd import capi
    rng = np.random.default_rng(20240807)
    for case in range(24):
        G = int(rng.choice([32, 64, 128]))
        K = G * int(rng.integers(1, 40))
        if K % 32:
            K = (K // 32 + 1) * 32
        while K % G:
            K += 32
        N = int(rng.integers(1, 700))
        M = int(rng.choice([1, 1, 1, 2, 3, 5, 8, 9, 17, 70]))
        qw, sc, zp, a = _make(oracle, M, N, K, G, seed=1000 + case, random_zeros=bool(case % 2))
        ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, G)
        _check(_run(dev, qw, sc, zp, a, G), ref32, f"random case {case}: {M}x{N}x{K} g{G}")


SKINNY_SHAPES = [(2, 4096, 4096), (16, 4096, 4096), (3, 100, 1408), (8, 264, 11008), (16, 17, 128), (5, 40, 256), (9, 2050, 2048), (13, 31, 14336), (7, 16400, 512),
                 # M > 16: 16-row slices of the batch on gridDim.y (taken while N is small, see skinny_supports)
                 (17, 40, 2048), (33, 200, 1024), (100, 264, 1408), (128, 512, 256), (40, 16400, 256)]


@pytest.mark.parametrize("M,N,K", SKINNY_SHAPES)
def test_small_batch_kernel_matches_oracle(dev, oracle, M, N, K):
    """w4a16_skinny.hip (3 <= M <= 128; M = 2 stays on the GEMV kernel): automatic and every forced K split, plain and random zero points, with the
    zero-point-8 promise, N tails (N % 16 != 0), a single k-block, rows past M, batch slices (M > 16, partial last slice)."""
    from tinychatengine_amd import capi
    L = capi.lib()
    try:
        for rz in (False, True):
            qw, sc, zp, a = _make(oracle, M, N, K, 128, seed=M * 13 + N + K, random_zeros=rz)
            ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, 128)
            for mode in (20, 21, 22, 24, 28, 30):  # 30: workgroups of 8 tiles sharing the activation blocks (K % 256 == 0, else it falls back)
                if 20 < mode < 30 and (mode - 20) > K // 128:
                    continue
                capi.check(L.tce_w4a16_set_debug_mode(mode))
                _check(_run(dev, qw, sc, zp, a, 128), ref32, f"skinny ks-mode {mode} {M}x{N}x{K} random_zeros={rz}")
                if not rz:
                    _check(_run(dev, qw, sc, zp, a, 128, flags=capi.TCE_W4_ZERO_POINT_IS_8), ref32, f"skinny z8 ks-mode {mode} {M}x{N}x{K}")
        # and the older paths still answer when the small-batch kernel is switched off
        capi.check(L.tce_w4a16_set_debug_mode(29))
        _check(_run(dev, qw, sc, zp, a, 128), ref32, f"skinny off {M}x{N}x{K}")
    finally:
        L.tce_w4a16_set_debug_mode(20)



def _cuda_gemv_golden():
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_cuda_gemv_golden", os.path.join(here, "golden", "make_cuda_gemv_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, np.load(os.path.join(here, "golden", "cuda_gemv_golden.npz"))


@pytest.mark.parametrize("idx", range(7))
def test_decode_against_the_reference_cuda_kernel_itself(dev, oracle, idx):
    """The HIP path against OUTPUTS OF THE REFERENCE'S OWN CUDA GEMV KERNEL (gemv_forward_cuda -> gemv_kernel_g128, kernels/cuda/gemv_cuda.cu:
    140-260) at the BASELINE decode shapes: tests/golden/cuda_gemv_golden.npz was produced by running that kernel's unmodified source through
    the host emulation in oracle/cuda_emul/ (tests/golden/make_cuda_gemv_golden.py); the inputs are regenerated from the recorded seeds.
    Tolerance: the north star's 1e-3 (with the floor for outputs that are tiny by cancellation, like everywhere)."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    mod, gold = _cuda_gemv_golden()
    M, N, K, seed = (int(v) for v in gold["cases"][idx])
    a, qw, sc, zp = mod.gemv_case(oracle, M, N, K, seed)
    want = gold[f"out_{M}_{N}_{K}"]
    lin = Linear_half_int4(torch.from_numpy(qw.view(np.int32)).to(dev), torch.from_numpy(sc).to(dev), torch.from_numpy(zp.view(np.int32)).to(dev), 128)
    x = torch.from_numpy(a).to(dev)
    for cfg in (None, (2, 16, 0, 2)):  # the automatic choice (row-block kernel) and the persistent kernel
        try:
            capi.set_gemv_config(*(cfg or (0, 0, 0, 0)))
            y = lin.forward(x)
            torch.cuda.synchronize()
        finally:
            capi.set_gemv_config()
        got = y.cpu().numpy()
        ok, worst = w4a16_close(got, want.astype(np.float32))
        assert ok, f"{M}x{N}x{K} cfg {cfg}: worst |err|/tol against the reference CUDA kernel = {worst:.3f}"
        if cfg is None:
            same = float((got.view(np.uint16) == want.view(np.uint16)).mean())
            rep = w4a16_report(got, want.astype(np.float32))
            rep["bit_identical_to_the_reference_cuda_kernel"] = round(same, 4)
            record_parity(f"decode M={M} {N}x{K} vs the reference CUDA kernel's own output", rep)
            assert same >= 0.80, same  # both round an fp32 sum to binary16: most outputs are the same half
