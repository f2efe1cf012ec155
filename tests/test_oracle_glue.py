"""Pins the oracle's restatements of the ops either side of the W4A16 linears against the REFERENCE's own
CUDA kernel sources, run on the CPU.

The reference has these ops only as CUDA kernels -- llm/src/ops/cuda/softmax.cu:4-40, BMM_F16T.cu:28-78, RotaryPosEmb.cu:4-34, and
add_half / SiLuMul_half in llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu:12-30 -- which cannot be run here on a device.  They are
plain per-thread loops (no shared memory, no barriers, no shuffles), so `make -C oracle glue` compiles those very sources against the
host emulation in oracle/cuda_emul/ (threads one after the other, binary16 intrinsics as single correctly rounded operations) into
oracle/_ref/glue_harness, and this file compares orc_* with what they compute, bit for bit.  What is pinned is the kernels' structure
and order of operations; hexp is the same model on both sides (C library expf rounded to binary16), as tce_oracle.c says.
generalT5LayerNorm (LlamaRMSNorm.cu:68-115) DOES use warp shuffles, a shared variable and __syncthreads: for it the emulation runs the
threads of a block as concurrent OS threads (tce_emul::launch_concurrent: pthread barriers for __syncthreads and for each warp's
shuffle), and orc_rmsnorm_half -- the hand-restated reduction tree -- is compared with it as well.  rsqrtf is 1 / sqrtf on both sides."""
import os
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(REPO, "oracle", "_ref", "glue_harness")

needs_harness = pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/glue_harness not built (needs /root/reference at build time)")


def _run(tmp_path, op, ints, inputs, n_out):
    paths = []
    for i, a in enumerate(inputs):
        p = tmp_path / f"in{i}.bin"
        np.ascontiguousarray(a, np.float16).tofile(p)
        paths.append(str(p))
    outs = [str(tmp_path / f"out{i}.bin") for i in range(n_out)]
    r = subprocess.run([HARNESS, op, *[str(v) for v in ints], *paths, *outs], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"glue_harness {op}: rc={r.returncode}\n{r.stdout}\n{r.stderr}"
    return [np.fromfile(o, np.float16) for o in outs]


def _bits(a):
    return np.ascontiguousarray(a, np.float16).view(np.uint16)


def _halves(rng, shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float16)


@needs_harness
@pytest.mark.parametrize("n", [1, 1023, 4096, 11008 + 5])
def test_add_half_and_silu_mul_against_the_reference_kernels(tmp_path, oracle, n):
    rng = np.random.default_rng(n)
    a, b = _halves(rng, n, 3.0), _halves(rng, n, 3.0)
    a[: min(n, 4)] = np.array([65504, -65504, 6e-8, 0], np.float16)[: min(n, 4)]  # overflow to inf, a subnormal, zero
    (ref,) = _run(tmp_path, "add", [n], [a, b], 1)
    assert np.array_equal(_bits(ref), _bits(oracle.add_half(a, b)))
    (ref,) = _run(tmp_path, "silu", [n], [a, b], 1)
    assert np.array_equal(_bits(ref), _bits(oracle.silu_mul_half(a, b)))


@needs_harness
@pytest.mark.parametrize("x,y,z", [(32, 1, 129), (3, 5, 64), (2, 17, 1), (1, 1, 2048)])
def test_softmax_against_the_reference_kernel(tmp_path, oracle, x, y, z):
    rng = np.random.default_rng(x * 100 + y * 10 + z)
    s = _halves(rng, (x, y, z), 4.0)
    s[0, 0, : min(z, 3)] = np.array([-65504, 11.0, -30.0], np.float16)[: min(z, 3)]  # a masked key, a dominant one, an underflowing one
    (ref,) = _run(tmp_path, "softmax", [x, y, z], [s], 1)
    assert np.array_equal(_bits(ref), _bits(oracle.softmax_half(s)).ravel())


@needs_harness
@pytest.mark.parametrize("batch,m,n,k", [(32, 1, 200, 128), (4, 7, 33, 64), (2, 3, 5, 300), (32, 1, 128, 257)])
def test_bmm_f16t_against_the_reference_operator(tmp_path, oracle, batch, m, n, k):
    rng = np.random.default_rng(batch + m + n + k)
    a, w = _halves(rng, (batch, m, k)), _halves(rng, (batch, n, k))
    a[0, 0, :2] = np.array([6e-5, 6e-8], np.float16)  # products in the subnormal range
    alpha = np.float16(0.0884)
    (ref,) = _run(tmp_path, "bmm", [batch, m, n, k, int(np.array([alpha]).view(np.uint16)[0])], [a, w], 1)
    assert np.array_equal(_bits(ref), _bits(oracle.bmm_f16t(a, w, alpha)).ravel())


@needs_harness
@pytest.mark.parametrize("heads,ln,hd,start", [(32, 1, 128, 77), (4, 9, 128, 0), (2, 3, 64, 5)])
def test_rope_against_the_reference_kernel(tmp_path, oracle, heads, ln, hd, start):
    rng = np.random.default_rng(heads + ln + hd + start)
    positions = start + ln + 2
    q, k = _halves(rng, (heads, ln, hd)), _halves(rng, (heads, ln, hd))
    cos, sin = _halves(rng, (positions, hd)), _halves(rng, (positions, hd))
    rq, rk = _run(tmp_path, "rope", [heads, ln, hd, start, positions], [q, k, cos, sin], 2)
    oq, ok = oracle.rope_half(q, k, cos, sin, start)
    assert np.array_equal(_bits(rq), _bits(oq).ravel()) and np.array_equal(_bits(rk), _bits(ok).ravel())


@needs_harness
@pytest.mark.parametrize("m,n", [(1, 4096), (3, 1024), (2, 520), (1, 11008), (2, 64)])
def test_rmsnorm_against_the_reference_kernel(tmp_path, oracle, m, n):
    """LlamaRMSNorm_cuda::forward with its own grid / block (min(n, 1024) / 2 threads; 512 when n % 32 != 0): thread-strided partial sums,
    the butterfly over each warp, the second butterfly over the warps' sums, the clamp, the fp16 store."""
    rng = np.random.default_rng(m * 7 + n)
    x = _halves(rng, (m, n), 2.5)
    x[0, :2] = np.array([60000, -60000], np.float16)  # large values: the clamp
    gamma = (1 + 0.2 * rng.standard_normal(n)).astype(np.float32)
    eps = 1e-6
    gpath = tmp_path / "gamma.bin"
    gamma.tofile(gpath)
    xin = tmp_path / "x.bin"
    x.tofile(xin)
    out = tmp_path / "out.bin"
    r = subprocess.run([HARNESS, "rmsnorm", str(m), str(n), repr(eps), str(xin), str(gpath), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ref = np.fromfile(out, np.float16)
    assert np.array_equal(_bits(ref), _bits(oracle.rmsnorm_half(x, gamma, eps)).ravel())


@needs_harness
@pytest.mark.parametrize("m,n", [(1, 768), (5, 768), (2, 2048), (3, 100)])
def test_layernorm_q_against_the_reference_source(tmp_path, oracle, m, n):
    """LayerNormQ::forward (llm/src/ops/LayerNormQ.cc:12-52) is host C++ in the reference: compiled as it is (-ffp-contract=off) and
    compared with orc_layernorm_q -- sequential fp32 sums, separate roundings, std::round."""
    rng = np.random.default_rng(m + n)
    x = (rng.standard_normal((m, n)) * 3).astype(np.float32)
    w = (rng.standard_normal(n) * 20).astype(np.float32)
    b = (rng.standard_normal(n) * 5).astype(np.float32)
    paths = []
    for name, a in (("x", x), ("w", w), ("b", b)):
        p = tmp_path / f"{name}.bin"
        a.tofile(p)
        paths.append(str(p))
    out = tmp_path / "out.bin"
    r = subprocess.run([HARNESS, "lnq", str(m), str(n), *paths, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    ref = np.fromfile(out, np.int8).reshape(m, n)
    assert np.array_equal(ref, oracle.layernorm_q(x, w, b).reshape(m, n))


@needs_harness
@pytest.mark.parametrize("heads,sq,tgz", [(12, 1, 513), (3, 5, 40), (12, 64, 64), (2, 7, 1)])
def test_opt_softmax_q_against_the_reference_sources(tmp_path, oracle, heads, sq, tgz):
    """batch_Add (llm/src/ops/batch_add.cc) and softmax (llm/src/ops/softmax.cc) are host C++ in the reference: compiled as they are, followed by
    the int8 conversion loop of Int8OPTAttention.cc:264-267, and compared with orc_opt_softmax_q bit for bit -- including the running maximum that
    starts from element [0][0][0] of the whole tensor and the double-precision quotient."""
    rng = np.random.default_rng(heads * 100 + sq * 10 + tgz)
    scores = (rng.standard_normal((heads, sq, tgz)) * 4).astype(np.float32)
    scores[0, 0, 0] = 9.5  # above most rows' own maxima: the m_data[0] quirk matters
    mask = np.zeros((sq, tgz), np.float32)
    if tgz > 1:
        mask[:, rng.integers(0, tgz, size=max(1, tgz // 5))] = np.finfo(np.float32).min
    sp, mp, out = tmp_path / "s.bin", tmp_path / "m.bin", tmp_path / "o.bin"
    scores.tofile(sp); mask.tofile(mp)
    r = subprocess.run([HARNESS, "optsm", str(heads), str(sq), str(tgz), str(sp), str(mp), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    ref = np.fromfile(out, np.int8).reshape(heads, sq, tgz)
    assert np.array_equal(ref, oracle.opt_softmax_q(scores, mask))


def _golden():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_cuda_gemv_golden", os.path.join(REPO, "tests", "golden", "make_cuda_gemv_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, np.load(os.path.join(REPO, "tests", "golden", "cuda_gemv_golden.npz"))


def _half_steps(a, b):
    ia = a.view(np.int16).astype(np.int32)
    ib = b.view(np.int16).astype(np.int32)
    ia = np.where(ia < 0, -32768 - ia, ia)
    ib = np.where(ib < 0, -32768 - ib, ib)
    return np.abs(ia - ib)


def _same_to_a_half_step(out_f16, ref32):
    """Two fp32 accumulations of the same products in different orders, each rounded to binary16 once: identical on nearly all
    outputs, one binary16 step apart on the rest -- except outputs that are tiny by cancellation (|ref| < rms / 64), where the fp32
    sums' own 1e-7-relative-to-the-terms difference spans several (tiny) binary16 steps: those are held to the absolute bound instead."""
    d = _half_steps(out_f16, ref32.astype(np.float16))
    rms = float(np.sqrt(np.mean(ref32.astype(np.float64) ** 2)))
    big = np.abs(ref32) >= rms / 64
    assert d[big].max() <= 1, int(d[big].max())
    assert np.abs(out_f16.astype(np.float64) - ref32)[~big].max(initial=0.0) <= 1e-3 * rms / 64
    assert int((d != 0).sum()) <= max(2, int(0.02 * d.size)), (int((d != 0).sum()), d.size)


@needs_harness
@pytest.mark.parametrize("M,N,K", [(1, 64, 1024), (2, 128, 4096), (1, 32, 11008), (3, 16, 256)])
def test_reference_cuda_gemv_kernel_against_the_w4a16_oracle(tmp_path, oracle, M, N, K):
    """THE hot-path kernel of the reference -- gemv_forward_cuda -> gemv_kernel_g128, kernels/cuda/gemv_cuda.cu:140-260 -- from its own
    source (block (32, 4) as concurrent threads, warp_reduce_sum through __shfl_down_sync), against orc_w4a16_gemv_q4_6: both accumulate in
    fp32 (in different orders) and round to fp16 once, so the outputs agree to one binary16 step, and nearly all are identical."""
    mod, _ = _golden()
    a, qw, sc, zp = mod.gemv_case(oracle, M, N, K, seed=M * 1000 + N + K)
    out = mod.run_reference_kernel(a, qw, sc, zp, M, N, K)
    ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, 128)
    _same_to_a_half_step(out, ref32)


def test_cuda_gemv_golden_vectors_against_the_oracle(oracle):
    """The committed golden outputs (tests/golden/cuda_gemv_golden.npz, made by the reference kernel on seeded inputs at the BASELINE decode
    shapes) against the oracle on the regenerated inputs: runs anywhere (no reference tree needed)."""
    mod, gold = _golden()
    for (M, N, K, seed) in gold["cases"][:4]:  # (the larger cases take the oracle ~10 s each: covered when the file is generated)
        a, qw, sc, zp = mod.gemv_case(oracle, int(M), int(N), int(K), int(seed))
        ref32 = oracle.w4a16_gemv_q4_6_mt(a, qw, sc, zp, int(M), int(N), int(K), 128)
        _same_to_a_half_step(gold[f"out_{M}_{N}_{K}"], ref32)
