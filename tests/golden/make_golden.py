"""Generate tests/golden/hotpath_golden.npz from the REFERENCE ITSELF.  Runs only in the build container
(/root/reference present); the .npz is committed so that the GPU box (no /root/reference) can check against it.

Sources of truth used here:
  * weight formats: the reference's own Python quantizer, imported from /root/reference/llm/tools/quantize_methods.py
    (quantize_row_q4_6 / quantize_row_q4_5);
  * arithmetic: oracle/_ref/libtce_ref.so = the reference's kernels/{matmul_imp,matmul_int4,matmul_int8}.cc,
    kernels/ref/*.cc and kernels/cuda/matmul_int4.cu compiled unmodified (oracle/Makefile).
Nothing from oracle/tce_oracle.c is used to produce expected values.

    python tests/golden/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/llm/tools")

from oracle.oracle import Reference, build  # noqa: E402

ALPHA, BETA = 0.0005035400390625, 0.02130126953125  # llm/tests/non_cuda/test_ops.cc:179


def ref_quantize(fn_name, w, group):
    import quantize_methods as qm
    assert group == 128, "the reference's q4_5/q4_6 writers are hard-wired to QK=128 (quantize_constants.py)"
    n, k = w.shape
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(np.ascontiguousarray(w, np.float32).tobytes())
        path = f.name
    try:
        qs, d, m, zp = getattr(qm, fn_name)(path, n * k, "fp32", k, n)
    finally:
        os.unlink(path)
    return (np.ascontiguousarray(qs).view(np.uint32), np.ascontiguousarray(np.asarray(d, dtype=np.float16)),
            np.ascontiguousarray(zp).astype(np.int32).view(np.uint32))


def unpack_q4_6(qw, n, k):
    sh = np.arange(8, dtype=np.uint32) * 4
    return ((qw[:, :, None] >> sh) & 0xF).astype(np.uint8).reshape(n, k)


def main():
    build(with_ref=True)
    ref = Reference()
    rng = np.random.default_rng(20240807)
    out = {}

    # ---- W4A16 on q4_6: K=1408 -> 11 groups, zeros width 2 (padded scale rows), M=3 ----
    N, K, G, M = 48, 1408, 128, 3
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    w[5, 128:256] = 0.0  # an all-zero group (d == 0 branch of the quantizer)
    qw, sc, zp = ref_quantize("quantize_row_q4_6", w, G)
    a16 = rng.standard_normal((M, K)).astype(np.float16)
    codes = unpack_q4_6(qw.reshape(N, K // 8), N, K)
    seq = (codes[:, 0::2] | (codes[:, 1::2] << 4)).astype(np.uint8)  # kernels/matmul_int4.cc:116-120
    s32 = sc.reshape(N, -1)[:, : K // G].astype(np.float32)
    exp = ref.naive_mat_mul_int4(a16.astype(np.float32), seq, s32, 8.0, M, N, K, G)
    out.update(w4_w=w, w4_qweight=qw.reshape(N, K // 8), w4_scales=sc.reshape(N, -1), w4_zeros=zp.reshape(N, -1), w4_a=a16,
               w4_expected_f32=exp, w4_dims=np.array([M, N, K, G]))

    # ---- AWQ q4_5 + binary16 arithmetic ----
    N, K, G, M = 64, 256, 128, 2
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    q5, s5, z5 = ref_quantize("quantize_row_q4_5", w, G)
    a16 = rng.standard_normal((M, K)).astype(np.float16)
    exp16 = ref.naive_mat_mul_fp16_int4(a16, q5.reshape(K, N // 8), s5.reshape(K // G, N), M, N, K, G)
    out.update(awq_w=w, awq_qweight=q5.reshape(K, N // 8), awq_scales=s5.reshape(K // G, N), awq_zeros=z5.reshape(K // G, N // 8),
               awq_a=a16, awq_expected_f16=exp16, awq_dims=np.array([M, N, K, G]))

    # ---- W8A8: all variants, with -128 corners and values that land on .5 ties ----
    M, N, K = 20, 48, 192
    A = rng.integers(-128, 128, (M, K), dtype=np.int8)
    B = rng.integers(-128, 128, (N, K), dtype=np.int8)
    A[0, :] = -128
    B[0, :] = -128  # acc = K * 16384
    bias8 = rng.integers(-128, 128, N, dtype=np.int8)
    biasf = rng.standard_normal(N).astype(np.float32)
    Bb = rng.integers(-128, 128, (M, N, K), dtype=np.int8)
    out.update(i8_A=A, i8_B=B, i8_bias8=bias8, i8_biasf=biasf, i8_Bb=Bb, i8_dims=np.array([M, N, K]),
               i8_alpha_beta=np.array([ALPHA, BETA], np.float32))
    out["i8_bias_i8"] = ref.int8_matmul_bias_i8(A, B, bias8, ALPHA, BETA, -128, 127, M, N, K)
    out["i8_bias_i8_relu"] = ref.int8_matmul_bias_i8(A, B, bias8, ALPHA, BETA, 0, 127, M, N, K)
    out["i8_bias_i8_over_column"] = ref.int8_matmul_bias_i8(A, B, bias8, ALPHA, BETA, -128, 127, M, N, K, over_column=True)
    out["i8_nobias_i8"] = ref.int8_matmul_nobias_i8(A, B, ALPHA, -128, 127, M, N, K)
    out["i8_nobias_batch_i8"] = ref.int8_matmul_nobias_i8(A, Bb, ALPHA, -128, 127, M, N, K, batch=True)
    out["i8_bias_f32"] = ref.int8_matmul_bias_f32(A, B, biasf, ALPHA, M, N, K)
    out["i8_nobias_f32"] = ref.int8_matmul_nobias_f32(A, B, ALPHA, M, N, K)
    out["i8_nobias_batch_f32"] = ref.int8_matmul_nobias_f32(A, Bb, ALPHA, M, N, K, batch=True)
    # exact .5 ties: alpha = 0.5, K = 1 products -> acc*0.5 in {.., -1.5, -0.5, 0.5, 1.5, ..}
    At = np.arange(-8, 8, dtype=np.int8).reshape(16, 1)
    Bt = np.array([[1], [3], [-1], [-3]], dtype=np.int8)
    out.update(i8_tie_A=At, i8_tie_B=Bt)
    out["i8_tie_nobias_i8"] = ref.int8_matmul_nobias_i8(At, Bt, 0.5, -128, 127, 16, 4, 1)
    # zero-point form (kernels/matmul_int8.cc)
    out["i8_naive"] = ref.naive_mat_mul_int8(A, np.ascontiguousarray(B.T), 3, -2, 0.02, 0.01, 0.05, -128, 127, M, N, K)

    # ---- fp32 helpers ----
    Af = rng.standard_normal((5, 64)).astype(np.float32)
    Bf = rng.standard_normal((7, 64)).astype(np.float32)
    out.update(f32_A=Af, f32_B=Bf, f32_expected=ref.fp32_matmul_transposed(Af, Bf, 5, 7, 64, use_ref_backend=True),
               f32_expected_imp=ref.fp32_matmul_transposed(Af, Bf, 5, 7, 64, use_ref_backend=False))

    out["sizeof_matmul_params"] = np.array([ref.sizeof_matmul_params()])
    path = os.path.join(HERE, "hotpath_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
