// ref_shim.cc -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" entry points around the REAL reference implementation
// (matmul::MatmulOperator compiled unmodified from /root/reference/kernels by
// oracle/Makefile, -DQM_REF).  Nothing of the reference is copied here: this file
// only fills the reference's own `matmul_params` descriptor the way the
// reference's L2 wrappers do (llm/src/ops/linear.cc:80-117, llm/src/ops/cuda/
// linear.cu:43-77, llm/src/ops/W8A8B8O8Linear.cc:38-78 ...) and calls the
// reference's member functions.  The resulting library, oracle/_ref/libtce_ref.so,
// pins oracle/tce_oracle.c and produces tests/golden/*.npz.
#include <cstdint>
#include <cstring>

#include "matmul.h"  // the reference's header, found via -I/root/reference/kernels

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {
void clear(matmul_params &p) { std::memset(static_cast<void *>(&p), 0, sizeof(p)); }

void fill_int8(matmul_params &p, int M, int N, int K, const int8_t *A, const int8_t *B, float alpha, float beta, int qmin,
               int qmax) {
    clear(p);
    // W8A8B8O8Linear.cc:15-36 / :47-58 -- A [M][K], B [N][K] with B.row = k, B.column = n
    p.A.row = M;
    p.A.column = K;
    p.A.int8_data_ptr = const_cast<int8_t *>(A);
    p.B.row = K;
    p.B.column = N;
    p.B.int8_data_ptr = const_cast<int8_t *>(B);
    p.C.row = M;
    p.C.column = N;
    p.C.qparams.q_min = static_cast<int8_t>(qmin);
    p.C.qparams.q_max = static_cast<int8_t>(qmax);
    p.A.qparams.zero_point = p.B.qparams.zero_point = p.C.qparams.zero_point = 0;
    p.alpha = alpha;
    p.beta = beta;
    p.opt_params.num_thread = 1;
    p.opt_params.blk_size = 4;
}
}  // namespace

REF_API int ref_sizeof_matmul_params() { return static_cast<int>(sizeof(matmul_params)); }

// kernels/matmul_int4.cc:105-127 (generic branch; library is built with -DQM_REF only)
REF_API void ref_naive_mat_mul_int4(int M, int N, int K, int G, const float *A, const uint8_t *B, const float *scales,
                                    float zero_point, float *C) {
    matmul_params p;
    clear(p);
    p.A.row = M;
    p.A.column = K;
    p.A.data_ptr = const_cast<float *>(A);
    p.B.row = N;  // linear.cc:99-100
    p.B.column = K / 2;
    p.B.int4_data_ptr = const_cast<uint8_t *>(B);
    p.C.row = M;
    p.C.column = N;
    p.C.data_ptr = C;
    p.scales = const_cast<float *>(scales);
    p.zero_point = &zero_point;
    p.block_size = G;
    matmul::MatmulOperator op;
    op.naive_mat_mul_int4(&p);
}

REF_API void ref_naive_mat_mul_int4_with_offset(int M, int N, int K, int G, const float *A, const uint8_t *B,
                                                const float *scales, const float *offset, float zero_point, float *C) {
    matmul_params p;
    clear(p);
    p.A.row = M;
    p.A.column = K;
    p.A.data_ptr = const_cast<float *>(A);
    p.B.row = N;
    p.B.column = K / 2;
    p.B.int4_data_ptr = const_cast<uint8_t *>(B);
    p.C.row = M;
    p.C.column = N;
    p.C.data_ptr = C;
    p.scales = const_cast<float *>(scales);
    p.offset = const_cast<float *>(offset);
    p.zero_point = &zero_point;
    p.block_size = G;
    matmul::MatmulOperator op;
    op.naive_mat_mul_int4_with_offset(&p);
}

// kernels/ref/matmul_ref_int4.cc:11-38; the caller passes B.row = K/2 (linear.cc:138-139)
REF_API int ref_ref_int4_fast(int M, int N, int K, int G, int b_row, const float *A, const uint8_t *B,
                              const float *scales, const float *offset, float *C) {
    if (G != 32) return -1;  // the reference asserts
    matmul_params p;
    clear(p);
    p.A.row = M;
    p.A.column = K;
    p.A.data_ptr = const_cast<float *>(A);
    p.B.row = b_row;
    p.B.column = N;
    p.B.int4_data_ptr = const_cast<uint8_t *>(B);
    p.C.row = M;
    p.C.column = N;
    p.C.data_ptr = C;
    p.scales = const_cast<float *>(scales);
    p.offset = const_cast<float *>(offset);
    p.block_size = G;
    matmul::MatmulOperator op;
    op.mat_mul_accelerator_int4_fast(&p);
    return 0;
}

// kernels/cuda/matmul_int4.cu:8-48 (host-only code in a .cu; compiled as C++)
REF_API void ref_naive_mat_mul_fp16_int4(int M, int N, int K, int G, const uint16_t *A, const uint32_t *qweight,
                                         const uint16_t *scales, uint16_t *C) {
    matmul_params p;
    clear(p);
    p.A.row = M;
    p.A.column = K;
    p.A.fp16_data_ptr = reinterpret_cast<naive_float16_t *>(const_cast<uint16_t *>(A));
    p.B.row = K;  // linear.cu:61-62
    p.B.column = N / 8;
    p.B.int32_data_ptr = reinterpret_cast<int32_t *>(const_cast<uint32_t *>(qweight));
    p.C.row = M;
    p.C.column = N;
    p.C.fp16_data_ptr = reinterpret_cast<naive_float16_t *>(C);
    p.fp16_scales = reinterpret_cast<naive_float16_t *>(const_cast<uint16_t *>(scales));
    p.block_size = G;
    matmul::MatmulOperator op;
    op.naive_mat_mul_fp16_int4(&p);
}

// ---- int8 (kernels/ref/matmul_ref_int8.cc) ----
REF_API void ref_int8_matmul_bias_i8(int M, int N, int K, const int8_t *A, const int8_t *B, const int8_t *bias,
                                     float alpha, float beta, int qmin, int qmax, int8_t *C) {
    matmul_params p;
    fill_int8(p, M, N, K, A, B, alpha, beta, qmin, qmax);
    p.bias.row = 1;
    p.bias.column = N;
    p.bias.int8_data_ptr = const_cast<int8_t *>(bias);
    p.C.int8_data_ptr = C;
    matmul::MatmulOperator op;
    op.mat_mul_accelerator_int8_fast_2x2_32unroll(&p);
}
REF_API void ref_int8_matmul_bias_i8_over_column(int M, int N, int K, const int8_t *A, const int8_t *B,
                                                 const int8_t *bias, float alpha, float beta, int qmin, int qmax,
                                                 int8_t *C) {
    matmul_params p;
    fill_int8(p, M, N, K, A, B, alpha, beta, qmin, qmax);
    p.bias.row = 1;
    p.bias.column = N;
    p.bias.int8_data_ptr = const_cast<int8_t *>(bias);
    p.C.int8_data_ptr = C;
    matmul::MatmulOperator op;
    op.mat_mul_accelerator_int8_fast_32unroll_over_column(&p);
}
REF_API void ref_int8_matmul_nobias_i8(int M, int N, int K, const int8_t *A, const int8_t *B, float alpha, int qmin,
                                       int qmax, int8_t *C) {
    matmul_params p;
    fill_int8(p, M, N, K, A, B, alpha, 0.f, qmin, qmax);
    p.C.int8_data_ptr = C;
    matmul::MatmulOperator op;
    op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias(&p);
}
REF_API void ref_int8_matmul_nobias_batch_i8(int M, int N, int K, const int8_t *A, const int8_t *B, float alpha,
                                             int qmin, int qmax, int8_t *C) {
    matmul_params p;
    fill_int8(p, M, N, K, A, B, alpha, 0.f, qmin, qmax);
    p.C.int8_data_ptr = C;
    matmul::MatmulOperator op;
    op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_batch(&p);
}
REF_API void ref_int8_matmul_bias_f32(int M, int N, int K, const int8_t *A, const int8_t *B, const float *bias,
                                      float alpha, float *C) {
    matmul_params p;
    fill_int8(p, M, N, K, A, B, alpha, 0.f, -128, 127);
    p.bias.row = 1;
    p.bias.column = N;
    p.bias.data_ptr = const_cast<float *>(bias);
    p.C.data_ptr = C;
    matmul::MatmulOperator op;
    op.mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32(&p);
}
REF_API void ref_int8_matmul_nobias_f32(int M, int N, int K, const int8_t *A, const int8_t *B, float alpha, float *C) {
    matmul_params p;
    fill_int8(p, M, N, K, A, B, alpha, 0.f, -128, 127);
    p.C.data_ptr = C;
    matmul::MatmulOperator op;
    op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32(&p);
}
REF_API void ref_int8_matmul_nobias_batch_f32(int M, int N, int K, const int8_t *A, const int8_t *B, float alpha,
                                              float *C) {
    matmul_params p;
    fill_int8(p, M, N, K, A, B, alpha, 0.f, -128, 127);
    p.C.data_ptr = C;
    matmul::MatmulOperator op;
    op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32_batch(&p);
}

// kernels/matmul_int8.cc:8-30 (B untransposed [K][N])
REF_API void ref_naive_mat_mul_int8(int M, int N, int K, const int8_t *A, const int8_t *B, int32_t A_zp, int32_t C_zp,
                                    float A_sc, float B_sc, float C_sc, int qmin, int qmax, int8_t *C) {
    matmul_params p;
    clear(p);
    p.A.row = M;
    p.A.column = K;
    p.A.int8_data_ptr = const_cast<int8_t *>(A);
    p.B.row = K;
    p.B.column = N;
    p.B.int8_data_ptr = const_cast<int8_t *>(B);
    p.C.row = M;
    p.C.column = N;
    p.C.int8_data_ptr = C;
    p.A.qparams.zero_point = A_zp;
    p.C.qparams.zero_point = C_zp;
    p.A.qparams.scale = A_sc;
    p.B.qparams.scale = B_sc;
    p.C.qparams.scale = C_sc;
    p.C.qparams.q_min = static_cast<int8_t>(qmin);
    p.C.qparams.q_max = static_cast<int8_t>(qmax);
    matmul::MatmulOperator op;
    op.naive_mat_mul_int8(&p);
}

// kernels/matmul_imp.cc:23-35 and kernels/ref/matmul_ref_fp32.cc:11-34
REF_API void ref_fp32_matmul_transposed(int M, int N, int K, const float *A, const float *B, float *C, int use_ref_backend) {
    matmul_params p;
    clear(p);
    p.A.row = M;
    p.A.column = K;
    p.A.data_ptr = const_cast<float *>(A);
    p.B.row = use_ref_backend ? K : N;  // fp32_ref_matmul reads n from B.column; mat_mul_transposed strides by B.column
    p.B.column = use_ref_backend ? N : K;
    p.B.data_ptr = const_cast<float *>(B);
    p.C.row = M;
    p.C.column = N;
    p.C.data_ptr = C;
    matmul::MatmulOperator op;
    if (use_ref_backend)
        op.mat_mul_accelerator_transposed_fastover_column(&p);
    else
        op.mat_mul_transposed(&p);
}

// Layout of the reference's descriptor (same index convention as tce_adapter_layout in
// tinychatengine_amd/adapter/matmul_operator_hip.cc); tests/test_boundary.py compares the two tables.
#include <cstddef>
REF_API long ref_layout(int idx) {
    switch (idx) {
        case 0: return (long)sizeof(matmul_params);
        case 1: return (long)sizeof(matrix);
        case 2: return (long)offsetof(matmul_params, B);
        case 3: return (long)offsetof(matmul_params, C);
        case 4: return (long)offsetof(matmul_params, bias);
        case 5: return (long)offsetof(matmul_params, opt_params);
        case 6: return (long)offsetof(matmul_params, alpha);
        case 7: return (long)offsetof(matmul_params, beta);
        case 8: return (long)offsetof(matmul_params, half_scales);
        case 9: return (long)offsetof(matmul_params, fp16_scales);
        case 10: return (long)offsetof(matmul_params, int32_zero_point);
        case 11: return (long)offsetof(matmul_params, block_size);
        case 12: return (long)offsetof(matrix, half_data_ptr);
        case 13: return (long)offsetof(matrix, int32_data_ptr);
        case 14: return (long)offsetof(matrix, int8_data_ptr);
        case 15: return (long)offsetof(matrix, qparams);
        case 16: return (long)(offsetof(matrix, qparams) + offsetof(quantization_params, q_min));
        case 17: return (long)offsetof(matmul_params, A_scales);
        default: return -1;
    }
}
