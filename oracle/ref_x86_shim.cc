// ref_x86_shim.cc -- TEST INFRASTRUCTURE ONLY.
// The reference's naive_mat_mul_int4 compiled with -DQM_x86 (kernels/matmul_int4.cc:78-104: the x86-interleaved nibble
// order, q4_3) with strict floating-point flags, so that oracle/tce_oracle.c's restatement of that branch can be pinned
// and the timed AVX W4A8 baseline can be sanity-checked against it the way the reference's own test does
// (llm/tests/non_cuda/test_ops.cc:648-653, MSE <= 7e-4).
#include <cstdint>
#include <cstring>

#include "matmul.h"

extern "C" __attribute__((visibility("default"))) void ref_naive_mat_mul_int4_x86(int M, int N, int K, int G, const float *A,
                                                                                  const uint8_t *B, const float *scales,
                                                                                  float *C) {
    matmul_params p;
    std::memset(static_cast<void *>(&p), 0, sizeof(p));
    float zp = 8.0f;
    p.A.row = M;
    p.A.column = K;
    p.A.data_ptr = const_cast<float *>(A);
    p.B.row = N;  // llm/src/ops/linear.cc:99-100
    p.B.column = K / 2;
    p.B.int4_data_ptr = const_cast<uint8_t *>(B);
    p.C.row = M;
    p.C.column = N;
    p.C.data_ptr = C;
    p.scales = const_cast<float *>(scales);
    p.zero_point = &zp;
    p.block_size = G;
    matmul::MatmulOperator op;
    op.naive_mat_mul_int4(&p);
}
