// ref_isa_shim.cc -- TEST INFRASTRUCTURE ONLY.
// The reference's naive_mat_mul_int4 compiled under ANOTHER of its per-ISA flavours (-DQM_ARM: kernels/matmul_int4.cc:50-76; -DQM_METAL: :16-49), each a different
// nibble order of the same int4 weights, with strict floating-point flags -- so that oracle/tce_oracle.c's restatements of those branches can be pinned bit for bit
// (the x86 flavour has its own shim, ref_x86_shim.cc).  Built twice by oracle/Makefile with -DTCE_REF_ISA_SYMBOL=ref_naive_mat_mul_int4_<isa>.
#include <cstdint>
#include <cstring>

#include "matmul.h"

extern "C" __attribute__((visibility("default"))) void TCE_REF_ISA_SYMBOL(int M, int N, int K, int G, const float *A, const uint8_t *B, const float *scales, float *C) {
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
