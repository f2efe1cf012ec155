// ref_avx_shim.cc -- TEST / BENCH-BASELINE INFRASTRUCTURE ONLY.
//
// extern "C" entry point around the reference's x86 fast path
// MatmulOperator::mat_mul_accelerator_int8_int4_fast_no_offset
// (kernels/avx/matmul_avx_int8_int4.cc:325-357), compiled unmodified from
// /root/reference by oracle/Makefile with the reference's own x86 flags
// (llm/Makefile:86-88).  It is W4A8 / group 32 / QM_x86-interleaved weights, i.e.
// NOT the same arithmetic as the W4A16 GPU path: it is only the *timed CPU
// baseline* reported next to GPU numbers, never a parity oracle.
// The descriptor is filled the way Linear_FP_int4::forward does
// (llm/src/ops/linear.cc:187-217).
#include <cstdint>
#include <cstring>

#include "matmul.h"

extern "C" __attribute__((visibility("default"))) int ref_avx_w4a8_g32(int M, int N, int K, int num_thread, float *A,
                                                                       uint8_t *B_q4_3, float *scales, float *offset,
                                                                       int8_t *A_int8_scratch, float *A_scales_scratch,
                                                                       float *C) {
    if (K % 64 != 0) return -1;
    matmul_params p;
    std::memset(static_cast<void *>(&p), 0, sizeof(p));
    p.A.row = M;
    p.A.column = K;
    p.A.data_ptr = A;
    p.A.int8_data_ptr = A_int8_scratch;
    p.A_scales = A_scales_scratch;
    p.B.row = K / 2;
    p.B.column = N;
    p.B.int4_data_ptr = B_q4_3;
    p.C.row = M;
    p.C.column = N;
    p.C.data_ptr = C;
    p.bias.data_ptr = nullptr;
    p.opt_params.num_thread = num_thread;  // NB: the reference creates its static pool on the FIRST call only
    p.scales = scales;
    p.offset = offset;
    p.block_size = 32;
    matmul::MatmulOperator op;
    op.mat_mul_accelerator_int8_int4_fast_no_offset(&p);
    return 0;
}
