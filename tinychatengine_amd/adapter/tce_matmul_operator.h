// tce_matmul_operator.h -- the reference's operator boundary, as seen by a QM_HIP build.
//
// TinyChatEngine has no plugin loader: llm/src/ops/** instantiate an empty `matmul::MatmulOperator` on the stack and
// call a non-virtual member with a pointer to a caller-owned `matmul_params` (reference kernels/matmul.h:52-92,
// 110-153); the backend is whichever object file defines those members at link time (llm/Makefile:64-65,86-88).
// matmul_operator_hip.cc defines them on top of the C ABI (include/tce_matmul.h).
//
// In the reference tree the adapter is compiled against the reference's OWN kernels/matmul.h (-DTCE_ADAPTER_USE_REFERENCE_HEADER;
// oracle/Makefile target `l2link` does exactly that and links the result with the reference's unmodified level-2 callers --
// tests/test_l2_link.py).  The re-declaration below exists so that libtce_matmul_operator.so and its self-test also build on
// a box without the reference tree (the GPU box): same names, field order and sizes (checked against the reference build by
// tests/test_boundary.py through tce_adapter_layout()), and the member functions the hot path needs.  float16_t is a 2-byte type in every
// reference flavour (kernels/matmul.h:12-28), so the layout is backend-independent.
#ifndef TCE_MATMUL_OPERATOR_H
#define TCE_MATMUL_OPERATOR_H

#include <stdint.h>

#ifndef TCE_ADAPTER_USE_REFERENCE_HEADER

struct tce_half_bits {  // stands in for `__half` / half_float::half: 2 bytes, never interpreted by the adapter
    uint16_t bits;
};
typedef tce_half_bits float16_t;
typedef tce_half_bits naive_float16_t;

struct quantization_params {
    float scale;
    bool per_channel = false;
    int32_t zero_point;
    int8_t q_min = -128, q_max = 127;
};

struct matrix {
    int row;
    int column;
    float *data_ptr;
    float16_t *half_data_ptr;
    naive_float16_t *fp16_data_ptr;
    int32_t *int32_data_ptr;
    int8_t *int8_data_ptr;
    uint8_t *uint8_data_ptr;
    uint8_t *int4_data_ptr;
    struct quantization_params qparams;
    int length() { return row * column; }
};

struct optimization_params {
    int blk_size;
    int num_thread = 8;
};

struct matmul_params {
    struct matrix A, B, C, bias;
    struct optimization_params opt_params;
    float alpha, beta;
    float16_t half_alpha;
    float *scales, *offset, *zero_point;  // int4, CPU layouts
    float16_t *half_scales;               // int4, q4_6 (GPU GEMV layout)
    naive_float16_t *fp16_scales;         // int4, q4_5 (AWQ GEMM layout)
    int *int32_zero_point;
    int block_size;
    float *A_scales;  // W4A8 CPU path only
    int8_t A_zero_point;
};

namespace matmul {
class MatmulOperator {
   public:
    // W4A16 (GPU layouts)
    void gemv_forward_cuda(const struct matmul_params *params);
    void naive_mat_mul_fp16_int4(const struct matmul_params *params);
    void gemm_forward_cuda(const struct matmul_params *params, int split_k_iters);
    void gemm_forward_cuda_8splits(const struct matmul_params *params, float16_t *split_8_buffer);
    void gemm_forward_cuda_half(const struct matmul_params *params, int split_k_iters);
    void gemm_forward_cuda_half_test(const struct matmul_params *params, int split_k_iters);
    // W8A8 (SmoothQuant)
    void mat_mul_accelerator_int8_fast_32unroll_over_column(const struct matmul_params *params);
    void mat_mul_accelerator_int8_fast_2x2_32unroll(const struct matmul_params *params);
    void mat_mul_accelerator_int8_fast_2x2_32unroll_nobias(const struct matmul_params *params);
    void mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_batch(const struct matmul_params *params);
    void mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32(const struct matmul_params *params);
    void mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32_batch(const struct matmul_params *params);
    void mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32(const struct matmul_params *params);
    void mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32_over_column(const struct matmul_params *params);
    // CPU-layout int4 entry points that llm/src/ops/linear.cc references in every build flavour; the CUDA backend
    // defines them as empty stubs (kernels/cuda/gemv_cuda.cu:262-268) and so does this one.
    void mat_mul_accelerator_int4_fast(const struct matmul_params *params);
    void mat_mul_accelerator_int4_fast_no_offset(const struct matmul_params *params);
};
}  // namespace matmul

#else
#include "matmul.h"  // the reference's header (with the QM_HIP branch of INTEGRATION.md)
#endif

// Per-tensor caches of the adapter (zero-point-is-8 flags, AWQ re-layouts; see matmul_operator_hip.cc): forget one buffer
// (call before freeing / rewriting it) or everything; number of live entries.
extern "C" void tce_adapter_forget(const void *ptr);
extern "C" void tce_adapter_forget_all(void);
extern "C" long tce_adapter_cache_entries(void);
// Optional load-time hook (round 5): build the packed copy of one q4_6 linear and start its zero-point check NOW (asynchronously, null stream) instead of inside the
// first gemv_forward_cuda that sees it; returns the device bytes held for it (0: shape has no packed form, or TCE_ADAPTER_PACK=0).  tce_adapter_device_bytes(): total
// device memory the adapter holds beyond the model's own tensors.
extern "C" long long tce_adapter_prepare(const void *qweight, const void *scales, const void *zeros, int N, int K, int group_size);
extern "C" long long tce_adapter_device_bytes(void);
// faults of the k-cut exchanges on the adapter's GEMM scratch area since start (synchronises the null stream; never seen other than 0; non-zero: NaN outputs, see the library header)
extern "C" long tce_adapter_gemm_faults(void);

// Layout probe used by the tests: index -> value (0 sizeof(matmul_params), 1 sizeof(matrix), 2.. offsets).
extern "C" long tce_adapter_layout(int idx);

#endif  // TCE_MATMUL_OPERATOR_H
