/* see cuda.h in this directory: test-only include shim.  `half` is the 2-byte storage type of kernels/matmul.h:17
 * (`typedef half float16_t`); the host sources compiled with this shim never do arithmetic on it. */
#ifndef TCE_ORACLE_CUDA_FP16_SHIM_H
#define TCE_ORACLE_CUDA_FP16_SHIM_H
#include "cuda.h"
struct half {
    uint16_t x;
};
struct half2 {
    half x, y;
};
#endif
