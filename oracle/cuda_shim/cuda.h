/* cuda.h -- TEST INFRASTRUCTURE (oracle/): lets the reference's UNMODIFIED host sources (llm/src/ops/cuda/linear.cu,
 * llm/src/ops/W8A8*.cc, BMM_S8T*.cc, llm/src/utils.cc) compile as plain host C++ under their own -DQM_CUDA flavour,
 * so that tests/test_l2_link.py can link them against the HIP adapter.  Nothing here executes: the sources only need the
 * names their headers mention (kernels/matmul.h:14-18, llm/include/utils.h:60-96, llm/include/operators.h:37-49).
 * Not part of the product; never included by tinychatengine_amd/. */
#ifndef TCE_ORACLE_CUDA_SHIM_H
#define TCE_ORACLE_CUDA_SHIM_H
#include <stdint.h>
#ifndef __global__
#define __global__
#endif
#ifndef __device__
#define __device__
#endif
#ifndef __host__
#define __host__
#endif
#ifndef __forceinline__
#define __forceinline__ inline
#endif
typedef int cudaError_t;
enum { cudaSuccess = 0 };
static inline const char *cudaGetErrorString(cudaError_t) { return "cuda shim"; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
#endif
