/* cuda.h -- TEST INFRASTRUCTURE (oracle/): a host EMULATION of the few CUDA names the reference's glue kernels use, so that the
 * reference's own kernel SOURCES -- llm/src/ops/cuda/{softmax,BMM_F16T,RotaryPosEmb}.cu and the add_half / SiLuMul_half kernels
 * of llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu -- run on the CPU, thread by thread, and pin the restatements in
 * oracle/tce_oracle.c (orc_softmax_half, orc_bmm_f16t, orc_rope_half, orc_add_half, orc_silu_mul_half).  The reference has these
 * ops only as CUDA kernels; none of them uses shared memory, __syncthreads or warp shuffles, so running the threads of a launch one
 * after the other is what the device does.  (generalT5LayerNorm does use them and stays unpinned.)
 *   __global__ kernels are plain functions; blockIdx / threadIdx / blockDim / gridDim are thread-local variables that
 *   tce_emul::launch() steps through; `kernel<<<grid, block>>>(args)` in a host wrapper is rewritten by the build recipe
 *   (oracle/Makefile, target `glue`: a sed pass into oracle/_ref/gen/, nothing is copied into the repository) to
 *   tce_emul::launch(tce_emul::cfg(grid, block), [&] { kernel(args); }).
 *   `half` arithmetic: every intrinsic is ONE correctly rounded binary16 operation (the reference's definition of __hadd, __hmul,
 *   __hfma, __hdiv); the rounding and the exact fused multiply-add come from libtce_oracle.so (orc_f64_to_f16, orc_hfma -- pinned
 *   against exact rational arithmetic in tests/test_oracle.py).  hexp is the C library's expf rounded to binary16, the same
 *   model the restatements use: the kernels' STRUCTURE and ORDER are what this pins, not CUDA's exponential.
 * Not part of the product; never included by tinychatengine_amd/. */
#ifndef TCE_ORACLE_CUDA_EMUL_H
#define TCE_ORACLE_CUDA_EMUL_H
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict
#define __CUDA_ARCH__ 800 /* the device path of softmax.cu:17-23 (pre-8.6: __hgt) */
#define __launch_bounds__(...)
/* kernels that use shared memory or barriers are NOT run by this emulation (threads run one after the other); the names exist so that
 * the translation units that also contain such kernels compile -- calling one aborts */
#define __shared__
static inline void __syncthreads() {
    fprintf(stderr, "cuda_emul: __syncthreads() -- a kernel that needs real thread concurrency was called\n");
    abort();
}

typedef int cudaError_t;
typedef void *cudaStream_t;
typedef void *cudaEvent_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3, cudaMemcpyDefault = 4 };
static inline const char *cudaGetErrorString(cudaError_t) { return "cuda emulation"; }
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline cudaError_t cudaMalloc(void **p, size_t n) { *p = malloc(n); return *p ? 0 : 2; }
template <typename T> static inline cudaError_t cudaMalloc(T **p, size_t n) { *p = (T *)malloc(n); return *p ? 0 : 2; }
template <typename T> static inline cudaError_t cudaMallocManaged(T **p, size_t n) { *p = (T *)malloc(n); return *p ? 0 : 2; }
template <typename T> static inline cudaError_t cudaMallocHost(T **p, size_t n) { *p = (T *)malloc(n); return *p ? 0 : 2; }
static inline cudaError_t cudaFree(void *p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaFreeHost(void *p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void *d, const void *s, size_t n, cudaMemcpyKind) { memmove(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemcpyAsync(void *d, const void *s, size_t n, cudaMemcpyKind, cudaStream_t = nullptr) { memmove(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemset(void *d, int v, size_t n) { memset(d, v, n); return cudaSuccess; }
static inline cudaError_t cudaEventCreate(cudaEvent_t *) { return cudaSuccess; }
static inline cudaError_t cudaEventRecord(cudaEvent_t, cudaStream_t = nullptr) { return cudaSuccess; }
static inline cudaError_t cudaEventSynchronize(cudaEvent_t) { return cudaSuccess; }
static inline cudaError_t cudaEventElapsedTime(float *ms, cudaEvent_t, cudaEvent_t) { *ms = 0.f; return cudaSuccess; }
static inline cudaError_t cudaEventDestroy(cudaEvent_t) { return cudaSuccess; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct tce_emul_idx {
    unsigned x, y, z;
};
extern thread_local tce_emul_idx blockIdx, threadIdx;
extern thread_local dim3 blockDim, gridDim;

namespace tce_emul {
struct config {
    dim3 grid, block;
};
inline config cfg(dim3 g, dim3 b, size_t = 0, cudaStream_t = nullptr) { return config{g, b}; }
inline config cfg(unsigned g, unsigned b, size_t = 0, cudaStream_t = nullptr) { return config{dim3(g), dim3(b)}; }
inline config cfg(dim3 g, unsigned b, size_t = 0, cudaStream_t = nullptr) { return config{g, dim3(b)}; }
inline config cfg(unsigned g, dim3 b, size_t = 0, cudaStream_t = nullptr) { return config{dim3(g), b}; }
/* every thread of every block, one after the other: valid for kernels without __syncthreads / shared memory / shuffles */
template <typename F>
inline void launch(const config &c, F &&body) {
    gridDim = c.grid;
    blockDim = c.block;
    for (unsigned bz = 0; bz < c.grid.z; ++bz)
        for (unsigned by = 0; by < c.grid.y; ++by)
            for (unsigned bx = 0; bx < c.grid.x; ++bx)
                for (unsigned tz = 0; tz < c.block.z; ++tz)
                    for (unsigned ty = 0; ty < c.block.y; ++ty)
                        for (unsigned tx = 0; tx < c.block.x; ++tx) {
                            blockIdx = tce_emul_idx{bx, by, bz};
                            threadIdx = tce_emul_idx{tx, ty, tz};
                            body();
                        }
}
}  // namespace tce_emul
#endif
