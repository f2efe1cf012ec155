/* cuda.h -- TEST INFRASTRUCTURE (oracle/): a host EMULATION of the few CUDA names the reference's glue kernels use, so that the
 * reference's own kernel SOURCES -- llm/src/ops/cuda/{softmax,BMM_F16T,RotaryPosEmb}.cu and the add_half / SiLuMul_half kernels
 * of llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu -- run on the CPU, thread by thread, and pin the restatements in
 * oracle/tce_oracle.c (orc_softmax_half, orc_bmm_f16t, orc_rope_half, orc_add_half, orc_silu_mul_half).  The reference has these
 * ops only as CUDA kernels; none of those five uses shared memory, __syncthreads or warp shuffles, so running the threads of a launch
 * one after the other is what the device does.  generalT5LayerNorm (llm/src/ops/cuda/LlamaRMSNorm.cu:68-115, orc_rmsnorm_half) does:
 * its blocks run through launch_concurrent(), the threads of a block as OS threads with pthread barriers for __syncthreads and for
 * each warp's __shfl_xor_sync.
 *   __global__ kernels are plain functions; blockIdx / threadIdx / blockDim / gridDim are thread-local variables that
 *   tce_emul::launch() steps through; `kernel<<<grid, block>>>(args)` in a host wrapper is rewritten by the build recipe
 *   (oracle/Makefile, target `glue`: oracle/cuda_emul/rewrite_launches.py into oracle/_ref/gen/, nothing is copied into the repository)
 *   to tce_emul::launch(tce_emul::cfg(grid, block), [&] { kernel(args); }) -- launch_concurrent for kernels that need it.
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
#ifdef __cplusplus
#include <algorithm>
#include <type_traits>
using std::max;
using std::min;
#endif

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict
#define __CUDA_ARCH__ 800 /* the device path of softmax.cu:17-23 (pre-8.6: __hgt) */
#define __launch_bounds__(...)
/* __shared__ is an empty word here: `static __shared__ T v[..]` (reduction.cuh) is then a function-local static, one object for all
 * threads, which is what a block needs as long as blocks run one after the other (they do); a NON-static `__shared__ T v;` inside a
 * kernel is rewritten to `static T v;` by the build recipe's sed pass.  __syncthreads / __shfl_xor_sync need the threads of a block to
 * run concurrently: tce_emul::launch_concurrent() below runs each thread of a block as an OS thread; under the sequential launch()
 * they abort. */
#define __shared__
#include <pthread.h>
namespace tce_emul {
struct block_state {
    pthread_barrier_t all;            /* __syncthreads */
    pthread_barrier_t warp[64];       /* one per warp of the block */
    double slot[64][32];              /* values exchanged by a warp's shuffle */
    unsigned warp_lanes[64];          /* active lanes of each warp */
    unsigned threads;
};
extern block_state *g_block;          /* null under the sequential launch() */
}  // namespace tce_emul
static inline void __syncthreads() {
    if (!tce_emul::g_block) {
        fprintf(stderr, "cuda_emul: __syncthreads() under the sequential launcher -- this kernel needs launch_concurrent()\n");
        abort();
    }
    pthread_barrier_wait(&tce_emul::g_block->all);
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

/* __shfl_xor_sync(mask, v, lane_mask, width = 32): every active lane of the warp publishes v, the warp meets, every lane reads its
 * partner's value (its own if the partner lane does not exist), the warp meets again before the slots are reused.  One-dimensional
 * blocks only (what the reference's reductions use). */
template <typename T>
static inline T __shfl_xor_sync(unsigned, T v, int lane_mask, int = 32) {
    tce_emul::block_state *b = tce_emul::g_block;
    if (!b) {
        fprintf(stderr, "cuda_emul: __shfl_xor_sync() under the sequential launcher\n");
        abort();
    }
    const unsigned lin = threadIdx.x + threadIdx.y * blockDim.x;  /* warps are formed over the linear thread index */
    const unsigned w = lin >> 5, lane = lin & 31, partner = lane ^ (unsigned)lane_mask;
    b->slot[w][lane] = (double)v;
    pthread_barrier_wait(&b->warp[w]);
    const T got = partner < b->warp_lanes[w] ? (T)b->slot[w][partner] : v;
    pthread_barrier_wait(&b->warp[w]);
    return got;
}
/* __shfl_down_sync(mask, v, delta): lane l reads lane l + delta; lanes whose source does not exist keep their own value */
template <typename T>
static inline T __shfl_down_sync(unsigned, T v, unsigned delta, int = 32) {
    tce_emul::block_state *b = tce_emul::g_block;
    if (!b) {
        fprintf(stderr, "cuda_emul: __shfl_down_sync() under the sequential launcher\n");
        abort();
    }
    const unsigned lin = threadIdx.x + threadIdx.y * blockDim.x;
    const unsigned w = lin >> 5, lane = lin & 31, src = lane + delta;
    b->slot[w][lane] = (double)v;
    pthread_barrier_wait(&b->warp[w]);
    const T got = src < b->warp_lanes[w] ? (T)b->slot[w][src] : v;
    pthread_barrier_wait(&b->warp[w]);
    return got;
}
struct float4 {
    float x, y, z, w;
};
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); } /* the device's is a 2-ulp approximation: same model as orc_rmsnorm_half */

namespace tce_emul {
struct config {
    dim3 grid, block;
};
inline config cfg(dim3 g, dim3 b, size_t = 0, cudaStream_t = nullptr) { return config{g, b}; }
inline config cfg(unsigned g, unsigned b, size_t = 0, cudaStream_t = nullptr) { return config{dim3(g), dim3(b)}; }
inline config cfg(dim3 g, unsigned b, size_t = 0, cudaStream_t = nullptr) { return config{g, dim3(b)}; }
inline config cfg(unsigned g, dim3 b, size_t = 0, cudaStream_t = nullptr) { return config{dim3(g), b}; }
/* blocks one after the other, the threads of a block as concurrent OS threads (at most 2048 per block, z = 1) */
template <typename F>
inline void launch_concurrent(const config &c, F &&body);
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
template <typename F>
struct thread_arg {
    F *body;
    tce_emul_idx bidx, tidx;
    dim3 bdim, gdim;
};
template <typename F>
void *thread_main(void *p) {
    thread_arg<F> *a = static_cast<thread_arg<F> *>(p);
    blockIdx = a->bidx;
    threadIdx = a->tidx;
    blockDim = a->bdim;
    gridDim = a->gdim;
    (*a->body)();
    return nullptr;
}
template <typename F>
inline void launch_concurrent(const config &c, F &&body) {
    const unsigned nt = c.block.x * c.block.y;
    if (c.block.z != 1 || nt == 0 || nt > 2048) {
        fprintf(stderr, "cuda_emul: launch_concurrent wants a one- or two-dimensional block of at most 2048 threads\n");
        abort();
    }
    static block_state st;
    st.threads = nt;
    const unsigned nwarps = (nt + 31) / 32;
    for (unsigned bz = 0; bz < c.grid.z; ++bz)
        for (unsigned by = 0; by < c.grid.y; ++by)
            for (unsigned bx = 0; bx < c.grid.x; ++bx) {
                pthread_barrier_init(&st.all, nullptr, nt);
                for (unsigned w = 0; w < nwarps; ++w) {
                    st.warp_lanes[w] = nt - w * 32 < 32 ? nt - w * 32 : 32;
                    pthread_barrier_init(&st.warp[w], nullptr, st.warp_lanes[w]);
                }
                g_block = &st;
                thread_arg<typename std::remove_reference<F>::type> *args = new thread_arg<typename std::remove_reference<F>::type>[nt];
                pthread_t *th = new pthread_t[nt];
                pthread_attr_t attr;
                pthread_attr_init(&attr);
                pthread_attr_setstacksize(&attr, 256 * 1024);
                for (unsigned t = 0; t < nt; ++t) {
                    args[t] = {&body, tce_emul_idx{bx, by, bz}, tce_emul_idx{t % c.block.x, t / c.block.x, 0}, c.block, c.grid};
                    if (pthread_create(&th[t], &attr, thread_main<typename std::remove_reference<F>::type>, &args[t]) != 0) {
                        fprintf(stderr, "cuda_emul: pthread_create failed\n");
                        abort();
                    }
                }
                for (unsigned t = 0; t < nt; ++t) pthread_join(th[t], nullptr);
                pthread_attr_destroy(&attr);
                delete[] th;
                delete[] args;
                g_block = nullptr;
                pthread_barrier_destroy(&st.all);
                for (unsigned w = 0; w < nwarps; ++w) pthread_barrier_destroy(&st.warp[w]);
            }
}
}  // namespace tce_emul
#endif
