// w4a16_awq.hip -- the AWQ "CUDA GEMM" (q4_5) weight layout on gfx950.
//
//  * awq_fp16acc_kernel reproduces MatmulOperator::naive_mat_mul_fp16_int4 (reference kernels/cuda/matmul_int4.cu:8-48)
//    bit for bit: zero point 8, every operation rounded to binary16 (the reference uses half_float 2.2.0, one RNE
//    rounding per operator), strictly sequential over k.  One thread per output; it is a test/compat entry point
//    exactly like the reference's, not a fast path.  This file is compiled with -ffp-contract=off so the product and
//    the sum are two roundings.
//  * awq_repack_kernel rewrites q4_5 (u32 [K][N/8], nibble order 0 2 4 6 1 3 5 7 along n; scales fp16 [K/G][N]) into
//    q4_6 (u32 [N][K/8] sequential along k; scales fp16 [N][zw*8]; zeros 0x88888888) so that the fast kernels
//    (w4a16_gemv.hip / w4a16_gemm.hip) serve the declared-but-undefined gemm_forward_cuda* surface
//    (reference kernels/matmul.h:140-145).
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

namespace {

__device__ __forceinline__ int awq_shift(int n) {
    // n % 8 -> bit offset inside the word (kernels/cuda/matmul_int4.cu:23-39: order 0 2 4 6 1 3 5 7)
    const int r = n & 7;
    return ((r & 1) ? 16 : 0) + (r >> 1) * 4;
}

__global__ __launch_bounds__(256) void awq_fp16acc_kernel(int M, int N, int K, int G, const half_t *__restrict__ A,
                                                          const unsigned *__restrict__ qw, const half_t *__restrict__ scales,
                                                          half_t *__restrict__ C) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    const int sh = awq_shift(n);
    const int nw = N >> 3;
    half_t acc = (half_t)0.0f;
    for (int k = 0; k < K; ++k) {
        const half_t s = scales[(size_t)(k / G) * N + n];
        const half_t in = A[(size_t)m * K + k];
        const unsigned word = qw[(size_t)k * nw + (n >> 3)];
        const half_t qz = (half_t)(int)((word >> sh) & 0xFu) - (half_t)8.0f;  // exact
        const half_t w = qz * s;                                            // one rounding
        const half_t prod = in * w;                                         // one rounding
        acc = acc + prod;                                                   // one rounding
    }
    C[(size_t)m * N + n] = acc;
}

__global__ __launch_bounds__(256) void awq_repack_kernel(int N, int K, int G, int zw, const unsigned *__restrict__ q5,
                                                         const half_t *__restrict__ s5, unsigned *__restrict__ q6,
                                                         half_t *__restrict__ s6, unsigned *__restrict__ z6) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int kw = K >> 3;
    const long long n_words = (long long)N * kw;
    if (idx < n_words) {
        const int n = (int)(idx / kw), j = (int)(idx % kw);
        const int sh = awq_shift(n);
        const int nw = N >> 3;
        unsigned out = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned word = q5[(size_t)(8 * j + i) * nw + (n >> 3)];
            out |= ((word >> sh) & 0xFu) << (4 * i);
        }
        q6[idx] = out;
    }
    const int sfw = zw * 8;
    if (idx < (long long)N * sfw) {
        const int n = (int)(idx / sfw), g = (int)(idx % sfw);
        s6[idx] = g < K / G ? s5[(size_t)g * N + n] : (half_t)0.0f;
    }
    if (idx < (long long)N * zw) z6[idx] = 0x88888888u;
}

__device__ int g_zero_mismatch;

__global__ __launch_bounds__(256) void zeros_check_kernel(const unsigned *__restrict__ z, long long n_words) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words && z[i] != 0x88888888u) g_zero_mismatch = 1;
}

// One workgroup walks the whole tensor (at most a few MB, once per tensor) and publishes the verdict itself: no second launch, no counter to zero.
__global__ __launch_bounds__(1024) void zeros_check_async_kernel(const unsigned *__restrict__ z, long long n_words, int *verdict) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int mine = 0;
    const long long n4 = (reinterpret_cast<uintptr_t>(z) & 15) == 0 ? n_words >> 2 : 0;
    const uint4_t *z4 = reinterpret_cast<const uint4_t *>(z);
    for (long long i = threadIdx.x; i < n4; i += 1024) {
        const uint4_t v = z4[i];
        mine |= (v[0] != 0x88888888u) | (v[1] != 0x88888888u) | (v[2] != 0x88888888u) | (v[3] != 0x88888888u);
    }
    for (long long i = n4 * 4 + threadIdx.x; i < n_words; i += 1024) mine |= z[i] != 0x88888888u;
    if (mine) bad = 1;  // (benign race: every writer stores 1)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(verdict, bad ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

// 1 if every packed zero-point word is 0x88888888, 0 if not, negative on a HIP error.  Synchronous (used once per
// weight tensor, at load time).
int check_zero_point_8(const void *zeros, long long n_words, hipError_t *hip_err) {
    int zero = 0, out = 0;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_zero_mismatch), &zero, sizeof(int));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(zeros_check_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, nullptr,
                           static_cast<const unsigned *>(zeros), n_words);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyFromSymbol(&out, HIP_SYMBOL(g_zero_mismatch), sizeof(int));
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return out ? 0 : 1;
}

int check_zero_point_8_async(const void *zeros, long long n_words, int *verdict, hipStream_t stream, hipError_t *hip_err) {
    hipLaunchKernelGGL(zeros_check_async_kernel, dim3(1), dim3(1024), 0, stream, static_cast<const unsigned *>(zeros), n_words, verdict);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

int launch_awq_fp16acc(int M, int N, int K, int G, const void *A, const void *qweight, const void *scales, void *C,
                       hipStream_t stream, hipError_t *hip_err) {
    const long long total = (long long)M * N;
    hipLaunchKernelGGL(awq_fp16acc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, M, N, K, G,
                       static_cast<const half_t *>(A), static_cast<const unsigned *>(qweight),
                       static_cast<const half_t *>(scales), static_cast<half_t *>(C));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

// workspace layout: [q6: N*K/8 u32][z6: N*zw u32][s6: N*zw*8 fp16]   (all 16-byte aligned when N*K/8 % 4 == 0)
int launch_awq_repack(int N, int K, int G, const void *qweight, const void *scales, void *workspace, hipStream_t stream,
                      hipError_t *hip_err) {
    const int zw = zeros_width(K, G);
    unsigned *q6 = static_cast<unsigned *>(workspace);
    unsigned *z6 = q6 + (size_t)N * (K / 8);
    half_t *s6 = reinterpret_cast<half_t *>(z6 + (size_t)N * zw);
    const long long total = (long long)N * (K / 8);  // >= N*zw*8 and >= N*zw because K/8 >= zw*8 only when K >= 64*zw...
    long long span = total;
    if ((long long)N * zw * 8 > span) span = (long long)N * zw * 8;
    hipLaunchKernelGGL(awq_repack_kernel, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, stream, N, K, G, zw,
                       static_cast<const unsigned *>(qweight), static_cast<const half_t *>(scales), q6, s6, z6);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
