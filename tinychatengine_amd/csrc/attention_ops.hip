// attention_ops.hip -- the two fp16 operators between the q/k/v and the o linears of the reference's Llama attention
// (SURVEY 8f rank 4), with the reference's arithmetic:
//   * BMM_F16T::forward -> mat_mul_transposed_cuda (llm/src/ops/cuda/BMM_F16T.cu:28-45): C[b][i][j] = __hmul(alpha, acc),
//     acc = __hfma(A[b][i][k], B[b][j][k], acc) for k ascending from acc = 0 -- binary16 accumulation, one rounding per
//     step.  Used for q k^T (alpha = the stored 1/sqrt(head_dim)) and for p v on the transposed V (alpha = 1):
//     Int4llamaAttention.cu:185, 211.
//   * softmax_cuda (llm/src/ops/cuda/softmax.cu:4-40): row maximum from -65504, sum = __hadd(sum, hexp(__hsub(x, max)))
//     for k ascending, out = __hdiv(hexp(__hsub(x, max)), sum).
// Both chains are sequential by definition (every step rounds to binary16), so "identical results" means walking them in
// order; what is parallel is everything around them:
//   * BMM: one LANE per output element like the reference, but a wave takes 64 consecutive j of one (b, i): the A row is a
//     wave-uniform broadcast, every lane walks its own B row with 16-byte loads; v_fma_f16 is the IEEE fused operation
//     (one rounding), fp16 denormals are on by default on gfx9.
//   * softmax: one wave per row; max and the exponentials are computed by all lanes (order-free / element-wise), staged in
//     LDS, lane 0 walks the binary16 sum, all lanes divide.
// hexp: the float exponential rounded to binary16 (the oracle's model of CUDA's hexp; see orc_softmax_half).
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

namespace {

__global__ __launch_bounds__(256) void bmm_f16t_kernel(const half_t *A, const half_t *B, half_t *C, int batch, int M, int N, int K, half_t alpha) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    const int b = blockIdx.z;
    if (j >= N) return;
    const half_t *a = A + ((size_t)b * M + i) * K;
    const half_t *w = B + ((size_t)b * N + j) * K;
    half_t acc = (half_t)0.f;
    int k = 0;
    if ((K & 7) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
        // four 16-byte pieces of each row requested before the 32 dependent fmas that consume them: the chain itself cannot be
        // shortened, but it should not also wait for one load at a time (p v at 2048 keys: 35 -> see attention_ops_decode.jsonl)
        for (; k + 32 <= K; k += 32) {
            half8_t av[4], wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                av[u] = *reinterpret_cast<const half8_t *>(a + k + 8 * u);
                wv[u] = *reinterpret_cast<const half8_t *>(w + k + 8 * u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 8; ++v) acc = __builtin_fmaf16(av[u][v], wv[u][v], acc);
        }
        for (; k < K; k += 8) {
            const half8_t av = *reinterpret_cast<const half8_t *>(a + k);
            const half8_t wv = *reinterpret_cast<const half8_t *>(w + k);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_fmaf16(av[u], wv[u], acc);
        }
    }
    for (; k < K; ++k) acc = __builtin_fmaf16(a[k], w[k], acc);
    C[((size_t)b * M + i) * N + j] = alpha * acc;  // __hmul: one rounding
}

__global__ __launch_bounds__(64) void softmax_half_kernel(const half_t *x, half_t *out, long long rows, int n) {
    extern __shared__ __attribute__((aligned(16))) half_t e_lds[];  // n halves
    const long long r = blockIdx.x;
    const int lane = threadIdx.x;
    const half_t *xr = x + r * n;
    float mx = -65504.0f;  // comparisons only: exact in any precision
    for (int k = lane; k < n; k += 64) mx = fmaxf(mx, (float)xr[k]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const half_t hmax = (half_t)mx;
    for (int k = lane; k < n; k += 64) {
        const half_t d = xr[k] - hmax;                 // __hsub
        e_lds[k] = (half_t)expf((float)d);             // hexp
    }
    __syncthreads();
    half_t sum = (half_t)0.f;
    if (lane == 0) {  // __hadd, in order; 16 bytes of exponentials per LDS read, four reads ahead of the adds
        int k = 0;
        for (; k + 32 <= n; k += 32) {
            half8_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8_t *>(e_lds + k + 8 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 8; ++w) sum = sum + v[u][w];
        }
        for (; k < n; ++k) sum = sum + e_lds[k];
    }
    sum = (half_t)__shfl((float)sum, 0, 64);  // exact round trip through float
    for (int k = lane; k < n; k += 64) out[r * n + k] = e_lds[k] / sum;  // __hdiv
}

// One decode step of one head as ONE workgroup: the reference's sequence qk_bmm -> batch_Add(mask) -> check_inf_half -> softmax
// -> pv_bmm (Int4llamaAttention.cu:184-211) with every operation and every order kept, only the launches merged:
//   scores: thread t takes keys t, t + 256, ...: hmul(alpha, chain of hfma over the head dimension), + mask, inf / nan -> -65504;
//   softmax: maximum by all threads (order-free), exponentials by all threads, the binary16 sum by thread 0 in key order,
//            quotients by all threads;
//   p v:    thread j < head_dim walks the keys in order over its row of the TRANSPOSED values (the reference transposes V for
//           this product too: value_states_transpose, :203-211), probabilities broadcast from LDS.
// q [heads][hd], K [heads][t][hd], Vt [heads][hd][t], mask [t] or null, out [heads][hd] (= the unshaped [1][heads * hd] row).
__global__ __launch_bounds__(1024) void attention_decode_kernel(const half_t *q, const half_t *K, const half_t *Vt, const half_t *mask, half_t *out, int t,
                                                                int hd, half_t alpha) {
    extern __shared__ __attribute__((aligned(16))) half_t sm[];
    half_t *sc = sm;                       // [t_pad]: scores, then probabilities
    half_t *ex = sm + ((t + 7) & ~7);      // [t_pad]: exponentials
    half_t *qs = ex + ((t + 7) & ~7);      // [hd]
    __shared__ float red[16];
    __shared__ half_t sum_sh;
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const half_t *qh = q + (size_t)h * hd;
    for (int i = tid; i < hd; i += 1024) qs[i] = qh[i];
    __syncthreads();
    const bool vec = (hd & 7) == 0 && (reinterpret_cast<uintptr_t>(K) & 15) == 0;
    float mx = -65504.0f;
    for (int k = tid; k < t; k += 1024) {
        const half_t *kr = K + ((size_t)h * t + k) * hd;
        half_t acc = (half_t)0.f;
        int d = 0;
        if (vec) {
            for (; d + 32 <= hd; d += 32) {
                half8_t kv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kv[u] = *reinterpret_cast<const half8_t *>(kr + d + 8 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const half8_t qv = *reinterpret_cast<const half8_t *>(qs + d + 8 * u);
#pragma unroll
                    for (int v = 0; v < 8; ++v) acc = __builtin_fmaf16(qv[v], kv[u][v], acc);
                }
            }
        }
        for (; d < hd; ++d) acc = __builtin_fmaf16(qs[d], kr[d], acc);
        half_t sv = alpha * acc;                       // __hmul
        if (mask) sv = sv + mask[k];                   // batch_Add_cuda: __hadd
        const float sf = (float)sv;
        if (__builtin_isinf(sf) || __builtin_isnan(sf)) sv = (half_t)-65504.0f;  // check_inf_half
        sc[k] = sv;
        mx = fmaxf(mx, (float)sv);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    const half_t hmax = (half_t)mx;
    for (int k = tid; k < t; k += 1024) ex[k] = (half_t)expf((float)(half_t)(sc[k] - hmax));  // hexp(__hsub)
    __syncthreads();
    if (tid == 0) {
        half_t sum = (half_t)0.f;
        int k = 0;
        for (; k + 32 <= t; k += 32) {
            half8_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8_t *>(ex + k + 8 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 8; ++w) sum = sum + v[u][w];
        }
        for (; k < t; ++k) sum = sum + ex[k];
        sum_sh = sum;
    }
    __syncthreads();
    const half_t sum = sum_sh;
    for (int k = tid; k < t; k += 1024) sc[k] = ex[k] / sum;  // __hdiv
    __syncthreads();
    const bool vecv = (t & 7) == 0 && (reinterpret_cast<uintptr_t>(Vt) & 15) == 0;
    for (int j = tid; j < hd; j += 1024) {
        const half_t *vr = Vt + ((size_t)h * hd + j) * t;
        half_t acc = (half_t)0.f;
        int k = 0;
        if (vecv) {
            for (; k + 32 <= t; k += 32) {
                half8_t vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) vv[u] = *reinterpret_cast<const half8_t *>(vr + k + 8 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const half8_t pv = *reinterpret_cast<const half8_t *>(sc + k + 8 * u);
#pragma unroll
                    for (int w = 0; w < 8; ++w) acc = __builtin_fmaf16(pv[w], vv[u][w], acc);
                }
            }
        }
        for (; k < t; ++k) acc = __builtin_fmaf16(sc[k], vr[k], acc);
        out[(size_t)h * hd + j] = acc;  // pv_bmm's alpha is 1: __hmul(1, acc) = acc
    }
}

// RotaryPosEmb_cuda_forward (llm/src/ops/cuda/RotaryPosEmb.cu:4-34), in place: x'[j] = hfma(x[j], cos[p][j], hmul(rot[j], sin[p][j])),
// rot = (-x[hd/2:], x[:hd/2]).  One workgroup per (head, token, q or k); the row is read completely before it is written.
__global__ __launch_bounds__(256) void rope_half_kernel(half_t *q, half_t *k, const half_t *cosv, const half_t *sinv, int len, int hd, int start_idx) {
    half_t *x = blockIdx.z ? k : q;
    if (!x) return;
    half_t *r = x + ((size_t)blockIdx.x * len + blockIdx.y) * hd;
    const half_t *c = cosv + (size_t)(blockIdx.y + start_idx) * hd, *s = sinv + (size_t)(blockIdx.y + start_idx) * hd;
    const int hp = hd / 2;
    half_t res[2];  // hd <= 512: at most two elements per thread
    int n = 0;
    for (int j = threadIdx.x; j < hd; j += 256) {
        const half_t rot = j < hp ? -r[j + hp] : r[j - hp];
        const half_t m = rot * s[j];                       // __hmul
        res[n++] = __builtin_fmaf16(r[j], c[j], m);        // __hfma
    }
    __syncthreads();
    n = 0;
    for (int j = threadIdx.x; j < hd; j += 256) r[j] = res[n++];
}

}  // namespace

int launch_rope_half(void *q, void *k, const void *cosv, const void *sinv, int heads, int len, int hd, int start_idx, hipStream_t stream, hipError_t *hip_err) {
    hipLaunchKernelGGL(rope_half_kernel, dim3(heads, len, 2), dim3(256), 0, stream, static_cast<half_t *>(q), static_cast<half_t *>(k), static_cast<const half_t *>(cosv),
                       static_cast<const half_t *>(sinv), len, hd, start_idx);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

int launch_attention_decode(const void *q, const void *K, const void *Vt, const void *mask, void *out, int heads, int t, int hd, unsigned short alpha_bits,
                            hipStream_t stream, hipError_t *hip_err) {
    half_t alpha;
    __builtin_memcpy(&alpha, &alpha_bits, 2);
    const size_t lds = ((size_t)2 * ((t + 7) & ~7) + hd) * 2;
    if (lds > 150 * 1024) return TCE_ERR_UNSUPPORTED_SHAPE;
    auto kfn = attention_decode_kernel;
    if (lds > 64 * 1024) {  // per launch: the attribute is per device, and a process may drive several (no cached "already raised" flag)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            if (hip_err) *hip_err = e;
            return TCE_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kfn, dim3(heads), dim3(1024), lds, stream, static_cast<const half_t *>(q), static_cast<const half_t *>(K), static_cast<const half_t *>(Vt),
                       static_cast<const half_t *>(mask), static_cast<half_t *>(out), t, hd, alpha);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

int launch_bmm_f16t(const void *A, const void *B, void *C, int batch, int M, int N, int K, unsigned short alpha_bits, hipStream_t stream, hipError_t *hip_err) {
    half_t alpha;
    __builtin_memcpy(&alpha, &alpha_bits, 2);
    dim3 grid((N + 255) / 256, M, batch);
    hipLaunchKernelGGL(bmm_f16t_kernel, grid, dim3(256), 0, stream, static_cast<const half_t *>(A), static_cast<const half_t *>(B), static_cast<half_t *>(C), batch, M, N, K, alpha);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

int launch_softmax_half(const void *x, void *out, long long rows, int n, hipStream_t stream, hipError_t *hip_err) {
    hipLaunchKernelGGL(softmax_half_kernel, dim3((unsigned)rows), dim3(64), (size_t)n * 2, stream, static_cast<const half_t *>(x), static_cast<half_t *>(out), rows, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
