// w8a8_lnq_fused.hip -- LayerNormQ and the int8 linears that read its output as ONE launch, for decode (m <= 8 rows).
//
// SURVEY section 8f rank 3.  The reference's OPT decoder layer runs, per token,
//     self_attn_layer_norm.forward(hidden_states, hidden_states_int8)              llm/src/ops/LayerNormQ.cc:12-52
//     q_proj.forward / k_proj.forward / v_proj.forward(hidden_states_int8, ...)    llm/src/nn_modules/Int8OPTAttention.cc:186-201
//                                                                                  -> W8A8B8O8Linear::forward, W8A8B8O8Linear.cc:38-78
//     final_layer_norm.forward(...) ; fc1.forward(...)                             llm/src/nn_modules/Int8OPTDecoderLayer.cc (LayerNormQ + W8A8B8O8LinearReLU)
// i.e. four / two launches whose work is ~1 us of memory traffic at OPT-125M sizes (768 x 768 int8) behind ~4 us of launch
// latency each.  Here every workgroup recomputes the (tiny) normalisation for its own use and then produces its slice of the
// output rows of up to TCE_MAX_GROUP linears: one launch.
//
// BIT-EXACT against LayerNormQ::forward followed by int8_ref_matmul (kernels/ref/matmul_ref_int8.cc:11-35):
//   * the reference's two row sums are sequential fp32 additions, so they are added in order (every lane holds the same accumulator and
//     gets the values by LDS broadcast reads, one dependent add per element; a wave per row for m > 1); the division, multiply and add of the output are separate roundings (-ffp-contract=off), std::round = half away
//     from zero -- the same code as tce_layernorm_q (glue.hip);
//   * the dot products are int32 (v_dot4_i32_i8), exact in any order;
//   * the epilogue is the int8 path's: (float)acc, * alpha, + (float)bias * beta, each rounded separately, roundf, clamp.
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

namespace {

struct LnqLinear {
    const int8_t *B;   // int8 [N][K]
    const void *bias;  // int8 [N] / fp32 [N] / null
    void *C;           // int8 [m][N] / fp32 [m][N]
    int N;
    float alpha, beta;
    int q_min, q_max, bias_kind, out_kind;
    int row_begin;  // first row of this linear in the launch's concatenated row space
};

struct LnqArgs {
    const float *x, *ln_w, *ln_b;
    int8_t *ln_out;  // optional: the normalised int8 rows [m][K]
    int m, K, count, total_rows;
    LnqLinear lin[TCE_MAX_GROUP];
};

constexpr int kRowsPerWave = 2;
constexpr int kWaves = 4;
constexpr int kMaxM = 8;

// LDS: [m][K] fp32 rows | [m][K] int8 | (pad) [m][2] stats | (pad) [min(m, kWaves)][K] fp32 squared deviations
__host__ __device__ constexpr size_t lnq_dev_offset(int m, int K) { return (((size_t)m * K * 5 + 16 + (size_t)m * 8 + 16) + 15) & ~(size_t)15; }

__global__ __launch_bounds__(64 * kWaves) void lnq_w8a8_kernel(const LnqArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, m = a.m;
    float *rows = reinterpret_cast<float *>(smem);                     // [m][K] fp32
    int8_t *q8 = reinterpret_cast<int8_t *>(smem + (size_t)m * K * 4);  // [m][K] int8
    float *stats = reinterpret_cast<float *>(smem + (size_t)m * K * 5 + 16 - ((size_t)m * K * 5) % 16);  // [m][2]: mean, std
    const int row0 = (blockIdx.x * kWaves + wave) * kRowsPerWave;
    const int pieces = K >> 4;
    // the first weight pieces of BOTH rows of the wave are requested before anything else: they do not depend on the normalisation and land while
    // the row sums are walked (K = 768: one piece per lane and row is the whole row)
    int4_t wfirst[kRowsPerWave][2];
    float ufirst[kRowsPerWave];
#pragma unroll
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
        const int row = row0 + rr < a.total_rows ? row0 + rr : a.total_rows - 1;
        int li = 0;
#pragma unroll
        for (int s = 1; s < TCE_MAX_GROUP; ++s)
            if (s < a.count && row >= a.lin[s].row_begin) li = s;
        const int4_t *brow = reinterpret_cast<const int4_t *>(a.lin[li].B + (size_t)(row - a.lin[li].row_begin) * K);
#pragma unroll
        for (int i = 0; i < 2; ++i) wfirst[rr][i] = lane + 64 * i < pieces ? brow[lane + 64 * i] : int4_t{0, 0, 0, 0};
        // the additive term of this output column, once (kernels/ref/matmul_ref_int8.cc:29-31)
        const LnqLinear &L = a.lin[li];
        const int n = row - L.row_begin;
        ufirst[rr] = 0.f;
        if (L.bias_kind == TCE_BIAS_INT8) ufirst[rr] = __fmul_rn((float)static_cast<const int8_t *>(L.bias)[n], L.beta);
        else if (L.bias_kind == TCE_BIAS_FP32) ufirst[rr] = static_cast<const float *>(L.bias)[n];
    }
    // ... and the affine parameters of the elements this thread will quantize in step 3 (m * K <= 1024: decode at OPT sizes)
    constexpr int PF = 4;
    const bool affine_prefetched = m * K <= 64 * kWaves * PF;
    float pw[PF], pb[PF];
    if (affine_prefetched) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int e = tid + 64 * kWaves * i;
            const int k = e < m * K ? e % K : 0;
            pw[i] = a.ln_w[k];
            pb[i] = a.ln_b[k];
        }
    }
    // ---- 1. the rows into LDS (coalesced) ----
    const int n4 = (m * K) >> 2;
    for (int p = tid; p < n4; p += 64 * kWaves) reinterpret_cast<float4_t *>(rows)[p] = reinterpret_cast<const float4_t *>(a.x)[p];
    __syncthreads();
    // ---- 2. the two sequential sums of every row (LayerNormQ.cc:27-37), in the reference's order: wave w takes rows w, w + 4
    //         (sequential_sum_bcast, tce_common.hpp: one dependent add per element) ----
    float *dev = reinterpret_cast<float *>(smem + lnq_dev_offset(m, K)) + (size_t)wave * K;  // this wave's squared deviations
    for (int r = wave; r < m; r += kWaves) {
        const float *xr = rows + (size_t)r * K;
        float mean = sequential_sum_bcast(xr, K);
        mean /= (float)K;
        for (int k = lane; k < K; k += 64) {
            const float d = xr[k] - mean;
            dev[k] = __fmul_rn(d, d);
        }
        const float sq = sequential_sum_bcast(dev, K);
        if (lane == 0) {
            stats[2 * r] = mean;
            stats[2 * r + 1] = sqrtf(sq / (float)K + 0.00001f);
        }
    }
    __syncthreads();
    // ---- 3. the int8 rows (LayerNormQ.cc:42-48), into LDS and -- by workgroup 0 -- to memory if asked for ----
    auto quantize = [&](int e, float lw, float lb) {
        const int r = e / K;
        const float t = __fdiv_rn(rows[e] - stats[2 * r], stats[2 * r + 1]);
        const float f = __fadd_rn(__fmul_rn(t, lw), lb);
        const int8_t q = (int8_t)(int)roundf(f);
        q8[e] = q;
        if (a.ln_out && blockIdx.x == 0) a.ln_out[e] = q;
    };
    if (affine_prefetched) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int e = tid + 64 * kWaves * i;
            if (e < m * K) quantize(e, pw[i], pb[i]);
        }
    } else {
        for (int e = tid; e < m * K; e += 64 * kWaves) quantize(e, a.ln_w[e % K], a.ln_b[e % K]);
    }
    __syncthreads();
    // ---- 4. this workgroup's output rows: a wave per row, lanes across K in 16-byte pieces ----
#pragma unroll
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
        const int row = row0 + rr;  // wave-uniform
        if (row >= a.total_rows) break;
        int li = 0;
#pragma unroll
        for (int s = 1; s < TCE_MAX_GROUP; ++s)
            if (s < a.count && row >= a.lin[s].row_begin) li = s;
        const LnqLinear &L = a.lin[li];
        const int n = row - L.row_begin;
        const int4_t *brow = reinterpret_cast<const int4_t *>(L.B + (size_t)n * K);
        int acc[kMaxM];
#pragma unroll
        for (int mm = 0; mm < kMaxM; ++mm) acc[mm] = 0;
        for (int p = lane, it = 0; p < pieces; p += 64, ++it) {
            const int4_t w = it == 0 ? wfirst[rr][0] : (it == 1 ? wfirst[rr][1] : brow[p]);
#pragma unroll
            for (int mm = 0; mm < kMaxM; ++mm) {
                if (mm < m) {
                    const int4_t xq = *reinterpret_cast<const int4_t *>(q8 + (size_t)mm * K + p * 16);
                    acc[mm] = __builtin_amdgcn_sdot4(w.x, xq.x, acc[mm], false);
                    acc[mm] = __builtin_amdgcn_sdot4(w.y, xq.y, acc[mm], false);
                    acc[mm] = __builtin_amdgcn_sdot4(w.z, xq.z, acc[mm], false);
                    acc[mm] = __builtin_amdgcn_sdot4(w.w, xq.w, acc[mm], false);
                }
            }
        }
        const float u = ufirst[rr];
#pragma unroll
        for (int mm = 0; mm < kMaxM; ++mm) {
            if (mm < m) {
                int v = acc[mm];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);  // int32: exact in any order
                if (lane == 0) {
                    float f = __fmul_rn((float)v, L.alpha);
                    if (L.out_kind == TCE_OUT_INT8) {
                        if (L.bias_kind == TCE_BIAS_INT8) f = __fadd_rn(f, u);
                        float r = roundf(f);
                        r = fmaxf(r, (float)L.q_min);
                        r = fminf(r, (float)L.q_max);
                        static_cast<int8_t *>(L.C)[(size_t)mm * L.N + n] = (int8_t)(int)r;
                    } else {
                        if (L.bias_kind == TCE_BIAS_FP32) f = __fadd_rn(f, u);
                        static_cast<float *>(L.C)[(size_t)mm * L.N + n] = f;
                    }
                }
            }
        }
    }
}

}  // namespace

int launch_lnq_w8a8_group(const float *x, const float *ln_w, const float *ln_b, int m, int k, const tce_w8a8_desc *lin, int count, void *ln_out,
                          hipStream_t stream, hipError_t *hip_err) {
    LnqArgs a{};
    a.x = x;
    a.ln_w = ln_w;
    a.ln_b = ln_b;
    a.ln_out = static_cast<int8_t *>(ln_out);
    a.m = m;
    a.K = k;
    a.count = count;
    int rows = 0;
    for (int i = 0; i < count; ++i) {
        LnqLinear &L = a.lin[i];
        L.B = static_cast<const int8_t *>(lin[i].B);
        L.bias = lin[i].bias;
        L.C = lin[i].C;
        L.N = lin[i].N;
        L.alpha = lin[i].alpha;
        L.beta = lin[i].beta;
        L.q_min = lin[i].q_min;
        L.q_max = lin[i].q_max;
        L.bias_kind = lin[i].bias_kind;
        L.out_kind = lin[i].out_kind;
        L.row_begin = rows;
        rows += lin[i].N;
    }
    for (int i = count; i < TCE_MAX_GROUP; ++i) a.lin[i] = a.lin[0];
    a.total_rows = rows;
    const size_t lds = lnq_dev_offset(m, k) + (size_t)(m < kWaves ? m : kWaves) * k * 4;  // a scratch row per wave that walks a row
    if (lds > 160 * 1024) return TCE_ERR_UNSUPPORTED_SHAPE;
    auto kfn = lnq_w8a8_kernel;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            if (hip_err) *hip_err = e;
            return TCE_ERR_HIP;
        }
    }
    const int per_wg = kWaves * kRowsPerWave;
    hipLaunchKernelGGL(kfn, dim3((rows + per_wg - 1) / per_wg), dim3(64 * kWaves), lds, stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
