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
// Two kernels.
// lnq_w8a8_kernel<2, 2, 4>: the OPT-125M form above -- a workgroup of 4 waves per 8 output rows, the first two 16-byte pieces per lane of a wave's two rows requested
// before the sums; ONE normalised row (the decode token) has its two sums walked by the four waves together (sequential_sum_speculated, below).
// lnq_w8a8_wide_kernel (k >= 1024: OPT-1.3B / 6.7B, where a row's two sequential sums are 2 x 4096 dependent additions -- 22 us for one wave -- and a launch's
// int8 weights are 50-67 MB): ONE workgroup of 16 waves per CU, each normalises once and owns a contiguous run of output rows per wave.
//   * the two sums are walked by all 16 waves at once (sequential_sum_speculated, tce_common.hpp: every wave adds its sixteenth of the row from 64 candidate
//     running values, wave 0 then picks the lanes that started from the true ones -- the additions that count are the reference's, in its order);
//   * a wave's first four rows are held in registers (16 x 16 bytes per lane: 67 MB across the chip -- all of OPT-6.7B's fc1), requested in FOUR batches at four
//     points of the sums (a CU accepts ~64 KB of requests; with everything requested up front the waves that walk the sums stood in the queue), so the weights
//     stream from memory WHILE the sums are walked; what is left of a longer run is streamed afterwards, eight 16-byte pieces per lane in flight;
//   * its barriers leave global loads in flight (lds_barrier), its bias words are requested by one unconditional load and converted late, and the build fails if
//     it spills a vector register (build.py): each of the three had put the first sum behind the last weight (debug mode 85 + scripts/lnq_wide_phases.py).
// Per launch at OPT-6.7B (q/k/v 3 x 4096 x 4096, fc1 16384 x 4096): 87 / 94 us for the 8-rows-per-workgroup form (1536-2048 workgroups: the sums are walked in
// several rounds), 49 / 53 us with one round of 4-wave workgroups (three per CU: their three chains share a SIMD), 23-24 / 24-25 us as it stands.
//
// BIT-EXACT against LayerNormQ::forward followed by int8_ref_matmul (kernels/ref/matmul_ref_int8.cc:11-35):
//   * the reference's two row sums are sequential fp32 additions, so they are added in order (a wave per row for m > 1 in the 4-wave form: every lane holds
//     the same accumulator and gets the values by LDS broadcast reads; else speculated across the waves, which changes when an addition is done, not which); the division, multiply and add of the output are separate roundings (-ffp-contract=off), std::round = half away
//     from zero -- the same code as tce_layernorm_q (glue.hip);
//   * the dot products are int32 (v_dot4_i32_i8), exact in any order;
//   * the epilogue is the int8 path's: (float)acc, * alpha, + (float)bias * beta, each rounded separately, roundf, clamp.
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

static thread_local int g_lnq_form = 0;  // debug: 1 = the workgroup-per-8-rows form at every k (A/B of the weights-resident form)
void set_lnq_form(int f) { g_lnq_form = f; }
static unsigned long long *g_lnq_stamps = nullptr;
void set_lnq_stamps(void *p) { g_lnq_stamps = static_cast<unsigned long long *>(p); }

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
    int rpw;  // output rows per wave (a contiguous run)
    int dbg;  // timing experiments (debug modes 83, 84): 2 no sums (mean 0, std 1: WRONG results), 3 no output rows
    unsigned long long *stamps;  // debug mode 85: workgroup 0's phase times (100 MHz clock) into the debug buffer
    LnqLinear lin[TCE_MAX_GROUP];
};

constexpr int kWaves = 4;
constexpr int kMaxM = 8;

// LDS: [m][K] fp32 rows | [m][K] int8 | (pad) [m][2] stats | (pad) [min(m, kWaves)][K] fp32 squared deviations
__host__ __device__ constexpr size_t lnq_dev_offset(int m, int K) { return (((size_t)m * K * 5 + 16 + (size_t)m * 8 + 16) + 15) & ~(size_t)15; }

// The bias of output column n is requested early and converted late (kernels/ref/matmul_ref_int8.cc:29-31 forms the additive term from it): ONE unconditional
// load of the aligned 32-bit word that holds it (an int8 value -> the word around it; no bias -> a word of the weights, ignored).  With a branch per bias kind the
// compiler put a wait for ALL loads issued so far between two rows' weight requests (measured: the four rows of a wave arrived one after the other, 10 us).
__device__ __forceinline__ int bias_raw(const LnqLinear &L, int n) {
    const char *base = L.bias_kind == TCE_BIAS_NONE ? reinterpret_cast<const char *>(L.B) : static_cast<const char *>(L.bias);
    const size_t off = L.bias_kind == TCE_BIAS_FP32 ? (size_t)n * 4 : (L.bias_kind == TCE_BIAS_INT8 ? (size_t)n : 0);
    const char *p = base + off;
    p -= reinterpret_cast<uintptr_t>(p) & 3;  // (pointer arithmetic, not an integer round trip: the load stays a global one)
    return *reinterpret_cast<const int *>(p);
}
__device__ __forceinline__ float bias_term(const LnqLinear &L, int n, int raw) {
    if (L.bias_kind == TCE_BIAS_INT8) {
        const int sh = (int)((reinterpret_cast<uintptr_t>(static_cast<const char *>(L.bias) + n) & 3) * 8);
        return __fmul_rn((float)(int)(int8_t)(raw >> sh), L.beta);
    }
    return L.bias_kind == TCE_BIAS_FP32 ? __builtin_bit_cast(float, raw) : 0.f;
}

// RW rows of a wave's run are requested ahead, PL 16-byte pieces per lane of each; PF affine parameter pairs per thread are requested ahead (else read in step 3)
template <int RW, int PL, int PF>
__global__ __launch_bounds__(64 * kWaves) void lnq_w8a8_kernel(const LnqArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, m = a.m;
    float *rows = reinterpret_cast<float *>(smem);                     // [m][K] fp32
    int8_t *q8 = reinterpret_cast<int8_t *>(smem + (size_t)m * K * 4);  // [m][K] int8
    float *stats = reinterpret_cast<float *>(smem + (size_t)m * K * 5 + 16 - ((size_t)m * K * 5) % 16);  // [m][2]: mean, std
    const int row0 = (blockIdx.x * kWaves + wave) * a.rpw;
    const int pieces = K >> 4;
    // the activation rows are requested FIRST (loads return in order: the sums must not wait behind the weights) ...
    constexpr int XB = 4;
    const int n4 = (m * K) >> 2;
    float4_t xfirst[XB];
#pragma unroll
    for (int i = 0; i < XB; ++i) xfirst[i] = tid + 64 * kWaves * i < n4 ? reinterpret_cast<const float4_t *>(a.x)[tid + 64 * kWaves * i] : float4_t{0.f, 0.f, 0.f, 0.f};
    // ... then the first weight pieces of the wave's first RW rows: they do not depend on the normalisation and land while the row sums are walked
    // (K = 768: one piece per lane and row is the whole row; <4, 4>: 4 rows of up to 4096 bytes completely)
    int4_t wfirst[RW][PL];
    int ufirst[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int rw = rr < a.rpw ? row0 + rr : row0;
        const int row = rw < a.total_rows ? rw : a.total_rows - 1;
        int li = 0;
#pragma unroll
        for (int s = 1; s < TCE_MAX_GROUP; ++s)
            if (s < a.count && row >= a.lin[s].row_begin) li = s;
        const int4_t *brow = reinterpret_cast<const int4_t *>(a.lin[li].B + (size_t)(row - a.lin[li].row_begin) * K);
#pragma unroll
        for (int i = 0; i < PL; ++i) wfirst[rr][i] = lane + 64 * i < pieces ? brow[lane + 64 * i] : int4_t{0, 0, 0, 0};
        // the bias of this output column is requested too -- RAW: converting it here would put a wait for ALL loads issued so far between two rows' requests
        const LnqLinear &L = a.lin[li];
        ufirst[rr] = bias_raw(L, row - L.row_begin);
    }
    // ... and the affine parameters of the elements this thread will quantize in step 3 (m * K <= 256 PF)
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
#pragma unroll
    for (int i = 0; i < XB; ++i)
        if (tid + 64 * kWaves * i < n4) reinterpret_cast<float4_t *>(rows)[tid + 64 * kWaves * i] = xfirst[i];
    for (int p = tid + 64 * kWaves * XB; p < n4; p += 64 * kWaves) reinterpret_cast<float4_t *>(rows)[p] = reinterpret_cast<const float4_t *>(a.x)[p];
    lds_barrier();
    // ---- 2. the two sequential sums of every row (LayerNormQ.cc:27-37), in the reference's order: wave w takes rows w, w + 4
    //         (sequential_sum_bcast, tce_common.hpp: one dependent add per element) ----
    float *dev = reinterpret_cast<float *>(smem + lnq_dev_offset(m, K)) + (size_t)wave * K;  // this wave's squared deviations
    if (m == 1 && a.dbg != 4) {
        // ONE row (the decode token): its two sums are walked by the four waves together (sequential_sum_speculated, tce_common.hpp: a quarter of the row per
        // wave from 64 candidate running values, bit-identical); a wave forms the squared deviations of its own quarter
        float *dev0 = reinterpret_cast<float *>(smem + lnq_dev_offset(m, K));
        float *sp = dev0 + K;
        float mean = sequential_sum_speculated<kWaves, false>(rows, K, sp, wave, lane);
        mean /= (float)K;
        int b, len;
        speculated_segment<kWaves>(K, wave, b, len);
        double ds = 0.0;
        for (int k = b + lane; k < b + len; k += 64) {
            const float d = rows[k] - mean;
            const float d2 = __fmul_rn(d, d);
            dev0[k] = d2;
            ds += (double)d2;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ds += __shfl_xor(ds, off, 64);
        const float sq = sequential_sum_speculated<kWaves, false>(dev0, K, sp, wave, lane, ds);
        if (tid == 0) {
            stats[0] = mean;
            stats[1] = sqrtf(sq / (float)K + 0.00001f);
        }
    } else
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
    lds_barrier();
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
    lds_barrier();
    // ---- 4. this wave's output rows, lanes across K in 16-byte pieces: the first RW rows start from the registers requested at the top ----
    auto finish_row = [&](const LnqLinear &L, int n, const int (&acc)[kMaxM], float u) {
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
    };
    auto linear_of = [&](int row) {
        int li = 0;
#pragma unroll
        for (int s = 1; s < TCE_MAX_GROUP; ++s)
            if (s < a.count && row >= a.lin[s].row_begin) li = s;
        return li;
    };
    auto dot_piece = [&](const int4_t &w, int p, int (&acc)[kMaxM]) {
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
    };
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int row = row0 + rr;  // wave-uniform
        if (rr >= a.rpw || row >= a.total_rows) break;
        const LnqLinear &L = a.lin[linear_of(row)];
        const int n = row - L.row_begin;
        const int4_t *brow = reinterpret_cast<const int4_t *>(L.B + (size_t)n * K);
        int acc[kMaxM];
#pragma unroll
        for (int mm = 0; mm < kMaxM; ++mm) acc[mm] = 0;
#pragma unroll
        for (int i = 0; i < PL; ++i)
            if (lane + 64 * i < pieces) dot_piece(wfirst[rr][i], lane + 64 * i, acc);
        for (int p = lane + 64 * PL; p < pieces; p += 64) dot_piece(brow[p], p, acc);
        finish_row(L, n, acc, bias_term(L, n, ufirst[rr]));
    }
    // rows of the run beyond the RW requested ahead (a launch held to the chip's capacity): streamed now, four pieces per lane in flight
    for (int rr = RW; rr < a.rpw; ++rr) {
        const int row = row0 + rr;
        if (row >= a.total_rows) break;
        const LnqLinear &L = a.lin[linear_of(row)];
        const int n = row - L.row_begin;
        const int4_t *brow = reinterpret_cast<const int4_t *>(L.B + (size_t)n * K);
        const int uraw = bias_raw(L, n);
        int acc[kMaxM];
#pragma unroll
        for (int mm = 0; mm < kMaxM; ++mm) acc[mm] = 0;
        for (int p0 = lane; p0 < pieces; p0 += 256) {
            int4_t w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = p0 + 64 * i < pieces ? brow[p0 + 64 * i] : int4_t{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (p0 + 64 * i < pieces) dot_piece(w[i], p0 + 64 * i, acc);
        }
        finish_row(L, n, acc, bias_term(L, n, uraw));
    }
}


// ---- the wide form: see the header comment ----
constexpr int kWideWaves = 16;
constexpr int kWideRW = 4;  // rows of a wave's run requested ahead ...
constexpr int kWidePL = 4;  // ... completely up to k = 4096 (four 16-byte pieces per lane)

// LDS: [m][K] fp32 rows | [m][K] int8 | (pad) [m][2] stats | (pad) [K] fp32 squared deviations | the speculated sum's scratch
__host__ __device__ constexpr size_t lnq_wide_lds(int m, int K) { return lnq_dev_offset(m, K) + (size_t)K * 4 + (size_t)kSpecScratchFloats(kWideWaves) * 4; }

__global__ __launch_bounds__(64 * kWideWaves) void lnq_w8a8_wide_kernel(const LnqArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = 64 * kWideWaves, RW = kWideRW, PL = kWidePL, PF = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, m = a.m;
    float *rows = reinterpret_cast<float *>(smem);
    int8_t *q8 = reinterpret_cast<int8_t *>(smem + (size_t)m * K * 4);
    float *stats = reinterpret_cast<float *>(smem + (size_t)m * K * 5 + 16 - ((size_t)m * K * 5) % 16);
    float *dev = reinterpret_cast<float *>(smem + lnq_dev_offset(m, K));
    float *sp = dev + K;
    const int row0 = (blockIdx.x * kWideWaves + wave) * a.rpw;
    const int pieces = K >> 4;
    unsigned long long *stamps = (a.stamps && blockIdx.x == 0) ? a.stamps : nullptr;
    if (stamps && tid == 0) stamps[0] = wall_clock64();
    // ---- 1. the activation rows into LDS, BEFORE anything else is requested: loads return in order, and with the weight requests in front of them (or merely
    //         behind them, but with a wait the compiler cannot count across their conditions) the sums would start when the last weight has arrived -- measured:
    //         10 us into the launch.  This way the weights' whole flight lies under the sums ----
    const int n4 = (m * K) >> 2;
    for (int p = tid; p < n4; p += NT) reinterpret_cast<float4_t *>(rows)[p] = reinterpret_cast<const float4_t *>(a.x)[p];
    asm volatile("" ::: "memory");
    // then the affine parameters and the first rows of the wave's run
    const bool affine_prefetched = m * K <= NT * PF;
    float pw[PF], pb[PF];
    if (affine_prefetched) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int e = tid + NT * i;
            const int k = e < m * K ? e % K : 0;
            pw[i] = a.ln_w[k];
            pb[i] = a.ln_b[k];
        }
    }
    auto linear_of = [&](int row) {
        int li = 0;
#pragma unroll
        for (int s = 1; s < TCE_MAX_GROUP; ++s)
            if (s < a.count && row >= a.lin[s].row_begin) li = s;
        return li;
    };
    // The wave's first four rows are requested in FOUR batches at four points of the sums, not all at once: a CU accepts only so many requests (~64 KB in
    // flight), a wave whose request is not accepted yet stands still, and with 256 KB per CU requested up front the first sum started when most of it had
    // arrived (measured: 8-10 us into the launch).  A batch is 64 KB per CU, 17 MB across the chip: about what streams in during one phase of the sums.
    int4_t wfirst[RW][PL];
    int ufirst[RW];  // (the bias RAW: see bias_raw)
    auto request_row = [&](int rr) {  // rr: compile-time after unrolling
        const int rw = rr < a.rpw ? row0 + rr : row0;
        const int row = rw < a.total_rows ? rw : a.total_rows - 1;
        const LnqLinear &L = a.lin[linear_of(row)];
        const int4_t *brow = reinterpret_cast<const int4_t *>(L.B + (size_t)(row - L.row_begin) * K);
#pragma unroll
        for (int i = 0; i < PL; ++i) wfirst[rr][i] = brow[lane + 64 * i < pieces ? lane + 64 * i : pieces - 1];  // (unconditional: pieces past the row are not used)
        ufirst[rr] = bias_raw(L, row - L.row_begin);
    };
    request_row(0);
    if (a.dbg == 2) {
        request_row(1);
        request_row(2);
        request_row(3);
    }
    lds_barrier();
    if (stamps && tid == 0) stamps[1] = wall_clock64();
    // ---- 2. the two sequential sums of every row (LayerNormQ.cc:27-37), all waves on one row at a time; a wave forms the squared deviations of ITS segment (and
    //         their double-precision sum) itself, so nothing but the speculated sum's own barriers stands between the two sums ----
    for (int r = 0; r < m; ++r) {
        const float *xr = rows + (size_t)r * K;
        float mean, sq;
        if (a.dbg == 2) {
            mean = 0.f;
            sq = (float)K;
        } else {
            mean = sequential_sum_speculated<kWideWaves>(xr, K, sp, wave, lane, stamps ? stamps + 8 : nullptr, [&]() { if (r == 0) request_row(1); });
            if (stamps && tid == 0) stamps[2] = wall_clock64();
            mean /= (float)K;
            int b, len;
            speculated_segment<kWideWaves>(K, wave, b, len);
            double ds = 0.0;
            for (int k = b + lane; k < b + len; k += 64) {
                const float d = xr[k] - mean;
                const float d2 = __fmul_rn(d, d);
                dev[k] = d2;
                ds += (double)d2;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) ds += __shfl_xor(ds, off, 64);
            if (r == 0) request_row(2);
            sq = sequential_sum_speculated<kWideWaves>(dev, K, sp, wave, lane, ds, stamps ? stamps + 12 : nullptr, [&]() { if (r == 0) request_row(3); });
            if (stamps && tid == 0) stamps[3] = wall_clock64();
        }
        if (tid == 0) {
            stats[2 * r] = mean;
            stats[2 * r + 1] = sqrtf(sq / (float)K + 0.00001f);
        }
    }
    lds_barrier();
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
            const int e = tid + NT * i;
            if (e < m * K) quantize(e, pw[i], pb[i]);
        }
    } else {
        for (int e = tid; e < m * K; e += NT) quantize(e, a.ln_w[e % K], a.ln_b[e % K]);
    }
    lds_barrier();
    if (stamps && tid == 0) stamps[4] = wall_clock64();
    // ---- 4. this wave's run of output rows, lanes across K in 16-byte pieces ----
    auto finish_row = [&](const LnqLinear &L, int n, const int (&acc)[kMaxM], float u) {
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
    };
    auto dot_piece = [&](const int4_t &w, int p, int (&acc)[kMaxM]) {
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
    };
    if (a.dbg == 3) return;
    const int last = row0 + a.rpw < a.total_rows ? row0 + a.rpw : a.total_rows;  // one past the wave's last row
    auto request = [&](int row, int p0, int4_t (&w)[PL]) {
        const int rc = row < last ? row : last - 1;
        const LnqLinear &L = a.lin[linear_of(rc)];
        const int4_t *brow = reinterpret_cast<const int4_t *>(L.B + (size_t)(rc - L.row_begin) * K);
#pragma unroll
        for (int i = 0; i < PL; ++i) w[i] = (row < last && p0 + 64 * i < pieces) ? brow[p0 + 64 * i] : int4_t{0, 0, 0, 0};
    };
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int row = row0 + rr;  // wave-uniform
        if (row >= last) break;
        const LnqLinear &L = a.lin[linear_of(row)];
        const int n = row - L.row_begin;
        const int4_t *brow = reinterpret_cast<const int4_t *>(L.B + (size_t)n * K);
        int acc[kMaxM];
#pragma unroll
        for (int mm = 0; mm < kMaxM; ++mm) acc[mm] = 0;
#pragma unroll
        for (int i = 0; i < PL; ++i)
            if (lane + 64 * i < pieces) dot_piece(wfirst[rr][i], lane + 64 * i, acc);
        for (int p = lane + 64 * PL; p < pieces; p += 64) dot_piece(brow[p], p, acc);
        finish_row(L, n, acc, bias_term(L, n, ufirst[rr]));
    }
    if (stamps && tid == 0) stamps[5] = wall_clock64();
    // what is left of the run (a launch with more than 4 rows per wave: over 16384 rows) is streamed now, two rows x four pieces per lane in flight
    int4_t wa[PL], wb[PL];
    if (row0 + RW < last) {
        request(row0 + RW, lane, wa);
        request(row0 + RW + 1, lane, wb);
    }
    for (int row = row0 + RW; row < last; row += 2) {
        int acca[kMaxM], accb[kMaxM];
#pragma unroll
        for (int mm = 0; mm < kMaxM; ++mm) acca[mm] = accb[mm] = 0;
        for (int p0 = lane; p0 < pieces; p0 += 64 * PL) {
            int4_t ca[PL], cb[PL];
#pragma unroll
            for (int i = 0; i < PL; ++i) {
                ca[i] = wa[i];
                cb[i] = wb[i];
            }
            // the next batch: the rest of these two rows, else the first pieces of the next two
            if (p0 + 64 * PL < pieces) {
                request(row, p0 + 64 * PL, wa);
                request(row + 1, p0 + 64 * PL, wb);
            } else if (row + 2 < last) {
                request(row + 2, lane, wa);
                request(row + 3, lane, wb);
            }
#pragma unroll
            for (int i = 0; i < PL; ++i) {
                if (p0 + 64 * i < pieces) {
                    dot_piece(ca[i], p0 + 64 * i, acca);
                    if (row + 1 < last) dot_piece(cb[i], p0 + 64 * i, accb);
                }
            }
        }
        {
            const LnqLinear &L = a.lin[linear_of(row)];
            finish_row(L, row - L.row_begin, acca, bias_term(L, row - L.row_begin, bias_raw(L, row - L.row_begin)));
        }
        if (row + 1 < last) {
            const LnqLinear &L = a.lin[linear_of(row + 1)];
            finish_row(L, row + 1 - L.row_begin, accb, bias_term(L, row + 1 - L.row_begin, bias_raw(L, row + 1 - L.row_begin)));
        }
    }
}

int launch_wide(LnqArgs &a, int rows, hipStream_t stream, hipError_t *hip_err) {
    const size_t lds = lnq_wide_lds(a.m, a.K);
    if (lds > 160 * 1024) return TCE_ERR_UNSUPPORTED_SHAPE;
    // CU count of the CURRENT device, cached per device id (one host thread may drive several devices: comm.hip's DeviceGuard mode)
    static int cus_of[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus_of[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus_of[dev] = prop.multiProcessorCount;
        if (cus_of[dev] <= 0) cus_of[dev] = 256;
    }
    const int cus = cus_of[dev];
    // one workgroup per CU (a second one would only share the CU's adders and LDS with the first); fewer when there are fewer than 2 rows per wave
    int grid = (rows + 2 * kWideWaves - 1) / (2 * kWideWaves);
    if (grid > cus) grid = cus;
    a.rpw = (rows + grid * kWideWaves - 1) / (grid * kWideWaves);
    grid = (rows + a.rpw * kWideWaves - 1) / (a.rpw * kWideWaves);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lnq_w8a8_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            if (hip_err) *hip_err = e;
            return TCE_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(lnq_w8a8_wide_kernel, dim3(grid), dim3(64 * kWideWaves), lds, stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
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
    const bool wide = (k >= 1024 || (g_lnq_form == 6 && k >= 256)) && g_lnq_form != 1;  // (debug mode 86: the wide form from k = 256 on)
    a.dbg = g_lnq_form >= 2 && g_lnq_form <= 4 ? g_lnq_form - 1 : (g_lnq_form == 7 ? 4 : 0);  // (debug mode 87: the 4-wave form walks one row's sums with one wave, as before)
    a.stamps = g_lnq_form == 5 ? g_lnq_stamps : nullptr;
    if (wide) return launch_wide(a, rows, stream, hip_err);
    const size_t lds = lnq_dev_offset(m, k) + (size_t)(m < kWaves ? m : kWaves) * k * 4 + (m == 1 ? (size_t)kSpecScratchFloats(kWaves) * 4 : 0);  // a scratch row per wave that walks a row (+ one row's speculated sums)
    if (lds > 160 * 1024) return TCE_ERR_UNSUPPORTED_SHAPE;
    auto kfn = lnq_w8a8_kernel<2, 2, 4>;
    const int grid = (rows + 2 * kWaves - 1) / (2 * kWaves);
    a.rpw = 2;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            if (hip_err) *hip_err = e;
            return TCE_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * kWaves), lds, stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
