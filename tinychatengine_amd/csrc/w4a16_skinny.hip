// w4a16_skinny.hip -- W4A16 for small batches, 3 <= M <= 16 (batched / speculative decoding) and, in 16-row slices of the
// batch on gridDim.y, up to M = 128 while N is small (short prompts), gfx950.
//
// Between the decode GEMV (M = 1: dot products on the 4x4x4 MFMA diagonal, one activation vector in LDS) and the prefill
// GEMM (M >= 17: 64-row activation tiles) the weights must still be streamed ONCE at HBM speed, but every weight now
// meets up to 16 activations.  The GEMV kernels repeat their MFMAs per activation row and are MFMA-issue-bound from
// M = 3 (22016x4096: 12.5 us at M = 1, 40 us at M = 8); the GEMM's k-loop is a 0.7 us serial chain per 128 k.  Measured
// here (profiles/r1/gemv_small_batch_sweep.jsonl): 4096^2 6.3-7.0 us for M = 3..16 (was 7.7-27.8), 22016x4096 19-24 us
// (was 24-40), 4096x11008 11.2-13.4 us (was 18.5-67).  Design:
//
//   * one 16-row weight tile per workgroup, its K range cut between the workgroup's KS waves (KS = 1..8, picked so that
//     a launch has a few thousand waves; fp32 partial tiles are summed through LDS in wave order: deterministic);
//   * v_mfma_f32_16x16x32_f16 with A = the weight tile (16 rows x 32 k), B = the activations (32 k x 16 batch rows): all
//     M <= 16 batch rows ride on one instruction;
//   * every wave is autonomous (no workgroup barrier in the k-loop): per 128-wide k-block it loads 1 KiB of packed
//     weights (4 consecutive lanes read one row's 64 bytes -- the operand fragment itself would make each quarter-wave
//     touch 16 rows, see w8a8_gemm.hip) and the 16 x 128 activation block (4 KiB, L1/L2 hits), passes both through its
//     own LDS slots (swizzled, conflict-free both ways) into fragment order, unpacks with the GEMV's 7-instruction bias
//     form (halves 1024 + 16 q) and removes bias and zero point algebraically:
//         sum_k (q - z) x = ( sum_k (1024 + 16 q) x - (1024 + 16 z) sum_k x ) / 16,
//     where sum_k x per (batch row, group) comes from the same MFMA with A = ones;
//   * scales and (1024 + 16 z) of the tile's rows and the wave's groups are staged once into LDS as halves (exact).
//
// Same math and tolerance as the GEMV (fp32 accumulation of exact products, one fp32 scale per group); G = 128 only.
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

namespace {

struct SkinnyArgs {
    const half_t *A;  // fp16 [M][lda]
    const uint4_t *qweight;
    const half_t *scales;
    const unsigned *zeros;
    half_t *C;
    int M, N, K, lda, ldc, scales_stride, zeros_stride;
    int epilogue;  // TCE_W4_SILU_MUL_PAIRS / TCE_W4_ADD_TO_C
    int groups_per_wave;
};

constexpr int kWBuf = 64 * 16;       // packed weight block of one wave: 16 rows x 64 bytes
constexpr int kXBuf = 256 * 16;      // activation block of one wave: 16 rows x 256 bytes

template <bool Z8>
__global__ __launch_bounds__(512) void w4a16_skinny_kernel(const SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = blockDim.x >> 6;
    const int n0 = blockIdx.x * 16;
    // batch rows 16 * blockIdx.y .. + 15 (M > 16: the weight tile is streamed once per 16 rows, the repeats out of L2 / MALL)
    const int m0 = blockIdx.y * 16;
    const half_t *Am = a.A + (size_t)m0 * a.lda;
    half_t *Cm = a.C + (size_t)m0 * a.ldc;
    const int Mloc = a.M - m0 < 16 ? a.M - m0 : 16;

    const int nkb = a.K >> 7;
    const int gpw = a.groups_per_wave;                     // k-blocks per wave (the last wave may have fewer)
    const int kb0 = wave * gpw;
    const int my = kb0 < nkb ? (nkb - kb0 < gpw ? nkb - kb0 : gpw) : 0;
    const int nchunks = a.K >> 5;

    // per-wave LDS: weight block, activation block, scale / (1024 + 16 z) tables [gpw][16] halves
    unsigned char *wv = smem + (size_t)wave * (kWBuf + kXBuf + (size_t)gpw * 64);
    uint4_t *lds_w = reinterpret_cast<uint4_t *>(wv);
    uint4_t *lds_x = reinterpret_cast<uint4_t *>(wv + kWBuf);
    half_t *tab_s = reinterpret_cast<half_t *>(wv + kWBuf + kXBuf);
    half_t *tab_c = tab_s + gpw * 16;

    // ---- tables of this wave's groups (strided 2-byte reads, once) ----
    for (int idx = lane; idx < my * 16; idx += 64) {
        const int g = idx >> 4, r = idx & 15;
        int n = n0 + r;
        n = n < a.N ? n : a.N - 1;
        const int kb = kb0 + g;
        tab_s[idx] = a.scales[(size_t)n * a.scales_stride + kb];
        unsigned z = 8u;
        if constexpr (!Z8) z = (a.zeros[(size_t)n * a.zeros_stride + (kb >> 3)] >> ((kb & 7) * 4)) & 0xFu;
        tab_c[idx] = (half_t)(float)(1024u + 16u * z);  // exact in binary16
    }

    // ---- lane roles ----
    const int r16 = lane & 15, kq = lane >> 4;        // MFMA view: A row / B column r16, k-slice kq
    const int lrow = lane >> 2, lchunk = lane & 3;    // weight load view: row lrow, 16-byte chunk lchunk of the k-block
    int wrow = n0 + lrow;
    wrow = wrow < a.N ? wrow : a.N - 1;
    const uint4_t *wsrc = a.qweight + (size_t)wrow * nchunks + lchunk;
    const int w_wslot = lrow * 4 + (lchunk ^ ((lrow >> 2) & 3));
    const int w_rslot = r16 * 4 + (kq ^ ((r16 >> 2) & 3));
    // activation block: piece p = j*64 + lane -> batch row p/16, 16-byte piece p%16 of the row's 256 bytes
    const half_t *xsrc[4];
    int x_wslot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xm = 4 * j + (lane >> 4), pc = lane & 15;
        const int m = xm < Mloc ? xm : Mloc - 1;  // rows past M repeat the last row; their outputs are not stored
        xsrc[j] = Am + (size_t)m * a.lda + pc * 8;
        x_wslot[j] = pc * 16 + (xm ^ pc);
    }

    struct Regs {
        uint4_t w;
        uint4_t x[4];
    };
    auto load = [&](Regs &r, int g) {  // g: k-block within this wave's range (clamped by the caller)
        const int kb = kb0 + g;
        r.w = wsrc[kb * 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r.x[j] = *reinterpret_cast<const uint4_t *>(xsrc[j] + kb * 128);
    };

    float4_t acc = float4_t{0.f, 0.f, 0.f, 0.f};  // D[n = 4*kq + i][m = r16], i = 0..3
    unsigned mask_hi;
    asm volatile("v_mov_b32 %0, 0x00F000F0" : "=v"(mask_hi));
    const unsigned magic = 0x64006400u;
    const half8_t ones = half8_t{(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};

    auto process = [&](const Regs &r, int g) {
        // registers -> LDS in fragment-friendly order.  A wave's LDS operations complete in order, so the reads below see
        // these writes without a barrier; the fence only keeps the compiler from reordering them.
        lds_w[w_wslot] = r.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_x[x_wslot[j]] = pair_permute(r.x[j]);
        __builtin_amdgcn_wave_barrier();
        const uint4_t wq = lds_w[w_rslot];  // row r16, k-chunk kq: word s feeds MFMA step s
        float4_t blk = float4_t{0.f, 0.f, 0.f, 0.f}, xs = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pc = kq * 4 + s;
            const half8_t xb = __builtin_bit_cast(half8_t, lds_x[pc * 16 + (r16 ^ pc)]);  // batch row r16, its piece pc
            const unsigned w = wq[s];
            const unsigned t0 = ((w << 4) & mask_hi) | magic;  // (1024+16 q0, 1024+16 q4)
            const unsigned t1 = (w & mask_hi) | magic;         // (q1, q5)
            const unsigned t2 = ((w >> 4) & mask_hi) | magic;  // (q2, q6)
            const unsigned t3 = ((w >> 8) & mask_hi) | magic;  // (q3, q7)
            const half8_t wa = __builtin_bit_cast(half8_t, uint4_t{t0, t1, t2, t3});
            blk = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb, blk, 0, 0, 0);
            xs = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, xb, xs, 0, 0, 0);  // every row: sum_k x[m = r16][k]
        }
        __builtin_amdgcn_wave_barrier();
        // rows 4*kq .. 4*kq+3 of the tile: their scale and (1024 + 16 z) for this group
        const half4_t s4 = *reinterpret_cast<const half4_t *>(tab_s + g * 16 + 4 * kq);
        const half4_t c4 = *reinterpret_cast<const half4_t *>(tab_c + g * 16 + 4 * kq);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = __builtin_fmaf(-(float)c4[i], xs[i], blk[i]);
            acc[i] = __builtin_fmaf((float)s4[i], t, acc[i]);
        }
    };

    // ---- the wave's k-blocks, loads one block ahead (two static register sets; clamped, never predicated) ----
    if (my > 0) {
        Regs r0, r1;
        load(r0, 0);
        for (int g = 0; g < my; g += 2) {
            load(r1, g + 1 < my ? g + 1 : my - 1);
            __builtin_amdgcn_sched_barrier(0);
            process(r0, g);
            if (g + 1 < my) {
                load(r0, g + 2 < my ? g + 2 : my - 1);
                __builtin_amdgcn_sched_barrier(0);
                process(r1, g + 1);
            }
        }
    }

    // ---- sum the K slices (wave order), scale by 1/16, epilogue, store ----
    if (KS > 1) {
        __syncthreads();  // every wave is done with its LDS slots: the front of LDS becomes the reduction buffer
        float4_t *red = reinterpret_cast<float4_t *>(smem);
        red[wave * 64 + lane] = acc;
        __syncthreads();
        if (wave != 0) return;
        acc = red[lane];
        for (int w = 1; w < KS; ++w) {
            const float4_t o = red[w * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += o[i];
        }
    }
    const int m = r16;
    if (m >= Mloc) return;
    const int nb = n0 + 4 * kq;
    half_t outv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) outv[i] = (half_t)(acc[i] * 0.0625f);
    half_t *crow = Cm + (size_t)m * a.ldc;
    if (a.epilogue & TCE_W4_SILU_MUL_PAIRS) {  // rows (2n, 2n+1) = (gate n, up n); nb is a multiple of 4
        if (nb + 1 < a.N) crow[nb >> 1] = silu_mul_half(outv[0], outv[1]);
        if (nb + 3 < a.N) crow[(nb >> 1) + 1] = silu_mul_half(outv[2], outv[3]);
    } else if (a.epilogue & TCE_W4_ADD_TO_C) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (nb + i < a.N) crow[nb + i] = crow[nb + i] + outv[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (nb + i < a.N) crow[nb + i] = outv[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Large N: every wave of the kernel above fetches its own 4 KiB activation block per KiB of weights through L1, which caps
// it at 2-2.4 TB/s once there are enough tiles to fill the chip without a K split (22016 x 4096: 19.5-24 us).  Here the 8
// waves of a workgroup take 8 neighbouring weight tiles over the WHOLE K range and share the activation blocks: the
// workgroup stages two k-blocks (16 rows x 256 k = 8 KiB, one 16-byte piece per thread, permuted once) into a
// double-buffered LDS image per barrier; weights, tables and arithmetic are as above.  K % 256 == 0.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSharedWaves = 8;

template <bool Z8>
__global__ __launch_bounds__(512) void w4a16_skinny_shared_kernel(const SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkb = a.K >> 7;
    const int nst = nkb >> 1;  // stages of two k-blocks
    // batch rows 16 * blockIdx.y .. + 15 (M > 16: the weight tile is streamed once per 16 rows, the repeats out of L2 / MALL)
    const int m0 = blockIdx.y * 16;
    const half_t *Am = a.A + (size_t)m0 * a.lda;
    half_t *Cm = a.C + (size_t)m0 * a.ldc;
    const int Mloc = a.M - m0 < 16 ? a.M - m0 : 16;
    const int tiles = (a.N + 15) >> 4;
    int tile = blockIdx.x * kSharedWaves + wave;
    const bool idle = tile >= tiles;  // a wave past the last tile walks the barriers on a clamped tile and stores nothing
    tile = idle ? tiles - 1 : tile;
    const int n0 = tile * 16;
    const int nchunks = a.K >> 5;

    // LDS: [2][2 x 256] activation pieces, then per wave: two weight blocks (2 KiB) and tables [nkb_pad][16] x 2 halves.
    // The tables are padded to a multiple of 8 k-blocks with scale 0: the stage loop below runs in groups of four stages on
    // clamped addresses and the surplus stages of the last group add exactly 0.
    const int nkb_pad = (nkb + 7) & ~7;
    uint4_t *lds_x = reinterpret_cast<uint4_t *>(smem);
    unsigned char *wv = smem + 2 * 2 * kXBuf + (size_t)wave * (2 * kWBuf + (size_t)nkb_pad * 64);
    uint4_t *lds_w = reinterpret_cast<uint4_t *>(wv);
    half_t *tab_s = reinterpret_cast<half_t *>(wv + 2 * kWBuf);
    half_t *tab_c = tab_s + nkb_pad * 16;
    for (int idx = lane; idx < nkb_pad * 16; idx += 64) {
        const int g = idx >> 4, r = idx & 15;
        int n = n0 + r;
        n = n < a.N ? n : a.N - 1;
        const int gc = g < nkb ? g : nkb - 1;
        const half_t sv = a.scales[(size_t)n * a.scales_stride + gc];
        tab_s[idx] = g < nkb ? sv : (half_t)0.f;
        unsigned z = 8u;
        if constexpr (!Z8) z = (a.zeros[(size_t)n * a.zeros_stride + (gc >> 3)] >> ((gc & 7) * 4)) & 0xFu;
        tab_c[idx] = (half_t)(float)(1024u + 16u * z);
    }

    const int r16 = lane & 15, kq = lane >> 4;
    const int lrow = lane >> 2, lchunk = lane & 3;
    int wrow = n0 + lrow;
    wrow = wrow < a.N ? wrow : a.N - 1;
    const uint4_t *wsrc = a.qweight + (size_t)wrow * nchunks + lchunk;
    const int w_wslot = lrow * 4 + (lchunk ^ ((lrow >> 2) & 3));
    const int w_rslot = r16 * 4 + (kq ^ ((r16 >> 2) & 3));
    // activation stage: thread t owns piece t of 512: k-block h = t / 256 of the stage, batch row (t % 256) / 16, piece t % 16
    const int xh = tid >> 8, xm = (tid & 255) >> 4, xpc = tid & 15;
    const half_t *xsrc = Am + (size_t)(xm < Mloc ? xm : Mloc - 1) * a.lda + xh * 128 + xpc * 8;
    const int x_wslot = xh * 256 + xpc * 16 + (xm ^ xpc);

    float4_t acc = float4_t{0.f, 0.f, 0.f, 0.f};
    unsigned mask_hi;
    asm volatile("v_mov_b32 %0, 0x00F000F0" : "=v"(mask_hi));
    const unsigned magic = 0x64006400u;
    const half8_t ones = half8_t{(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};

    // one stage: both k-blocks' weights through the wave's LDS slots, then four independent MFMA chains
    auto stage = [&](const uint4_t *xi, int g) {
        const uint4_t wq0 = lds_w[w_rslot], wq1 = lds_w[64 + w_rslot];
        float4_t blk0 = float4_t{0.f, 0.f, 0.f, 0.f}, xs0 = blk0, blk1 = blk0, xs1 = blk0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pc = kq * 4 + s;
            const half8_t xb0 = __builtin_bit_cast(half8_t, xi[pc * 16 + (r16 ^ pc)]);
            const half8_t xb1 = __builtin_bit_cast(half8_t, xi[256 + pc * 16 + (r16 ^ pc)]);
            const unsigned w = wq0[s], v = wq1[s];
            const half8_t a0 = __builtin_bit_cast(half8_t, uint4_t{((w << 4) & mask_hi) | magic, (w & mask_hi) | magic, ((w >> 4) & mask_hi) | magic, ((w >> 8) & mask_hi) | magic});
            const half8_t a1 = __builtin_bit_cast(half8_t, uint4_t{((v << 4) & mask_hi) | magic, (v & mask_hi) | magic, ((v >> 4) & mask_hi) | magic, ((v >> 8) & mask_hi) | magic});
            blk0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xb0, blk0, 0, 0, 0);
            blk1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xb1, blk1, 0, 0, 0);
            xs0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, xb0, xs0, 0, 0, 0);
            xs1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, xb1, xs1, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        const half4_t s40 = *reinterpret_cast<const half4_t *>(tab_s + g * 16 + 4 * kq);
        const half4_t c40 = *reinterpret_cast<const half4_t *>(tab_c + g * 16 + 4 * kq);
        const half4_t s41 = *reinterpret_cast<const half4_t *>(tab_s + g * 16 + 16 + 4 * kq);
        const half4_t c41 = *reinterpret_cast<const half4_t *>(tab_c + g * 16 + 16 + 4 * kq);
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // k-block order, as the autonomous kernel
            const float t0 = __builtin_fmaf(-(float)c40[i], xs0[i], blk0[i]);
            acc[i] = __builtin_fmaf((float)s40[i], t0, acc[i]);
            const float t1 = __builtin_fmaf(-(float)c41[i], xs1[i], blk1[i]);
            acc[i] = __builtin_fmaf((float)s41[i], t1, acc[i]);
        }
    };

    // stage s = k-blocks 2 s, 2 s + 1.  Weights are requested four stages ahead (register ring W), activations two stages
    // ahead (ring X) and written to the LDS image one stage ahead; every address is clamped, nothing is predicated.
    const int last = nst - 1;
    auto xload = [&](int s) { return *reinterpret_cast<const uint4_t *>(xsrc + (s < last ? s : last) * 256); };
    uint4_t W[4][2], X[2];
    X[0] = xload(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = j < last ? j : last;
        W[j][0] = wsrc[(2 * s) * 4];
        W[j][1] = wsrc[(2 * s + 1) * 4];
        if (j == 0) X[1] = xload(1);
    }
    lds_x[x_wslot] = pair_permute(X[0]);
    __syncthreads();
    for (int s4 = 0; s4 < nst; s4 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = s4 + j;
            // the ring registers are dead once their block sits in LDS, so the next request can land in the same registers
            // (requesting first would leave old and new values live together and cost copies, i.e. vmcnt(0), at the back edge)
            lds_w[w_wslot] = W[j][0];
            lds_w[64 + w_wslot] = W[j][1];
            __builtin_amdgcn_sched_barrier(0);
            X[j & 1] = xload(s + 2);
            const int sn = s + 4 < last ? s + 4 : last;
            W[j][0] = wsrc[(2 * sn) * 4];
            W[j][1] = wsrc[(2 * sn + 1) * 4];
            __builtin_amdgcn_sched_barrier(0);
            stage(lds_x + (j & 1) * 512, 2 * s);
            lds_x[((j + 1) & 1) * 512 + x_wslot] = pair_permute(X[(j + 1) & 1]);
            __syncthreads();
        }
    }

    const int m = r16;
    if (idle || m >= Mloc) return;
    const int nb = n0 + 4 * kq;
    half_t outv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) outv[i] = (half_t)(acc[i] * 0.0625f);
    half_t *crow = Cm + (size_t)m * a.ldc;
    if (a.epilogue & TCE_W4_SILU_MUL_PAIRS) {
        if (nb + 1 < a.N) crow[nb >> 1] = silu_mul_half(outv[0], outv[1]);
        if (nb + 3 < a.N) crow[(nb >> 1) + 1] = silu_mul_half(outv[2], outv[3]);
    } else if (a.epilogue & TCE_W4_ADD_TO_C) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (nb + i < a.N) crow[nb + i] = crow[nb + i] + outv[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (nb + i < a.N) crow[nb + i] = outv[i];
    }
}

thread_local int g_skinny_ks = 0;  // forced K split (tuning), 0 = automatic
thread_local int g_skinny_max_m = 128;

}  // namespace

void set_skinny_config(int ks) { g_skinny_ks = ks; }
void set_skinny_max_m(int m) { g_skinny_max_m = m; }

bool skinny_supports(const tce_w4a16_desc &d) {
    // M = 2 stays with the GEMV kernels (5.2 vs 6.5 us at 4096^2, 15.5 vs 20 us at 22016x4096); from 3 rows up this one wins
    if (!(d.M >= 3 && d.M <= g_skinny_max_m && d.group_size == 128 && d.K % 128 == 0 && !d.rmsnorm_gamma &&
          (long long)d.N * (d.K / 2) < (1LL << 40)))
        return false;
    if (d.M <= 16) return true;
    // 17 <= M <= 128: 16-row slices of the batch on gridDim.y against the MFMA GEMM, whose 64-row tiles leave most CUs idle
    // when N is small (4096 x 4096, M = 32: 64 workgroups, 24 us; here 8.8 us) and win once there are enough of them
    // (22016 x 4096 from M = 24).  Both estimates in us, fitted to profiles/r1/m_sweep_17_384.jsonl and gemm_dma_*.jsonl.
    const float scale = (float)d.N / 4096.f * (float)d.K / 4096.f;
    // (round 4, scripts/mid_m_sweep.py: 5.5 / 7.8 / 10.4 / 12.5 / 19.8 us at 1 / 2 / 3 / 4 / 6 slices on 4096 x 4096 -- the sixth slice costs more than the first five suggest,
    //  and at M = 96 the GEMM was ahead on both N = 4096 shapes: 16.1 against 19.8 us, 37.9 against 46.7 at K = 11008; never more than four slices)
    const int slices = (d.M + 15) / 16;
    if (slices > 4) return false;
    const float here = (3.0f + 2.5f * (float)slices) * scale;
    const float gemm = gemm_dma_estimate_us(d.M, d.N, d.K);  // (fitted to the same sweeps; 16.6 us measured at M = 128, 4096 x 4096 against 19.2)
    return here < gemm;
}

int launch_w4a16_skinny(const tce_w4a16_desc &d, hipStream_t stream, hipError_t *hip_err) {
    if (!skinny_supports(d)) return TCE_ERR_UNSUPPORTED_SHAPE;
    SkinnyArgs a{};
    const int zw = zeros_width(d.K, d.group_size);
    a.A = static_cast<const half_t *>(d.A);
    a.qweight = static_cast<const uint4_t *>(d.qweight);
    a.scales = static_cast<const half_t *>(d.scales);
    a.zeros = static_cast<const unsigned *>(d.zeros);
    a.C = static_cast<half_t *>(d.C);
    a.M = d.M;
    a.N = d.N;
    a.K = d.K;
    a.lda = d.lda ? d.lda : d.K;
    a.epilogue = d.flags & (TCE_W4_SILU_MUL_PAIRS | TCE_W4_ADD_TO_C);
    a.ldc = d.ldc ? d.ldc : ((a.epilogue & TCE_W4_SILU_MUL_PAIRS) ? d.N / 2 : d.N);
    a.scales_stride = d.scales_stride ? d.scales_stride : zw * 8;
    a.zeros_stride = d.zeros_stride ? d.zeros_stride : zw;
    const int nkb = d.K / 128;
    const int tiles = (d.N + 15) / 16;
    const bool z8 = (d.flags & TCE_W4_ZERO_POINT_IS_8) != 0;
    // enough tiles to fill the chip without a K split: workgroups of 8 tiles sharing the activation blocks (ks == 9 forces
    // this form, any other forced ks the autonomous one)
    const int mtiles = (d.M + 15) / 16;
    if ((g_skinny_ks == 9 || (g_skinny_ks == 0 && tiles >= 1024)) && d.K % 256 == 0) {
        const size_t lds = (size_t)2 * 2 * kXBuf + (size_t)kSharedWaves * (2 * kWBuf + (size_t)((nkb + 7) & ~7) * 64);
        if (lds <= 160 * 1024) {
            a.groups_per_wave = nkb;
            auto kfn = z8 ? w4a16_skinny_shared_kernel<true> : w4a16_skinny_shared_kernel<false>;
            if (lds > 64 * 1024) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) {
                    if (hip_err) *hip_err = e;
                    return TCE_ERR_HIP;
                }
            }
            hipLaunchKernelGGL(kfn, dim3((tiles + kSharedWaves - 1) / kSharedWaves, mtiles), dim3(64 * kSharedWaves), lds, stream, a);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) {
                if (hip_err) *hip_err = e;
                return TCE_ERR_HIP;
            }
            return TCE_OK;
        }
    }
    // waves per tile: enough waves in the launch to cover the chip a few times, each with at least two k-blocks
    int ks = g_skinny_ks == 9 ? 0 : g_skinny_ks;
    if (ks == 0) {
        ks = 1;
        while (ks < 8 && (long)tiles * mtiles * ks < 3072 && nkb / (ks * 2) >= 2) ks *= 2;
    }
    if (ks > nkb) ks = nkb;
    if (ks < 1 || ks > 8) return TCE_ERR_BAD_ARG;
    a.groups_per_wave = (nkb + ks - 1) / ks;
    size_t lds = (size_t)ks * (kWBuf + kXBuf + (size_t)a.groups_per_wave * 64);
    if (lds < (size_t)ks * 1024) lds = (size_t)ks * 1024;
    if (lds > 160 * 1024) return TCE_ERR_UNSUPPORTED_SHAPE;
    auto kfn = z8 ? w4a16_skinny_kernel<true> : w4a16_skinny_kernel<false>;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            if (hip_err) *hip_err = e;
            return TCE_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kfn, dim3(tiles, mtiles), dim3(64 * ks), lds, stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
