// w4a16_gemm_pk.hip -- W4A16 prefill GEMM for gfx950 on PRE-PACKED weights: 128 activation rows per wave.
//
// Why another GEMM (profiles/r1/rocprofv3_summary_gemm_m512.txt): the LDS-DMA kernel (w4a16_gemm_dma.hip) gives a
// wave 64 rows x 32 columns, so every weight is dequantized M / 64 times and each MFMA carries 6.6 VALU instructions
// (12 per weight word in natural nibble order, a per-k-block scale fma on a second accumulator set that also caps the
// occupancy at one wave per SIMD).  Here
//   * a wave owns 128 rows x 32 columns (8 x 2 MFMA tiles of 16x16, v_mfma_f32_16x16x32_f16): one unpacked weight fragment
//     feeds 8 MFMAs, the dequantization work per MFMA halves;
//   * the weights come from the q4_mfma copy built once per tensor by tce_w4a16_prepack (layout: w4a16_mfma_layout.hpp): a
//     wave's load of one (column tile, k-block) is one contiguous KiB, a lane's four words are its four MFMA steps of
//     CONSECUTIVE k (any group size steps inside one group -- no LDS re-deal for groups of 64 / 32), and the nibble order
//     makes the exact int4 -> fp16 unpack 9 VALU per word;
//   * ONE accumulator set: the accumulator is kept in units of the current group's scale -- before a group's MFMAs it is
//     multiplied by e[g-1] / e[g] (e = the fp16 group scale, made non-zero by the prepack), the MFMAs then accumulate the
//     exact products (q - z) * x straight into it, and e[last] is applied once at the end:
//         sum_g e_g * blk_g  ==  e_last * ((...(blk_0 * e_0/e_1 + blk_1) * e_1/e_2 + ...) + blk_last)
//     -- the same number of fp32 roundings per group as `acc += e_g * blk_g` (one), without the second 64 registers, so two
//     waves per SIMD fit (<= 256 registers);
//   * the activation tile reaches LDS by DMA (global_load_lds_dwordx4) in HALF-stages of 128 rows x 64 k (16 KiB) in a ring of
//     four per wave quartet: a half-stage is refilled (two k-blocks ahead) as soon as every wave has read it, so a quartet
//     needs 64 KiB, two workgroups fit a CU, and the in-order vmcnt that retires a block's weight words has retired both of its
//     half-stages as well;
//   * forms: KS = 1 -- 4 waves, two workgroups per CU; KS = 2 -- 8 waves on one 128x128 tile, the k-blocks alternating between
//     the two quartets (one shared barrier sequence), partial sums joined through LDS in a fixed order (quartet 0 + quartet 1).
//
// Numerics: products exact (integers |q - z| <= 15 times fp16 in fp32), fp32 accumulation on the matrix pipe, one fp32
// multiply per accumulator register and group, fp16 RNE store -- the precision class of the reference GEMV
// (kernels/cuda/gemv_cuda.cu:181-193); tolerance and oracle as for every W4A16 path (tests/test_gpu_w4a16_pk.py).
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"
#include "w4a16_mfma_layout.hpp"

namespace tce {

namespace {

// ------------------------------------------------------------------------------------------------------------------------
// prepack: q4_6 -> q4_mfma
// ------------------------------------------------------------------------------------------------------------------------
struct PrepackArgs {
    const unsigned *qweight;  // u32 [N][K/8]
    const half_t *scales;     // fp16 [N][scales_stride]
    const unsigned *zeros;    // u32 [N][zeros_stride]
    unsigned *words;          // out
    uint2_t *consts;          // out: {e as f32 bits, zc}
    float *last;              // out
    half_t *dscales;          // out: fp16 [NT16][K/G][16]
    unsigned *dzeros;         // out: u32 [NT16][K/G][2]
    int N, K, log2g, scales_stride, zeros_stride;
};

// one 64-thread block per (row tile, k-block): thread = lane (q, n16) writes its four words
__global__ __launch_bounds__(64) void prepack_words_kernel(const PrepackArgs a) {
    const int jt = blockIdx.x, kb = blockIdx.y, lane = threadIdx.x;
    const int q = lane >> 4, n16 = lane & 15;
    const int n = jt * 16 + n16;
    const bool live = n < a.N;
    const int nr = live ? n : a.N - 1;
    unsigned out[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k0 = kb * 128 + 32 * s + 8 * q;  // 8 consecutive k = one source word
        const int g = k0 >> a.log2g;
        unsigned src = a.qweight[(size_t)nr * (a.K >> 3) + (k0 >> 3)];
        const unsigned z = (a.zeros[(size_t)nr * a.zeros_stride + (g >> 3)] >> ((g & 7) * 4)) & 0xFu;
        const unsigned short sb = __builtin_bit_cast(unsigned short, a.scales[(size_t)nr * a.scales_stride + g]);
        // a group whose scale is +-0 contributes nothing in the reference (s * (q - z) == 0): its codes become the zero point
        if ((sb & 0x7FFFu) == 0u) src = z * 0x11111111u;
        if (!live) src = 0x88888888u;
        unsigned w = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) w |= ((src >> (4 * e)) & 0xFu) << (4 * pk::nibble_index(e));
        out[s] = w;
    }
    uint4_t *dst = reinterpret_cast<uint4_t *>(a.words) + ((size_t)jt * (a.K >> 7) + kb) * 64 + lane;
    *dst = uint4_t{out[0], out[1], out[2], out[3]};
}

// one thread per row: the effective scales e[g] (sequential over the groups) and the zero-point constants
__global__ __launch_bounds__(64) void prepack_consts_kernel(const PrepackArgs a) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    const int np = pk::nt16(a.N) * 16;
    if (n >= np) return;
    const int G = 1 << a.log2g, ng = a.K >> a.log2g;
    if (n >= a.N) {
        for (int g = 0; g < ng; ++g) a.consts[pk::const_index(n, g, a.K, G)] = uint2_t{__builtin_bit_cast(unsigned, 1.0f), pk::zc_word(8)};
        a.last[n] = 0.f;
        return;
    }
    float e = 1.0f;  // effective scale in front of the first group: the first non-zero scale of the row (or 1)
    for (int g = 0; g < ng; ++g) {
        const float s = (float)a.scales[(size_t)n * a.scales_stride + g];
        if (s != 0.f) {
            e = s;
            break;
        }
    }
    for (int g = 0; g < ng; ++g) {
        const float s = (float)a.scales[(size_t)n * a.scales_stride + g];
        if (s != 0.f) e = s;
        const unsigned z = (a.zeros[(size_t)n * a.zeros_stride + (g >> 3)] >> ((g & 7) * 4)) & 0xFu;
        a.consts[pk::const_index(n, g, a.K, G)] = uint2_t{__builtin_bit_cast(unsigned, e), pk::zc_word(z)};
    }
    a.last[n] = e;
}

// one thread per (row tile, group): the decode kernel's side arrays (w4a16_gemv_i8.hip) -- the fp16 scales and the zero points of the tile's 16 rows, side by side
__global__ __launch_bounds__(64) void prepack_decode_kernel(const PrepackArgs a) {
    const int ng = a.K >> a.log2g;
    const long long idx = (long long)blockIdx.x * 64 + threadIdx.x;
    if (idx >= (long long)pk::nt16(a.N) * ng) return;
    const int jt = (int)(idx / ng), g = (int)(idx % ng);
    unsigned zw[2] = {0u, 0u};
    unsigned short sb[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = jt * 16 + i;
        unsigned z = 8u;
        sb[i] = 0;
        if (n < a.N) {
            z = (a.zeros[(size_t)n * a.zeros_stride + (g >> 3)] >> ((g & 7) * 4)) & 0xFu;
            sb[i] = __builtin_bit_cast(unsigned short, a.scales[(size_t)n * a.scales_stride + g]);
        }
        zw[i >> 3] |= z << (4 * (i & 7));
    }
    uint4_t *ds = reinterpret_cast<uint4_t *>(a.dscales) + (size_t)idx * 2;
    ds[0] = uint4_t{sb[0] | ((unsigned)sb[1] << 16), sb[2] | ((unsigned)sb[3] << 16), sb[4] | ((unsigned)sb[5] << 16), sb[6] | ((unsigned)sb[7] << 16)};
    ds[1] = uint4_t{sb[8] | ((unsigned)sb[9] << 16), sb[10] | ((unsigned)sb[11] << 16), sb[12] | ((unsigned)sb[13] << 16), sb[14] | ((unsigned)sb[15] << 16)};
    reinterpret_cast<uint2_t *>(a.dzeros)[idx] = uint2_t{zw[0], zw[1]};
}

// ------------------------------------------------------------------------------------------------------------------------
// the GEMM
// ------------------------------------------------------------------------------------------------------------------------
typedef float float2_t __attribute__((ext_vector_type(2)));
constexpr int kPkFaultWord = 1023;  // last word of the scratch area's counter page
constexpr unsigned kPkPoison = 0x80000000u;  // a tile counter whose hand-off ran out of patience (sticky until the host clears the page)
struct PkGemmArgs {
    const half_t *A;
    const uint4_t *words;   // [NT16][NKB][64]
    const uint2_t *consts;  // [NT16][K/G][16]
    half_t *C;
    int M, N, K, lda, ldc;
    int n_blocks, m_blocks;  // 128 x 128 tiles
    int add_to_c;
    int pairs;  // TCE_W4_SILU_MUL_PAIRS: output column n / 2 = SiLuMul_half(column n, column n + 1) for even n; C is [M][N / 2]
    int xm, m_per, n_per;
    int split_s;          // > 1: the k-blocks of the tiles in slots >= full_slots are cut into split_s runs, one workgroup each (one quartet,
    int full_slots;       //      128 x 128 tiles only); slot = workgroup index / 8 of the tile's first run.  0: every tile is cut
    int handoff;          // split_s == 2 only (round 5): run 0 takes a slightly SHORTER part of the k range, writes its partial tile through and raises the tile's counter; run 1 --
                          // dispatched behind run 0 of the same tile: block indices grow with the run -- finishes its longer part, finds the counter raised (or waits a bounded
                          // time), adds run 0's tile to its own in run order and stores.  Off the critical path: run 1's write-through, the counter's round trip
    int prio;             // != 0: the two waves sharing a SIMD run at different priorities for the whole kernel (launch_w4a16_gemm_pk sets it for the one-quartet wide form)
    unsigned *counters;   // [cut tiles], zero between launches (scratch); word kPkFaultWord counts hand-offs whose wait ran out (never seen; the tests hold it at zero)
    float4_t *partials;   // [cut tiles][split_s][16 accumulators][256 threads] fp32 x 4 (scratch)
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_void_t;

__device__ __forceinline__ void pk_dma16(const void *src, void *lds_dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((global_void_t *)src, (lds_void_t *)lds_dst_wave_uniform, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void pk_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int MASK, int COUNT>
__device__ __forceinline__ void sched_group() {
    __builtin_amdgcn_sched_group_barrier(MASK, COUNT, 0);
}

// 16-row MFMA tiles per wave: template parameter MT of the kernel body -- 8 (128 rows: rounds 2-4) or 16 (256 rows, round 5: every unpacked weight word and every
// landed asm load feeds twice the MFMAs; one wave per SIMD, the 128 accumulator registers beside double-buffered A fragments the compiler may keep in AGPRs)
// 16-column MFMA tiles per wave: template parameter NT of the kernel body -- 2 (32 columns; 4 waves side by side = 128 columns: rounds 2-5) or 4 (round 5, the "wide" form:
// 128 rows x 64 columns per wave, 128 x 256 per quartet -- every A fragment read from LDS feeds FOUR MFMAs and every activation DMA twice the MFMAs of the other forms; what
// bounds those is not the vector ALU but the fragment reads, the DMA issue and the barriers, which serialise with the MFMA stream of an in-order wave: section 3.2 of DESIGN.md)

// The compiler allocates v0 .. v231 only; v232 .. v255 belong to the inline asm below (a block's words and constants in flight).
constexpr int kAsmVgprBase = 232;  // 128-row forms (and every group size); the 256-row form does not reserve registers at all (round 5: amdgpu_num_vgpr turned out to be a HINT in this toolchain -- a kernel under pressure is given v232 .. v255 as well; the 128-row forms stay below 232 by themselves, checked in the ISA)
// ABL: timing experiments only (tce_w4a16_set_debug_mode(600 + ABL); results are then meaningless): bit 0 no rescale, 1 no unpack,
// 2 no fragment reads, 3 no MFMAs, 4 no activation DMAs, 5 no barriers.
// NS = 2: two wave quartets side by side on a 128 x 256 tile, sharing ONE activation ring (KS = 1 then): the activation bytes a CU
// pulls through L2 -> LDS per MFMA halve.  That path, not the matrix pipe, bounds the 128 x 128 forms: the activation DMAs of
// M = 2048, 4096 x 4096 alone take 32 us of the 68 (profiles/r2/gemm_pk_ablation.jsonl; ~64 GB/s per CU), the MFMAs alone 38.
template <int KS, int LG, int ABL, int NS, int kMT, int kNT = 2>
__device__ __forceinline__ void w4a16_gemm_pk_body(const PkGemmArgs &g) {
    static_assert(KS == 1 || NS == 1, "two quartets either split K or sit side by side");
    static_assert(kNT == 2 || ((kNT == 4 || kNT == 3) && kMT == 8 && (NS == 1 || kNT == 4) && LG == 7), "wide form: 128 rows x 64 (48) columns per wave, one quartet per 128 x 256 (192) tile (or two alternating its k-blocks), groups of 128");
    constexpr bool WIDE = kNT >= 3;  // (kNT == 3: 128 x 192 tiles -- 58 column blocks on N = 11008 where 256-wide tiles make 43: 232 tiles instead of 172 for the 256 CUs at M = 512)
    static_assert(kMT == 8 || (kMT == 16 && (NS == 1 || KS == 1)), "256-row wave tiles: one quartet per tile, two quartets splitting the k-blocks of one tile, or two quartets side by side on one activation ring");
    constexpr int ROWS = 16 * kMT;                   // rows of the activation tile
    constexpr int HALF_BYTES = ROWS * pk::kHalfK * 2;  // one half-stage of activations: 16 / 32 KiB
    static_assert(kMT == 8 || LG == 7, "256-row wave tiles: groups of 128 only (12 landing registers for the asm loads)");
    constexpr int ASMB = kAsmVgprBase;  // first register of the 128-row forms' landing area (the 256-row form lands its asm loads in ordinary variables: see `inflight`)
    constexpr int NTHREADS = 256 * KS * NS;
    constexpr int AB = ABL & 63;         // loop parts switched off; bit 6 (64): the rescale as four scalar multiplies per tile (this round's first form, for the A/B)
    constexpr int NWR = 4 * NS;          // waves feeding (and reading) one ring
    constexpr int DPW = (ROWS / 8) / NWR;  // DMA instructions per wave and half-stage (8 rows each)
    constexpr int BN = 64 * kNT * NS;
    constexpr int GPB = 128 >> LG;  // groups per k-block
    constexpr int SPG = 4 / GPB;    // MFMA steps per group
    // ring depth in half-stages per quartet: four (a half-stage is refilled two k-blocks ahead) -- or TWO for the 256-row tile shared by two quartets (round 5:
    // 2 x 2 x 32 KiB = the 128 KiB one workgroup gets): a slot is refilled with the NEXT block's half as soon as its quartet has read it, half a k-block ahead of its
    // first use -- which, with two waves sharing every SIMD, is a whole lone-quartet k-block of wall-clock time
    constexpr int RD = (kMT == 16 && KS == 2) ? 2 : 4;
    constexpr int QUARTET_BYTES = RD * HALF_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // K split across workgroups: part p of a tile is workgroup p * (grid / split_s) + (the tile's index): bid % 8 -- the XCD the
    // hardware puts a workgroup on -- stays what the tile mapping below assumes
    // Which tiles are cut: those in slots >= full_slots, i.e. the ones whose first runs are dispatched LAST -- a launch of 344 tiles
    // runs 256 whole tiles (one per CU) and the k-blocks of the other 88 as 3 x 88 short runs beside them, instead of two whole tiles
    // on 88 CUs and one on the rest.
    // the forms whose tiles' k range may be cut across workgroups: one quartet per tile (any tile shape), and -- round 6, form 16 -- the 128 x 128 tile whose k-blocks
    // alternate between two quartets: each RUN alternates between them, the quartets join through LDS as always, quartet 0 carries the exchange between the runs
    constexpr bool CUT = NS == 1 && (KS == 1 || (KS == 2 && kMT == 8 && kNT == 2));
    int bid = blockIdx.x, part = 0;
    if (CUT && g.split_s > 1) {
        const int per = 8 * g.m_per * g.n_per;
        if (bid >= per) {
            const int tail_wgs = per - 8 * g.full_slots, e = bid - per;
            part = 1 + e / tail_wgs;
            bid = 8 * g.full_slots + (e - (part - 1) * tail_wgs);
        }
    }
    const int xcd = bid & 7, slot = bid >> 3;
    if (g.prio) {  // (the second quartet of a workgroup / the workgroups of the second dispatch round of a CU -- 32 CUs per XCD -- give way)
        const bool low = (KS == 2 || NS == 2) ? (int)(threadIdx.x >> 8) == 1 : (((g.prio == 4 ? slot : slot >> 5)) & 1) == 1;
        if (!low) {
            if (g.prio == 2) __builtin_amdgcn_s_setprio(3);
            else if (g.prio == 3) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(2);
        }
    }
    const int m_blk = (xcd % g.xm) * g.m_per + slot % g.m_per;
    const int n_blk = (xcd / g.xm) * g.n_per + slot / g.m_per;
    if (n_blk >= g.n_blocks || m_blk >= g.m_blocks) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = KS == 2 ? (wave8 & 3) : wave8;  // index within the ring's waves: DMA share and column position
    const int grp = KS == 2 ? wave8 >> 2 : 0;
    unsigned char *const ring = smem + grp * QUARTET_BYTES;
    const int n16 = lane & 15, q = lane >> 4;
    const int m_base = m_blk * ROWS, nb0 = n_blk * BN;
    const int nkb = g.K >> 7;
    const int split = (CUT && g.split_s > 1 && slot >= g.full_slots) ? g.split_s : 1;
    // this workgroup's run of k-blocks (the hand-off form: run 0 is two k-blocks shorter than run 1 -- about the time its write-through takes to become visible)
    const bool handoff = CUT && kMT == 8 && kNT == 2 && split == 2 && g.handoff != 0;
    // (g.handoff = 1 + the k-blocks run 0 is shorter by; never fewer than two k-blocks for run 0 -- ADVICE r5: an unclamped delta of 8 at K = 1024 left run 0 without a k-block)
    const int n0_raw = nkb >= 8 ? (nkb - (g.handoff - 1)) / 2 : nkb / 2;
    const int n0_h = n0_raw >= 2 ? n0_raw : (nkb >= 4 ? 2 : nkb / 2);
    const int kb_lo = handoff ? (part == 0 ? 0 : n0_h) : part * nkb / split;
    const int nloc = handoff ? (part == 0 ? n0_h : nkb - n0_h) : (part + 1) * nkb / split - kb_lo;
    const int T = (nloc + KS - 1) / KS;  // iterations (own k-blocks, the last one may be past the run for quartet 1)

    // ---- DMA sources of a half-stage: instruction ii of this wave fills 8 rows; the lane fetches the piece that belongs at its position ----
    // (uniform 64-bit base + 32-bit lane offset: the address arithmetic of a refill is scalar; launch_w4a16_gemm_pk refuses M * lda * 2 >= 4 GiB)
    constexpr bool BUFDMA = true;  // the activation DMAs through a buffer descriptor (one lane offset, everything else scalar) -- the 256-row and wide forms since round 5, every form since round 6 (the narrow forms' per-piece 64-bit address adds were 12 of a k-block's 151 vector instructions: 512 x 4096 x 4096 26.2 -> 26.1 us, 384 rows 25.05 -> 24.56, x 11008 54.0 -> 53.35; two builds alternating)
    unsigned a_voff[BUFDMA ? 1 : DPW];
#pragma unroll
    for (int ii = 0; ii < (BUFDMA ? 0 : DPW); ++ii) {
        const int row = pk::dma_row(wave, ii, lane, NWR);
        int m = m_base + row;
        m = m < g.M ? m : g.M - 1;  // rows past M repeat the last row; their outputs are not stored
        a_voff[ii] = (unsigned)m * (unsigned)(g.lda * 2) + (unsigned)(pk::dma_src_piece(row, lane) << 4);
    }
    const char *const a_bytes = reinterpret_cast<const char *>(g.A);
    // 256-row form: the same pieces through a buffer descriptor over A -- ONE lane offset (row-in-eight x pitch + swizzled piece; the swizzle term (row >> 1) & 7 does
    // not depend on the instruction: rows advance by 32 per instruction), everything else in the scalar offset; rows past M are out of range and read as zeros
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(g.A), 0, (int)((unsigned)g.M * (unsigned)(g.lda * 2)), 0x00020000);
    const unsigned a_lane_off = (unsigned)(lane >> 3) * (unsigned)(g.lda * 2) + (unsigned)(((lane & 7) ^ (((wave & 1) * 4 + (lane >> 4)) & 7)) << 4);
    (void)rs_a;
    (void)a_lane_off;
    // own half-block h (0 .. 2T-1): k-block grp + (h >> 1) * KS, half h & 1; clamped, never predicated (the counted waits rely on it)
    auto issue_half = [&](int h) {
        if constexpr (AB & 16) return;
        int kb = grp + (h >> 1) * KS;
        kb = kb_lo + (kb < nloc ? kb : nloc - 1);
        const char *src = a_bytes + ((size_t)kb * 256 + (h & 1) * 128);  // wave-uniform
        unsigned char *st = ring + (h & (RD - 1)) * HALF_BYTES;
        if constexpr (BUFDMA) {
            const unsigned s0 = (unsigned)(m_base + wave * 8) * (unsigned)(g.lda * 2) + (unsigned)(kb * 256 + (h & 1) * 128);
#pragma unroll
            for (int ii = 0; ii < DPW; ++ii)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void_t *)(st + pk::dma_lds_base(wave, ii, NWR)), 16, a_lane_off, s0 + (unsigned)ii * (unsigned)(NWR * 8 * g.lda * 2), 0, 0);
        } else {
#pragma unroll
            for (int ii = 0; ii < DPW; ++ii) pk_dma16(src + a_voff[ii], st + pk::dma_lds_base(wave, ii, NWR));
        }
    };
    // one instruction of a half-stage's refill (wide form: dealt out between the MFMA groups of the region behind the barrier instead of going out back to back --
    // a DMA instruction holds the wave's issue port for 60-180 cycles, MI355X_MICROARCH.md, and four in a row leave the matrix pipe idle for most of that)
    auto issue_piece = [&](int h, int ii) {
        if constexpr (BUFDMA && !(AB & 16)) {
            int kb = grp + (h >> 1) * KS;
            kb = kb_lo + (kb < nloc ? kb : nloc - 1);
            unsigned char *st = ring + (h & (RD - 1)) * HALF_BYTES;
            const unsigned s0 = (unsigned)(m_base + wave * 8) * (unsigned)(g.lda * 2) + (unsigned)(kb * 256 + (h & 1) * 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void_t *)(st + pk::dma_lds_base(wave, ii, NWR)), 16, a_lane_off, s0 + (unsigned)ii * (unsigned)(NWR * 8 * g.lda * 2), 0, 0);
        }
    };
    // ---- the lane's weights and constants ----
    const int jt0 = n_blk * (BN / 16) + wave * kNT;  // first 16-column tile of this wave
    const int ntiles16 = pk::nt16(g.N);
    const uint4_t *w_tile[kNT];  // wave-uniform bases; lane offsets lane * 16 / n16 * 8 bytes
    const uint2_t *c_tile[kNT];
#pragma unroll
    for (int j = 0; j < kNT; ++j) {
        int jt = jt0 + j;
        jt = jt < ntiles16 ? jt : ntiles16 - 1;  // tiles past N: clamped loads, nothing stored
        w_tile[j] = g.words + (size_t)jt * nkb * 64;
        c_tile[j] = g.consts + (size_t)jt * (nkb * GPB) * 16;
    }
    const unsigned w_voff = lane * 16, c_voff = n16 * 8;
    int a_off[2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) a_off[sl] = pk::frag_offset(0, n16, q, sl);  // + i * 2048 per m-tile

    // (a VALU instruction of this chip reads ONE scalar / literal operand: of the two constants of `(w & mask) | 0x6400_6400` one lives in a vector register.  The narrow
    // forms keep the two masks there; the wide form, which has no register to spare, keeps the 0x6400 pair and takes both masks from scalar registers)
    unsigned mask_lo = 0x000F000Fu, k64 = 0x64006400u;
    unsigned k2v = 0xD480D480u;  // -(64 + 8) twice: the third operand of the packed fma beside a scalar 1/16 -- in a vector register like k64
    if constexpr (WIDE) asm volatile("v_mov_b32 %0, 0x64006400\n\tv_mov_b32 %1, 0xD480D480" : "=v"(k64), "=v"(k2v));
    else asm volatile("v_mov_b32 %0, 0x000F000F" : "=v"(mask_lo));
    const unsigned mask_hi = mask_lo << 4;
    const half2_t sixteenth = as_half2(0x2C002C00u);

    float4_t acc[kMT][kNT];
#pragma unroll
    for (int i = 0; i < kMT; ++i)
#pragma unroll
        for (int j = 0; j < kNT; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    float e_prev[kNT];  // effective scale the accumulator is currently expressed in
#pragma unroll
    for (int j = 0; j < kNT; ++j) e_prev[j] = 0.f;  // "no group yet": the first ratio is forced to 1

    // (wide form: offered for linears whose zero points are all 8 -- TCE_W4_ZERO_POINT_IS_8, what the reference's quantizer writes -- so a block's constants are the
    //  effective scales alone: 4 landing registers + 4 current ones less than with the packed zero-point words; other linears keep the narrow forms)
    struct CW {
        unsigned x;
    };
    using const_t = std::conditional_t<WIDE, CW, uint2_t>;
    struct BlockRegs {
        uint4_t w[kNT];
        const_t c[kNT][GPB];
    };
    BlockRegs cur;
    auto block_of = [&](int t) {
        const int kb = grp + t * KS;
        return kb_lo + (kb < nloc ? kb : nloc - 1);
    };
    // the first block's registers by ordinary loads (the compiler waits for them), everything later by inline asm one block ahead
    {
        const int kb = block_of(0);
#pragma unroll
        for (int j = 0; j < kNT; ++j) {
            cur.w[j] = w_tile[j][(size_t)kb * 64 + lane];
#pragma unroll
            for (int gi = 0; gi < GPB; ++gi) {
                const uint2_t c2_ = c_tile[j][(size_t)(kb * GPB + gi) * 16 + n16];
                if constexpr (WIDE) cur.c[j][gi].x = c2_.x;
                else cur.c[j][gi] = c2_;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue_half(0);
    issue_half(1);
    if constexpr (RD == 4) {
        issue_half(2);
        issue_half(3);
    }

    // ---- pieces of a k-block ----
    // fragments of the step being multiplied and of the next one.  kMT == 16: ONE set of A fragments (64 registers beside the 128 accumulators), refilled tile by
    // tile -- m-tile i's fragment of the next step is read as soon as its two MFMAs of this step have issued
    constexpr bool ROLL = kMT == 16;
    constexpr bool INFL = ROLL || WIDE;  // the next block's words land in ordinary variables behind a counted wait (see `inflight`); one set of A / B fragments
    half8_t af[INFL ? 1 : 2][kMT], bf[INFL ? 1 : 2][kNT];  // (256-row form: ONE set of B fragments too -- the next step's are unpacked behind this step's last MFMA)
    half2_t c1[kNT], c2[kNT];        // (-(1024+z)) x2 and (-(64+z)) x2 of the group being unpacked
    auto read_a = [&](half8_t (&dst)[kMT], const unsigned char *half_stage, int sl) {
        const unsigned char *st = half_stage + a_off[sl];
#pragma unroll
        for (int i = 0; i < kMT; ++i) dst[i] = *reinterpret_cast<const half8_t *>(st + i * 2048);
    };
    auto set_group = [&](const BlockRegs &br, int gi) {
#pragma unroll
        for (int j = 0; j < kNT; ++j) {
            unsigned zc = 0;
            if constexpr (!WIDE) zc = br.c[j][gi].y;
            c1[j] = as_half2(__builtin_amdgcn_perm(zc, zc, 0x01000100u));  // low half twice
            c2[j] = as_half2(__builtin_amdgcn_perm(zc, zc, 0x03020302u));  // high half twice
        }
    };
    auto unpack = [&](half8_t (&dst)[kNT], const BlockRegs &br, int s) {
#pragma unroll
        for (int j = 0; j < kNT; ++j) {
            const unsigned w = br.w[j][s];
            const unsigned sh = w >> 8;
            const half2_t d0 = as_half2((w & mask_lo) | 0x64006400u) + c1[j];
            const half2_t d1 = __builtin_elementwise_fma(as_half2((w & mask_hi) | 0x64006400u), sixteenth, c2[j]);
            const half2_t d2 = as_half2((sh & mask_lo) | 0x64006400u) + c1[j];
            const half2_t d3 = __builtin_elementwise_fma(as_half2((sh & mask_hi) | 0x64006400u), sixteenth, c2[j]);
            dst[j] = half8_t{d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, d3.x, d3.y};
        }
    };
    // wide form: one column tile's fragment of step s; the zero-point constants come straight from the block's zc word -- the two splats fold into op_sel of the packed
    // add / fma (no v_perm, no c1 / c2 registers)
    auto unpack_col = [&](half8_t &dst, const BlockRegs &br, int s, int j) {
        const half2_t k1 = as_half2(0xE408E408u), k2 = as_half2(k2v);  // -(1024 + 8), -(64 + 8)
        const unsigned w = br.w[j][s];
        const unsigned sh = w >> 8;
        const half2_t d0 = as_half2((w & 0x000F000Fu) | k64) + k1;
        const half2_t d1 = __builtin_elementwise_fma(as_half2((w & 0x00F000F0u) | k64), sixteenth, k2);
        const half2_t d2 = as_half2((sh & 0x000F000Fu) | k64) + k1;
        const half2_t d3 = __builtin_elementwise_fma(as_half2((sh & 0x00F000F0u) | k64), sixteenth, k2);
        dst = half8_t{d0.x, d0.y, d1.x, d1.y, d2.x, d2.y, d3.x, d3.y};
    };
    auto rescale = [&](const float (&e_new)[kNT]) {  // acc <- acc * e_prev / e_g: the accumulator moves into the units of group g
#pragma unroll
        for (int j = 0; j < kNT; ++j) {
            // e != 0 by construction of the table; 1-ulp reciprocal (an IEEE division is ~10 instructions per group and column tile)
            float r = e_prev[j] == 0.f ? 1.0f : e_prev[j] * __builtin_amdgcn_rcpf(e_new[j]);
            e_prev[j] = e_new[j];
            // as two-element vector products: v_pk_mul_f32, two per accumulator tile (left to itself hipcc emits four v_mul_f32 here --
            // 64 instead of 32 VALU instructions per group beside the MFMAs -- while it packs the same loop after the k-loop)
            if constexpr (ABL & 64) {
#pragma unroll
                for (int i = 0; i < kMT; ++i)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) acc[i][j][rr] *= r;
            } else {
                const float2_t r2{r, r};
#pragma unroll
                for (int i = 0; i < kMT; ++i) {
                    const float2_t lo = float2_t{acc[i][j][0], acc[i][j][1]} * r2, hi = float2_t{acc[i][j][2], acc[i][j][3]} * r2;
                    acc[i][j] = float4_t{lo.x, lo.y, hi.x, hi.y};
                }
            }
        }
    };
    // Region s of a block = the 16 MFMAs of step s beside the fragment reads and the unpack of the NEXT step (for s = 3: step 0 of
    // the next block) and, at the first step of a group, the rescale.  The issue order is pinned: the matrix pipe leaves room
    // for about two 4-cycle VALU instructions per 16-cycle MFMA (scripts/probes/valu_probe.hip: T = 16 + 4.2 * max(0, v - 2)
    // cycles per MFMA with v and_or / packed instructions beside it, whatever the number of waves), so the VALU work and the LDS
    // reads are dealt out between the MFMAs instead of being issued as blocks between MFMA runs.
    float e_grp[GPB][kNT];  // effective scales of the block being multiplied (kept apart: `cur` moves on before step 3)
    auto load_e = [&](const BlockRegs &br) {
#pragma unroll
        for (int gi = 0; gi < GPB; ++gi)
#pragma unroll
            for (int j = 0; j < kNT; ++j) e_grp[gi][j] = __builtin_bit_cast(float, br.c[j][gi].x);
    };
    auto region = [&](auto s_c, const BlockRegs &br_next, const unsigned char *half_stage_next, int dma_h = 0) {
        constexpr int s = decltype(s_c)::value;
        constexpr int sn = (s + 1) & 3;  // the step whose fragments are fetched here
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WIDE) {
            // wide form (128 rows x 64 columns per wave): two phases of four m-tiles, each walked column tile by column tile -- eight fenced groups of four MFMAs:
            //   [the group's share of the rescale] [its 4 MFMAs] [behind a phase's LAST group: its four m-tiles' fragments of the NEXT step, into the registers those
            //   MFMAs just read -- sixteen MFMAs ahead of their use] [second phase: the next step's fragment of THIS column tile, which no MFMA of this step reads again].
            // One set of A fragments (32 registers) and one of B fragments (16) beside the 128 accumulators; 36 unpack instructions per 32 MFMAs.
            float r[kNT];
            if constexpr (s % SPG == 0 && !(AB & 1)) {
#pragma unroll
                for (int j = 0; j < kNT; ++j) {
                    const float en = e_grp[s / SPG][j];
                    r[j] = e_prev[j] == 0.f ? 1.0f : e_prev[j] * __builtin_amdgcn_rcpf(en);
                    e_prev[j] = en;
                }
            }
            // (the second local step's fragment offset is the first one's with bit 6 flipped -- piece index + 4 under the swizzle; made here, by an instruction the
            //  compiler may not hoist, instead of living in a register across the loop)
            int a_off_sn = a_off[0];
            if constexpr (sn & 1) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(a_off_sn) : "v"(a_off[0]));
            const unsigned char *st_next = half_stage_next + a_off_sn;
            auto rescale_group = [&](auto u_c) {  // the four tiles of group u move into the new group's units
                constexpr int ph = decltype(u_c)::value / kNT, j = decltype(u_c)::value % kNT;
                const float2_t r2{r[j], r[j]};
#pragma unroll
                for (int i = 4 * ph; i < 4 * ph + 4; ++i) {
                    const float2_t lo = float2_t{acc[i][j][0], acc[i][j][1]} * r2, hi = float2_t{acc[i][j][2], acc[i][j][3]} * r2;
                    acc[i][j] = float4_t{lo.x, lo.y, hi.x, hi.y};
                }
            };
            constexpr bool RESC = s % SPG == 0 && !(AB & 1);
            if constexpr (s >= 2 && (ABL & 256)) {
#pragma unroll
                for (int ii = 0; ii < DPW; ++ii) issue_piece(dma_h, ii);
            }
            if constexpr (RESC) rescale_group(std::integral_constant<int, 0>{});
            static_for<0, 2 * kNT>([&](auto u_c) {
                constexpr int u = decltype(u_c)::value, ph = u / kNT, j = u % kNT;
                // the NEXT group's rescale rides beside this group's MFMAs (two packed multiplies behind each MFMA), the first group's stands in front of the region
                if constexpr (RESC && u + 1 < 2 * kNT) rescale_group(std::integral_constant<int, u + 1>{});
                if constexpr (!(AB & 8)) {
#pragma unroll
                    for (int i = 4 * ph; i < 4 * ph + 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 4 * ph; i < 4 * ph + 4; ++i) asm volatile("" ::"v"(af[0][i]));
                    asm volatile("" ::"v"(bf[0][j]));
                }
                // regions 2 / 3 stand behind the barriers that free a half-stage: its refill, one instruction per group.  (placement A/B, same results: ABL bit 7 -- beside the
                // second phase's groups, where the unpack keeps the vector ALU busy; bit 8 -- all in front of the region, the first version)
                if constexpr (s >= 2 && (ABL & 384) == 0 && u < DPW) issue_piece(dma_h, u);
                if constexpr (s >= 2 && (ABL & 128) && u >= 2 * kNT - DPW) issue_piece(dma_h, u - (2 * kNT - DPW));
                if constexpr (j == kNT - 1 && !(AB & 4)) {
#pragma unroll
                    for (int i = 4 * ph; i < 4 * ph + 4; ++i) af[0][i] = *reinterpret_cast<const half8_t *>(st_next + i * 2048);
                }
                if constexpr (ph == 1 && !(AB & 2)) unpack_col(bf[0][j], br_next, sn, j);
                if constexpr (AB == 0) {  // MFMA, then its share of the group's vector instructions (rescale: 8; unpack: 9)
                    constexpr int nv = (RESC && u + 1 < 2 * kNT ? 8 : 0) + (ph == 1 ? 9 : 0);
                    static_for<0, 4>([&](auto m_c) {
                        constexpr int m = decltype(m_c)::value;
                        sched_group<0x008, 1>();
                        if constexpr ((nv * (m + 1)) / 4 - (nv * m) / 4 > 0) sched_group<0x002, (nv * (m + 1)) / 4 - (nv * m) / 4>();
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            return;
        }
        if constexpr (ROLL) {
            // 256-row form: four chunks of four m-tiles, each fenced -- [the chunk's share of the rescale] [its 8 MFMAs] [its tiles' fragments of the NEXT step into
            // the registers those MFMAs just read] [a share of the next step's unpack].  (MFMA and VALU issue serialise on a SIMD, scripts/probes/valu_probe.hip: what
            // matters is that nothing waits, not the fine interleave; the fences keep the scheduler from hoisting the fragment reads over the MFMAs that still need
            // the old values -- 64 live fragment registers, not 128 -- which sched_group_barrier sequences did not hold it to.)
            float r[kNT];
            if constexpr (s % SPG == 0 && !(AB & 1)) {
#pragma unroll
                for (int j = 0; j < kNT; ++j) {
                    const float en = e_grp[s / SPG][j];
                    r[j] = e_prev[j] == 0.f ? 1.0f : e_prev[j] * __builtin_amdgcn_rcpf(en);
                    e_prev[j] = en;
                }
            }
            const unsigned char *st_next = half_stage_next + a_off[sn & 1];
            static_for<0, 4>([&](auto c_c) {
                constexpr int c = decltype(c_c)::value;
                if constexpr (s % SPG == 0 && !(AB & 1)) {
#pragma unroll
                    for (int i = 4 * c; i < 4 * c + 4; ++i)
#pragma unroll
                        for (int j = 0; j < kNT; ++j) {
                            const float2_t r2{r[j], r[j]};
                            const float2_t lo = float2_t{acc[i][j][0], acc[i][j][1]} * r2, hi = float2_t{acc[i][j][2], acc[i][j][3]} * r2;
                            acc[i][j] = float4_t{lo.x, lo.y, hi.x, hi.y};
                        }
                }
                if constexpr (!(AB & 8)) {
#pragma unroll
                    for (int i = 4 * c; i < 4 * c + 4; ++i)
#pragma unroll
                        for (int j = 0; j < kNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 4 * c; i < 4 * c + 4; ++i) asm volatile("" ::"v"(af[0][i]));
#pragma unroll
                    for (int j = 0; j < kNT; ++j) asm volatile("" ::"v"(bf[0][j]));
                }
                if constexpr (!(AB & 4)) {
#pragma unroll
                    for (int i = 4 * c; i < 4 * c + 4; ++i) af[0][i] = *reinterpret_cast<const half8_t *>(st_next + i * 2048);
                }
                if constexpr (c == 3 && !(AB & 2)) {
                    if constexpr (sn % SPG == 0) set_group(br_next, sn / SPG);
                    unpack(bf[0], br_next, sn);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            return;
        }
        if constexpr (s % SPG == 0 && !(AB & 1)) rescale(e_grp[s / SPG]);
        if constexpr (!(AB & 4) && !ROLL) read_a(af[(s + 1) & 1], half_stage_next, sn & 1);
        if constexpr (!(AB & 2)) {
            if constexpr (sn % SPG == 0) set_group(br_next, sn / SPG);
            unpack(bf[(s + 1) & 1], br_next, sn);
        }
        if constexpr (!(AB & 8)) {
#pragma unroll
            for (int i = 0; i < kMT; ++i)
#pragma unroll
                for (int j = 0; j < kNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
        } else {  // keep what the MFMAs would have consumed alive
#pragma unroll
            for (int i = 0; i < kMT; ++i) asm volatile("" ::"v"(af[(s + 1) & 1][i]));
#pragma unroll
            for (int j = 0; j < kNT; ++j) asm volatile("" ::"v"(bf[(s + 1) & 1][j]));
        }
        if constexpr (AB == 0) static_for<0, kMT * kNT>([&](auto u_c) {
            constexpr int u = decltype(u_c)::value;
            if constexpr (s % SPG == 0) sched_group<0x002, (ABL & 64) ? 4 : 2>();   // the tile's two packed multiplies, then its MFMA
            sched_group<0x008, 1>();                               // MFMA
            if constexpr (u < kMT) sched_group<0x100, 1>();        // DS read
            sched_group<0x002, (u < 4 ? 2 : 1)>();                 // VALU: unpack of the next step (18 + constants)
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    // A block's words and constants are requested TWO blocks ahead by inline asm (hipcc would sink ordinary loads to their use and
    // answer them with vmcnt(0), draining the DMAs) -- into the registers v232.. that the compiler never allocates
    // (amdgpu_num_vgpr above) and therefore never sees as values: an asm load's VGPR destination counts as written at the end of
    // the statement, so hipcc is free to copy it (loop-carried phi, coalescing) before the data has landed; that is what a first
    // version of this loop did, silently.  (Accumulator registers would do as well, but any AGPR use makes hipcc split the 256
    // registers 128 / 128 and move the MFMA accumulators behind v_accvgpr copies.)  After the counted wait at the end of a block
    // the landed words are copied out (12 v_mov per k-block for groups of 128) into ordinary variables, and the same registers
    // take the next request.
#define TCE_PK_CLOB232 "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
    static_assert(kNT == 2 || INFL, "the request / collect macros below spell out two column tiles");
#define TCE_PK_REQ_W(J, CLOB)                                                                                                        \
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 v[%c2:%c3], %0, %1" ::"v"(w_voff), "s"(w_tile[J] + (size_t)kbn * 64), "i"(ASMB + 4 * J), "i"(ASMB + 4 * J + 3) \
                 : "memory", CLOB);
#define TCE_PK_REQ_C(J, GI, CLOB)                                                                                                    \
    if constexpr (GI < GPB)                                                                                                          \
        asm volatile("s_nop 4\n\tglobal_load_dwordx2 v[%c2:%c3], %0, %1" ::"v"(c_voff), "s"(c_tile[J] + (size_t)(kbn * GPB + GI) * 16),   \
                     "i"(ASMB + 4 * kNT + 2 * (J * GPB + GI)), "i"(ASMB + 4 * kNT + 2 * (J * GPB + GI) + 1)                                        \
                     : "memory", CLOB);
    auto request_block = [&](int t_of_block) {
        const int kbn = block_of(t_of_block);
        // (s_nop 4: the scalar base was just computed by SALU instructions, and nothing pads an SGPR hazard inside an asm statement)
        {
            TCE_PK_REQ_W(0, TCE_PK_CLOB232) TCE_PK_REQ_W(1, TCE_PK_CLOB232)
            TCE_PK_REQ_C(0, 0, TCE_PK_CLOB232) TCE_PK_REQ_C(0, 1, TCE_PK_CLOB232) TCE_PK_REQ_C(0, 2, TCE_PK_CLOB232) TCE_PK_REQ_C(0, 3, TCE_PK_CLOB232)
            TCE_PK_REQ_C(1, 0, TCE_PK_CLOB232) TCE_PK_REQ_C(1, 1, TCE_PK_CLOB232) TCE_PK_REQ_C(1, 2, TCE_PK_CLOB232) TCE_PK_REQ_C(1, 3, TCE_PK_CLOB232)
        }
    };
#define TCE_PK_GET(DST, R)                                                     \
    {                                                                          \
        unsigned v_;                                                           \
        asm volatile("v_mov_b32 %0, v%c1" : "=v"(v_) : "i"(ASMB + (R)));      \
        DST = v_;                                                              \
    }
#define TCE_PK_GET_C(J, GI)                                                    \
    if constexpr (GI < GPB) {                                                  \
        TCE_PK_GET(dst.c[J][GI].x, 4 * kNT + 2 * (J * GPB + GI))               \
        TCE_PK_GET(dst.c[J][GI].y, 4 * kNT + 2 * (J * GPB + GI) + 1)           \
    }
    auto collect_block = [&](BlockRegs &dst) {  // only behind the counted wait that retires the request
        if constexpr (!INFL) {
        TCE_PK_GET(dst.w[0].x, 0) TCE_PK_GET(dst.w[0].y, 1) TCE_PK_GET(dst.w[0].z, 2) TCE_PK_GET(dst.w[0].w, 3)
        TCE_PK_GET(dst.w[1].x, 4) TCE_PK_GET(dst.w[1].y, 5) TCE_PK_GET(dst.w[1].z, 6) TCE_PK_GET(dst.w[1].w, 7)
        TCE_PK_GET_C(0, 0) TCE_PK_GET_C(0, 1) TCE_PK_GET_C(0, 2) TCE_PK_GET_C(0, 3)
        TCE_PK_GET_C(1, 0) TCE_PK_GET_C(1, 1) TCE_PK_GET_C(1, 2) TCE_PK_GET_C(1, 3)
        }
    };
    constexpr int NWL = kNT * (1 + GPB);  // VMEM instructions of request_block
    // 256-row form: no reserved registers (the attribute that reserved them is only a hint, see kAsmVgprBase).  A block's words and constants are requested by asm
    // loads into ORDINARY variables at the START of the iteration that ends with their first use, and become usable through the asm statement that carries the counted
    // wait (it names them as in/out operands): between the two the compiler sees live values it has no reason to touch -- no loop back edge is crossed, so no phi copy --
    // and every real use depends on the wait.  (A spill of one of them in between would store stale data: the build fails on ANY spill in this kernel, build.py.)
    BlockRegs inflight;
    auto request_inflight = [&](int t_of_block) {
        const int kbn = block_of(t_of_block);
#pragma unroll
        for (int j = 0; j < kNT; ++j) {
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(inflight.w[j]) : "v"(w_voff), "s"(w_tile[j] + (size_t)kbn * 64) : "memory");
            if constexpr (WIDE) asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(inflight.c[j][0].x) : "v"(c_voff), "s"(c_tile[j] + (size_t)kbn * 16) : "memory");
            else asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(inflight.c[j][0]) : "v"(c_voff), "s"(c_tile[j] + (size_t)kbn * 16) : "memory");
        }
    };
    // the counted wait behind which `inflight` becomes valid (N = the newer VMEM instructions that may stay in flight); it names every landing register as in / out
    static_assert(!INFL || GPB == 1, "the wait statement names one group per k-block");
#define TCE_PK_WAIT_INFLIGHT(N)                                                                                                                                          \
    if constexpr (kNT == 3) {                                                                                                                                            \
        asm volatile("s_waitcnt vmcnt(%c6)"                                                                                                                              \
                     : "+v"(inflight.w[0]), "+v"(inflight.w[1]), "+v"(inflight.w[2 % kNT]), "+v"(inflight.c[0][0].x), "+v"(inflight.c[1][0].x), "+v"(inflight.c[2 % kNT][0].x) \
                     : "n"(N)                                                                                                                                            \
                     : "memory");                                                                                                                                        \
    } else if constexpr (kNT == 4) {                                                                                                                                     \
        asm volatile("s_waitcnt vmcnt(%c8)"                                                                                                                              \
                     : "+v"(inflight.w[0]), "+v"(inflight.w[1]), "+v"(inflight.w[2 % kNT]), "+v"(inflight.w[3 % kNT]), "+v"(inflight.c[0][0].x), "+v"(inflight.c[1][0].x), \
                       "+v"(inflight.c[2 % kNT][0].x), "+v"(inflight.c[3 % kNT][0].x)                                                                                      \
                     : "n"(N)                                                                                                                                            \
                     : "memory");                                                                                                                                        \
    } else {                                                                                                                                                             \
        asm volatile("s_waitcnt vmcnt(%c4)" : "+v"(inflight.w[0]), "+v"(inflight.w[1]), "+v"(inflight.c[0][0]), "+v"(inflight.c[1][0]) : "n"(N) : "memory");             \
    }
    if constexpr (RD == 2) {
        pk_wait_vmcnt<DPW>();  // the first block's even half has landed (its odd half may still be in flight: awaited in front of the loop's first barrier)
    } else if constexpr (INFL) {
        pk_wait_vmcnt<2 * DPW>();
    } else {
        request_block(1);
        pk_wait_vmcnt<2 * DPW + NWL>();  // the first block's two half-stages have landed (this wave's part); the barrier makes that the workgroup's
    }
    if constexpr (!(AB & 32)) __builtin_amdgcn_s_barrier();
    // fragments of step 0 of the first block
    read_a(af[0], ring, 0);
    if constexpr (WIDE) {
#pragma unroll
        for (int j = 0; j < kNT; ++j) unpack_col(bf[0][j], cur, 0, j);
    } else {
        set_group(cur, 0);
        unpack(bf[0], cur, 0);
    }
    load_e(cur);

    if constexpr (RD == 2) {
        // ---- two half-stage slots per quartet (256-row tile, two quartets alternating k-blocks) ----
        // slot 0 = the even half of the quartet's current block (steps 0, 1), slot 1 = its odd half (steps 2, 3).  Per iteration, in VMEM issue order:
        //   [odd(t): issued behind barrier B2 of the previous iteration]  [words(t+1)]  B1  [even(t+1) -> slot 0]  B2  [odd(t+1) -> slot 1]
        // B1 sits behind region 0 (the last reads of even(t)); the wait in front of it, vmcnt(NWL), retires odd(t) -- read from region 1 on, i.e. one barrier later.
        // B2 sits behind region 2 (the last reads of odd(t)); the wait in front of it, vmcnt(0), retires words(t+1) and even(t+1) -- read by region 3, behind B2.
        // Every read of a staged half is thus separated from the wait that retires it by a barrier every wave of the workgroup has passed, and every refill from the
        // last read of the slot by a barrier behind that wave's lgkmcnt(0).
        static_assert(RD != 2 || (kNT == 2 && GPB == 1), "the wait statement below names two column tiles, one group per k-block");
        const unsigned char *st_even = ring, *st_odd = ring + HALF_BYTES;
        for (int t = 0; t < T; ++t) {
            constexpr bool live = true;  // (this form is offered for an even number of k-blocks only: both quartets run every iteration -- a conditional region costs
                                         //  the loop 48 spilled registers, and a spill inside the window of the in-flight asm loads would store stale data)
            request_inflight(t + 1);
            if (live) region(std::integral_constant<int, 0>{}, cur, st_even);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pk_wait_vmcnt<NWL>();
            if constexpr (!(AB & 32)) __builtin_amdgcn_s_barrier();
            issue_half(2 * t + 2);
            if (live) {
                region(std::integral_constant<int, 1>{}, cur, st_odd);
                region(std::integral_constant<int, 2>{}, cur, st_odd);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TCE_PK_WAIT_INFLIGHT(0)
            cur = inflight;
            if constexpr (!(AB & 32)) __builtin_amdgcn_s_barrier();
            issue_half(2 * t + 3);
            if (live) region(std::integral_constant<int, 3>{}, cur, st_even);
            load_e(cur);
        }
    } else
    for (int t = 0; t < T; ++t) {
        // wave-uniform; only quartet 1's last iteration can be past the run (the wide form with two quartets is offered for an even number of k-blocks only: a conditional
        // region inside the window of the in-flight asm loads invites spills -- see the two-slot loop above)
        const bool live = (WIDE && KS == 2) ? true : grp + t * KS < nloc;
        const unsigned char *st_even = ring + ((2 * t) & 3) * HALF_BYTES, *st_odd = ring + ((2 * t + 1) & 3) * HALF_BYTES;
        const unsigned char *st_even_next = ring + ((2 * t + 2) & 3) * HALF_BYTES;
        if constexpr (ROLL) request_inflight(t + 1);
        if (live) {
            region(std::integral_constant<int, 0>{}, cur, st_even);  // MFMAs of step 0 | fragments of step 1 (even half-stage)
            // (wide form: the request goes out behind region 0 -- with the rescale ratios of region 0 dead and a quarter of the block's words unpacked the landing
            //  registers fit; two regions of MFMAs remain to cover the loads)
            if constexpr (WIDE) request_inflight(t + 1);
            region(std::integral_constant<int, 1>{}, cur, st_odd);   // step 1 | step 2 (odd half-stage)
        }
        // all waves have taken their last fragment of the even half-stage: it may be refilled (own half-block 2t+4)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!(AB & 32)) __builtin_amdgcn_s_barrier();
        if constexpr (!WIDE) issue_half(2 * t + 4);
        if (live) region(std::integral_constant<int, 2>{}, cur, st_odd, 2 * t + 4);  // step 2 | step 3 (wide form: with the refill of the even half-stage dealt out between its groups)
        // End of the block's LDS reads.  VMEM order since the even half-stage of the NEXT block was requested: [its odd half-stage]
        // [its words / constants] [half 2t+4].  vmcnt(DPW) leaves only the DMAs of half 2t+4 in flight: both half-stages of
        // block t+1 and its words have landed (in-order counter); the barrier extends that to the workgroup and orders this
        // block's reads of the odd half-stage before its refill.
        if constexpr (INFL) {
            TCE_PK_WAIT_INFLIGHT(DPW)
            cur = inflight;  // `cur` (block t) is used up: step 3's fragments are unpacked
        } else {
            pk_wait_vmcnt<DPW>();
            collect_block(cur);  // `cur` (block t) is used up: step 3's fragments are unpacked
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!(AB & 32)) __builtin_amdgcn_s_barrier();
        if constexpr (!WIDE) issue_half(2 * t + 5);
        if constexpr (!INFL) request_block(t + 2);
        if (live) region(std::integral_constant<int, 3>{}, cur, st_even_next, 2 * t + 5);  // step 3 | step 0 of the next block (`cur` is the next block now)
        load_e(cur);
    }
#undef TCE_PK_CLOB232
#undef TCE_PK_WAIT_INFLIGHT
#undef TCE_PK_REQ_W
#undef TCE_PK_REQ_C
#undef TCE_PK_GET
#undef TCE_PK_GET_C
    pk_wait_vmcnt<0>();
    __syncthreads();  // every wave's last (clamped, redundant) DMAs have landed: the ring may be overwritten

    // ---- into true units: e of the quartet's last group ----
    // Round 6, the lost-lanes defect (profiles/r6/pk_lost_lanes_rule.md): hipcc packs these multiplies two by two (v_pk_mul_f32) and, with e_prev[0..] living in 64-bit
    // register pairs, broadcast the ODD register of a pair through `op_sel:[0,1]` -- the low product takes the second source's HIGH register.  On gfx950 that form's low
    // product came back as 0.0 in lanes 48-63 once in ~10^2-10^3 launches (always behind a long enough idle stretch of the wave's vector ALU): the accumulator register of
    // one column tile lost its whole history.  The scale of each column tile is therefore made an independent 32-bit value here (the empty statement): the packed multiply
    // then reads it as the LOW register of its pair (op_sel_hi:[1,0], the form the loop's rescale has always had and that never failed), and build.py's ISA lint
    // (isa_lint.py) refuses any object that still holds a packed-f32 instruction with a low-half op_sel.
    float e_last[kNT];
#pragma unroll
    for (int j = 0; j < kNT; ++j) {
        e_last[j] = e_prev[j];
        asm volatile("" : "+v"(e_last[j]));
    }
#pragma unroll
    for (int j = 0; j < kNT; ++j)
#pragma unroll
        for (int i = 0; i < kMT; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc[i][j][rr] *= e_last[j];
    // (the products stand in binary32 registers before anything converts them: in the 128 x 512 form hipcc folded this multiply into the epilogue's conversion --
    //  v_fma_mixlo_f16, ONE rounding from the exact product to binary16 instead of two -- and 0.007 % of the outputs differed from the other forms' by an ulp;
    //  every form of this kernel rounds the same way; tests/test_gpu_w4a16_pk.py holds every wide form to the 128 x 128 form's bits)
    if constexpr (WIDE && NS == 2) {
#pragma unroll
        for (int i = 0; i < kMT; ++i)
#pragma unroll
            for (int j = 0; j < kNT; ++j) asm volatile("" : "+v"(acc[i][j]));
    }

    if constexpr (KS == 2) {  // quartet 1 hands its partial sums over: [register][thread of the quartet] floats (64 KiB)
        float *red = reinterpret_cast<float *>(smem);
        const int t4 = tid & 255;
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < kMT; ++i)
#pragma unroll
                for (int j = 0; j < kNT; ++j)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) red[((i * kNT + j) * 4 + rr) * 256 + t4] = acc[i][j][rr];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < kMT; ++i)
#pragma unroll
                for (int j = 0; j < kNT; ++j)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) acc[i][j][rr] += red[((i * kNT + j) * 4 + rr) * 256 + t4];
        }
        __syncthreads();
    }

    if (CUT && split > 1) {
        // (KS == 2: the joined tile sits in quartet 0's registers -- `grp == 0` moves the partials; everything that synchronises or decides is done by all 512 threads)
        // ---- K split across workgroups: the run's partial tile (true units, fp32) goes to the scratch area with write-through stores;
        //      the workgroup that arrives LAST at the tile's counter adds the split_s partials in run order -- so the sum does not
        //      depend on who was last -- and stores the tile.  (Same visibility
        //      rules as the fast attention step's chunk merge: acknowledged device-scope stores, then the counter, then coherent loads.)
        const int tile_lin = bid - 8 * g.full_slots;  // index among the cut tiles (holes of the grid included)
        const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(g.partials + (size_t)tile_lin * split * (kMT * kNT * 256), 0,
                                                                               split * kMT * kNT * 256 * 16, 0x00020000);
        if (handoff && part == 1) {
            // run 1 of a hand-off: no partial of its own leaves the registers.  Wait for run 0's (bounded: ~0.3 s of polls -- the launch then ends with this run's part
            // alone rather than hanging a queue; it has never been seen to happen), take the counter back to zero for the next launch
            unsigned *flag = reinterpret_cast<unsigned *>(smem);
            __syncthreads();
            // Round 6: a wait that runs out must be LOUD and must not reach into later launches.  The counter word is 0 (idle), 1 (run 0's tile is there) or
            // kPkPoison: a run 1 whose wait runs out poisons the word (compare-and-swap 0 -> poison; if run 0 published in that instant the swap fails and all is well),
            // counts the fault and stores NaN for the whole tile; run 0 publishes with compare-and-swap 0 -> 1, so a late run 0 cannot un-poison the word; every later
            // launch that meets the poisoned word stores NaN and counts a fault as well -- until the host clears the scratch area's first 4096 bytes
            // (tce_w4a16_gemm_scratch_faults reports the count).  Round 5 stored the tile with this run's k range alone, returned TCE_OK and zeroed the word, which the late
            // run 0 then left at 1 for the next launch to trip over.
            // (Visibility: run 0 stores its partial tile write-through (sc0 sc1), waits for the acknowledgement and only then publishes; the reads below are sc0 sc1 loads
            //  of those lines -- the pairing MI355X_MICROARCH.md lists as valid without a separate acquire.)
            if (tid == 0) {
                unsigned v = 0;
                for (int n = 0; n < (1 << 20) && !v; ++n) {
                    v = __hip_atomic_load(g.counters + tile_lin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!v) __builtin_amdgcn_s_sleep(2);
                }
                if (v == 0u) {  // out of patience
                    unsigned expect = 0u;
                    if (!__hip_atomic_compare_exchange_strong(g.counters + tile_lin, &expect, kPkPoison, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) v = expect;
                }
                if (v == 1u) __hip_atomic_store(g.counters + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                else __hip_atomic_fetch_add(g.counters + kPkFaultWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = v == 1u ? 1u : 2u;
            }
            __syncthreads();
            const bool have = *flag == 1u;
            __syncthreads();  // (the flag word is part of the output tile's LDS image below)
            if (!have) {
#pragma unroll
                for (int i = 0; i < kMT; ++i)
#pragma unroll
                    for (int j = 0; j < kNT; ++j) acc[i][j] = float4_t{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
            }
            if (have && grp == 0) {  // run order: run 0's partial first, then this run's (the same sum the last-arriver form computes)
                uint4_t t4[kMT * kNT];
#pragma unroll
                for (int r = 0; r < kMT * kNT; ++r) t4[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, (r * 256 + tid) * 16, 0, /*sc0|sc1*/ 17);
#pragma unroll
                for (int i = 0; i < kMT; ++i)
#pragma unroll
                    for (int j = 0; j < kNT; ++j) {
                        const float4_t own = acc[i][j];
                        acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
                        acc[i][j] += __builtin_bit_cast(float4_t, t4[i * kNT + j]);
                        acc[i][j] += own;
                    }
            }
        } else {
        if (grp == 0) {
#pragma unroll
        for (int i = 0; i < kMT; ++i)
#pragma unroll
            for (int j = 0; j < kNT; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, acc[i][j]), rs_p, ((part * kMT * kNT + i * kNT + j) * 256 + tid) * 16, 0, /*sc0|sc1*/ 17);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned *flag = reinterpret_cast<unsigned *>(smem);
        __syncthreads();
        if (handoff) {  // run 0 of a hand-off: its tile is acknowledged; raise the counter (0 -> 1 only: a poisoned word stays poisoned) and leave
            if (tid == 0) {
                unsigned expect = 0u;
                (void)__hip_atomic_compare_exchange_strong(g.counters + tile_lin, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(g.counters + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned last = old == (unsigned)split - 1 ? 1u : 0u;
            if (last) __hip_atomic_store(g.counters + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            if (old >= (unsigned)split) {  // a word an earlier launch's fault left poisoned: nobody would ever be last -- run 0 stores NaN for the tile, the fault is counted
                __hip_atomic_fetch_add(g.counters + kPkFaultWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = part == 0 ? 2u : 0u;
            }
            *flag = last;
        }
        __syncthreads();
        const unsigned role = *flag;
        if (role == 0u) return;
        __syncthreads();  // (the flag word is part of the output tile's LDS image below)
        if (role == 2u) {
#pragma unroll
            for (int i = 0; i < kMT; ++i)
#pragma unroll
                for (int j = 0; j < kNT; ++j) acc[i][j] = float4_t{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        } else
        // the sum in run order 0, 1, ..., split_s - 1 whoever computes it
        if constexpr (kMT * kNT > 16) {
            // 256-row and wide tiles: 128 accumulator registers leave no room for a second copy -- every run's partial (this one's included: the same values it stored) is
            // read back, eight registers at a time
#pragma unroll
            for (int i = 0; i < kMT; ++i)
#pragma unroll
                for (int j = 0; j < kNT; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < split && grp == 0; ++p) {
#pragma unroll
                for (int c0 = 0; c0 < kMT * kNT; c0 += 8) {
                    uint4_t t4[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) t4[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, ((p * kMT * kNT + c0 + r) * 256 + tid) * 16, 0, /*sc0|sc1*/ 17);
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[(c0 + r) / kNT][(c0 + r) % kNT] += __builtin_bit_cast(float4_t, t4[r]);
                }
            }
        } else {
        // (128-row tiles) the other runs' partials from memory, its own from its registers (the same values it stored)
        float4_t own[kMT][kNT];
#pragma unroll
        for (int i = 0; i < kMT; ++i)
#pragma unroll
            for (int j = 0; j < kNT; ++j) {
                own[i][j] = acc[i][j];
                acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
            }
        for (int p = 0; p < split && grp == 0; ++p) {
            if (p == part) {  // workgroup-uniform
#pragma unroll
                for (int i = 0; i < kMT; ++i)
#pragma unroll
                    for (int j = 0; j < kNT; ++j) acc[i][j] += own[i][j];
                continue;
            }
            uint4_t t4[kMT * kNT];
#pragma unroll
            for (int r = 0; r < kMT * kNT; ++r) t4[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, ((p * kMT * kNT + r) * 256 + tid) * 16, 0, /*sc0|sc1*/ 17);
#pragma unroll
            for (int i = 0; i < kMT; ++i)
#pragma unroll
                for (int j = 0; j < kNT; ++j) acc[i][j] += __builtin_bit_cast(float4_t, t4[i * kNT + j]);
        }
        }
        }  // (not run 1 of a hand-off)
    }

    // ---- epilogue: the tile leaves through LDS as 16-byte row pieces (accumulator layout: lane = column n16, registers = 4 consecutive rows) ----
    half_t *lds_c = reinterpret_cast<half_t *>(smem);  // [ROWS][BN]
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < kMT; ++i)
#pragma unroll
            for (int j = 0; j < kNT; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) lds_c[(i * 16 + q * 4 + rr) * BN + (wave * kNT + j) * 16 + n16] = (half_t)acc[i][j][rr];
    }
    __syncthreads();
    const bool vec_ok = (g.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0;
    constexpr int PPR = BN / 8;  // 16-byte pieces per row
    for (int e = tid; e < ROWS * PPR; e += NTHREADS) {
        const int row = e / PPR, pc = e % PPR;
        const int m = m_base + row, n = nb0 + pc * 8;
        if (m >= g.M || n >= g.N) continue;
        const half8_t v = *reinterpret_cast<const half8_t *>(lds_c + row * BN + pc * 8);
        if (g.pairs) {  // interleaved gate / up columns: the fp16 values the two linears would have stored, then SiLuMul_half (Int4llamaDecoderLayer.cu:20-30, 96-102)
            half_t *c2 = g.C + (size_t)m * g.ldc + (n >> 1);
            if (n + 8 <= g.N && (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 7) == 0) {  // one 8-byte store for the piece's four outputs (round 6: four 2-byte stores until now)
                *reinterpret_cast<half4_t *>(c2) = half4_t{silu_mul_half(v[0], v[1]), silu_mul_half(v[2], v[3]), silu_mul_half(v[4], v[5]), silu_mul_half(v[6], v[7])};
                continue;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (n + 2 * u + 1 < g.N) c2[u] = silu_mul_half(v[2 * u], v[2 * u + 1]);
            continue;
        }
        half_t *c = g.C + (size_t)m * g.ldc + n;
        if (vec_ok && n + 8 <= g.N) {
            half8_t o = v;
            if (g.add_to_c) {
                const half8_t old = *reinterpret_cast<const half8_t *>(c);
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = (half_t)(old[u] + v[u]);
            }
            *reinterpret_cast<half8_t *>(c) = o;
        } else {
            for (int u = 0; u < 8 && n + u < g.N; ++u) c[u] = g.add_to_c ? (half_t)(c[u] + v[u]) : v[u];
        }
    }
}

template <int KS, int LG, int ABL = 0, int NS = 1>
__global__ __launch_bounds__(256 * KS * NS, 2) __attribute__((amdgpu_num_vgpr(232))) void w4a16_gemm_pk_kernel(const PkGemmArgs g) {
    w4a16_gemm_pk_body<KS, LG, ABL, NS, 8>(g);
}
// 256-row wave tiles (round 5): ONE wave per SIMD (the 128 KiB ring admits one workgroup per CU anyway), so the wave may use the whole register file -- v0..v231 for
// the compiler (v232..v255 stay the asm loads'), AGPRs for what does not fit (the second set of A fragments)
template <int LG, int ABL = 0>
__global__ __launch_bounds__(256, 2) void w4a16_gemm_pk256_kernel(const PkGemmArgs g) {
    w4a16_gemm_pk_body<1, LG, ABL, 1, 16>(g);
}

// wide form (round 5): 128 rows x 64 columns per wave.  One quartet per 128 x 256 tile, two workgroups per CU (64 KiB ring each) ...
template <int LG, int ABL = 0>
__global__ __launch_bounds__(256, 2) void w4a16_gemm_pkw_kernel(const PkGemmArgs g) {
    w4a16_gemm_pk_body<1, LG, ABL, 1, 8, 4>(g);
}
// ... or two quartets alternating the k-blocks of ONE 128 x 256 tile (launches with at most one tile per CU: both SIMD slots of a CU still carry a wave)
template <int LG, int ABL = 0>
__global__ __launch_bounds__(512, 2) void w4a16_gemm_pkwx2_kernel(const PkGemmArgs g) {
    w4a16_gemm_pk_body<2, LG, ABL, 1, 8, 4>(g);
}

// 128 x 512 tiles: two quartets side by side on ONE activation ring (half the activation DMA instructions per MFMA again); one workgroup per CU
template <int LG, int ABL = 0>
__global__ __launch_bounds__(512, 2) void w4a16_gemm_pkw512_kernel(const PkGemmArgs g) {
    w4a16_gemm_pk_body<1, LG, ABL, 2, 8, 4>(g);
}
// the same with 48 columns per wave (128 x 192 tiles)
template <int LG, int ABL = 0>
__global__ __launch_bounds__(256, 2) void w4a16_gemm_pkw3_kernel(const PkGemmArgs g) {
    w4a16_gemm_pk_body<1, LG, ABL, 1, 8, 3>(g);
}
template <int LG, int ABL = 0>
__global__ __launch_bounds__(512, 2) void w4a16_gemm_pkw3x2_kernel(const PkGemmArgs g) {
    w4a16_gemm_pk_body<2, LG, ABL, 1, 8, 3>(g);
}

thread_local int g_pk_ks = 0;  // 0: choose per launch, 1 / 2: forced (tuning)
thread_local int g_pk_xm = 0;
thread_local int g_pk_abl = 0;  // timing experiments: parts of the loop switched off (one quartet, groups of 128 only)
thread_local int g_pk_split_force = 0;  // tuning: the number of runs a cut tile's k range is divided into (0: the cost model's choice)
thread_local int g_pk256_auto = 1;      // 0: the dispatcher never picks the 256-row forms by itself (A/B runs: tce_w4a16_set_debug_mode(650 / 651))
constexpr float kPk256wUsPerKBlock = 4.3f;  // eight waves of a 256 x 256 tile walking one k-block
constexpr float kPk256x2UsPerPair = 3.45f;   // two quartets sharing a CU walking one 256-row k-block each
thread_local int g_pk_handoff_delta = 2;   // k-blocks run 0 of a hand-off is shorter than run 1 (tce_w4a16_set_debug_mode(6950 + d))
thread_local int g_pk_prio = -1;  // -1: the rule below (the wide form with one quartet per tile); 0 .. 4 forced (tce_w4a16_set_debug_mode(696 / 697 / 6972 .. 6974; 698: the rule))
thread_local int g_pk_handoff = 1;         // 1: a k range cut in two runs is a directed hand-off (tce_w4a16_set_debug_mode(694): the last-arriver exchange, for the A/B)
constexpr float kPkHandoffUs = 5.0f;       // run 1's read of run 0's tile + what is left of run 0's write-through when run 1 arrives; fitted with the weights coming from HBM (scripts/probes/gemm_pk_handoff_cold_ab.py: 512 x 4096 x 4096 26.15 -> 24.85 us; at K = 11008 two handed-off runs take 54.55, four runs through the last arriver 53.5 -- the model keeps four there)
thread_local int g_pk_form16_auto = 1;                   // 1: the dispatcher may pick form 16 by itself (tce_w4a16_set_debug_mode(6916): never)
// fitted to profiles/r6/gemm_pk_form16_ab.jsonl (same process, weights in rotation): 512 x 4096 x 4096 27.3 -> 26.1 us, x 11008 53.1-54.1 -> 50.9-53.8, 384 x 4096 x 4096 24.8-25.5 -> 24.0-24.5;
// ties or losses where the model keeps the other forms (x 14336: four runs 64.7 against 63.6-67.6; 512 x 2048 x 8192, 64 tiles: four runs 29.4 against 36.5)
constexpr float kPk16ExtraUs = -2.0f;
thread_local int g_pk_wide_auto = 1;                    // 1: the dispatcher may pick the wide forms 10 / 11 / 12 by itself (tce_w4a16_set_debug_mode(692): never)
// fitted to profiles/r5/gemm_pkw_sweep.jsonl: 2048 x 4096 x 4096 (256 tiles, one per CU) 68.1 us; 4096 x 4096 x 4096 (512 tiles, two per CU) 112.8 us; two quartets on one tile 62.7 / 147.0 us at K = 4096 / 11008
constexpr float kPkWideAloneUs = 2.02f;    // wide form: one quartet alone on its CU walking a k-block (128 MFMAs per wave)
constexpr float kPkWidePairUs = 3.42f;     // wide form: two workgroups sharing a CU, one k-block each
constexpr float kPkWide512Us = 3.2f;       // 128 x 512 tile (eight waves, one ring) walking one k-block: 4096 x 4096 x 4096 (256 tiles) 109.0 us, 4096 x 4096 x 11008 270.4, 8192 x 4096 x 4096 210.9
constexpr float kPkWideX2PairUs = 3.45f;   // wide form: two quartets of ONE workgroup alternating a tile's k-blocks, per pair of k-blocks
constexpr float kPk256UsPerKBlock = 2.25f;  // one workgroup per CU walking a 256-row k-block (128 MFMAs per wave); fitted in round 5 (profiles/r5/gemm_pk256_sweep.jsonl)

template <int KS, int LG, int ABL = 0, int NS = 1>
hipError_t launch_pk(PkGemmArgs &g, hipStream_t stream) {
    size_t lds = (size_t)KS * 4 * pk::kHalfBytes;  // the rings; the quartet exchange (64 KiB) and the output tile (32 / 64 KiB) reuse them
    auto kfn = w4a16_gemm_pk_kernel<KS, LG, ABL, NS>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int per = 8 * g.m_per * g.n_per;
    hipLaunchKernelGGL(kfn, dim3(g.split_s > 1 ? per + (per - 8 * g.full_slots) * (g.split_s - 1) : per), dim3(256 * KS * NS), lds, stream, g);
    return hipGetLastError();
}

// 256-row tile shared by TWO quartets that alternate its k-blocks (512 threads, two waves per SIMD, 2 x 64 KiB rings, partial tiles joined through LDS)
template <int LG, int ABL = 0>
__global__ __launch_bounds__(512, 2) void w4a16_gemm_pk256x2_kernel(const PkGemmArgs g) {
    w4a16_gemm_pk_body<2, LG, ABL, 1, 16>(g);
}

// 256 x 256 tile: two quartets side by side (256 rows x 32 columns per wave) on ONE activation ring of four half-stages -- half the DMA instructions and LDS writes per MFMA
template <int LG, int ABL = 0>
__global__ __launch_bounds__(512, 2) void w4a16_gemm_pk256w_kernel(const PkGemmArgs g) {
    w4a16_gemm_pk_body<1, LG, ABL, 2, 16>(g);
}
template <int LG, int ABL = 0>
hipError_t launch_pk256w(PkGemmArgs &g, hipStream_t stream) {
    const size_t lds = (size_t)4 * 256 * pk::kHalfK * 2;  // 128 KiB: the ring; the 256 x 256 output tile (128 KiB) reuses it
    auto kfn = w4a16_gemm_pk256w_kernel<LG, ABL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(8 * g.m_per * g.n_per), dim3(512), lds, stream, g);
    return hipGetLastError();
}

template <int LG, int ABL = 0>
hipError_t launch_pk256x2(PkGemmArgs &g, hipStream_t stream) {
    const size_t lds = (size_t)2 * 2 * 256 * pk::kHalfK * 2;  // two rings of two half-stages of 256 rows x 64 k: 128 KiB (the quartet exchange, 128 KiB, and the output tile reuse them)
    auto kfn = w4a16_gemm_pk256x2_kernel<LG, ABL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(8 * g.m_per * g.n_per), dim3(512), lds, stream, g);
    return hipGetLastError();
}

template <int LG, int ABL = 0>
hipError_t launch_pkw(PkGemmArgs &g, hipStream_t stream) {
    const size_t lds = (size_t)4 * pk::kHalfBytes;  // ring of four half-stages of 128 rows x 64 k: 64 KiB (the 128 x 256 output tile, 64 KiB, reuses it)
    auto kfn = w4a16_gemm_pkw_kernel<LG, ABL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int per = 8 * g.m_per * g.n_per;
    hipLaunchKernelGGL(kfn, dim3(g.split_s > 1 ? per + (per - 8 * g.full_slots) * (g.split_s - 1) : per), dim3(256), lds, stream, g);
    return hipGetLastError();
}
template <int LG, int ABL = 0>
hipError_t launch_pkwx2(PkGemmArgs &g, hipStream_t stream) {
    const size_t lds = (size_t)2 * 4 * pk::kHalfBytes;  // two rings: 128 KiB (the quartet exchange, 128 KiB, and the output tile reuse them)
    auto kfn = w4a16_gemm_pkwx2_kernel<LG, ABL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(8 * g.m_per * g.n_per), dim3(512), lds, stream, g);
    return hipGetLastError();
}

hipError_t launch_pkw512(PkGemmArgs &g, hipStream_t stream) {
    const size_t lds = (size_t)128 * 512 * 2;  // the 128 x 512 output tile (128 KiB); the ring (64 KiB) sits inside it
    auto kfn = w4a16_gemm_pkw512_kernel<7, 0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(8 * g.m_per * g.n_per), dim3(512), lds, stream, g);
    return hipGetLastError();
}

template <int KS>
hipError_t launch_pkw3(PkGemmArgs &g, hipStream_t stream) {
    const size_t lds = (size_t)KS * 4 * pk::kHalfBytes;
    hipError_t e;
    if constexpr (KS == 2) {
        auto kfn = w4a16_gemm_pkw3x2_kernel<7, 0>;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(8 * g.m_per * g.n_per), dim3(512), lds, stream, g);
    } else {
        auto kfn = w4a16_gemm_pkw3_kernel<7, 0>;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(8 * g.m_per * g.n_per), dim3(256), lds, stream, g);
    }
    return hipGetLastError();
}

template <int LG, int ABL = 0>
hipError_t launch_pk256(PkGemmArgs &g, hipStream_t stream) {
    const size_t lds = (size_t)4 * 256 * pk::kHalfK * 2;  // ring of four half-stages of 256 rows x 64 k: 128 KiB (the output tile, 64 KiB, reuses it)
    auto kfn = w4a16_gemm_pk256_kernel<LG, ABL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int per = 8 * g.m_per * g.n_per;
    hipLaunchKernelGGL(kfn, dim3(g.split_s > 1 ? per + (per - 8 * g.full_slots) * (g.split_s - 1) : per), dim3(256), lds, stream, g);
    return hipGetLastError();
}

}  // namespace

#ifdef TCE_LAB
void set_gemm_pk_ablation(int abl) { g_pk_abl = abl; }
#else
void set_gemm_pk_ablation(int) { g_pk_abl = 0; }  // (the loop-parts-switched-off instantiations exist in the lab build only: build.py --lab)
#endif
void set_gemm_pk256_auto(int on) { g_pk256_auto = on ? 1 : 0; }
void set_gemm_pk_wide_auto(int on) { g_pk_wide_auto = on ? 1 : 0; }
void set_gemm_pk_form16_auto(int on) { g_pk_form16_auto = on ? 1 : 0; }
void set_gemm_pk_handoff(int on) { g_pk_handoff = on ? 1 : 0; }
void set_gemm_pk_prio(int on) { g_pk_prio = on >= 0 && on <= 4 ? on : -1; }
void set_gemm_pk_handoff_delta(int d) { g_pk_handoff_delta = d >= 0 && d <= 8 ? d : 2; }
void set_gemm_pk_split(int s) { g_pk_split_force = s >= 2 && s <= 4 ? s : 0; }

void set_gemm_pk_mode(int form, int xm) {
    g_pk_ks = (form >= 1 && form <= 16) ? form : 0;  // 16 (round 6): 128 x 128 tiles, two quartets alternating the k-blocks of a RUN, every tile's k range handed off between two workgroups  // 15: the wide form on 128 x 512 tiles (two quartets side by side on one activation ring)  // 13 / 14: the wide form on 128 x 192 tiles, one quartet per tile / two alternating its k-blocks  // 10 / 11 / 12: the wide form (one quartet per 128 x 256 tile; two quartets alternating its k-blocks; every tile's k range cut across workgroups)  // 9: 256 x 256 tiles, two quartets side by side (debug mode 2669)  // 6: 256-row wave tiles, whole tiles; 7: the same with every tile's k range cut across workgroups
    g_pk_xm = (xm == 1 || xm == 2 || xm == 4 || xm == 8) ? xm : 0;
}

size_t prepack_bytes(int N, int K, int G) {
    if (N <= 0 || K <= 0 || K % 128 != 0 || (G != 128 && G != 64 && G != 32)) return 0;
    return pk::total_bytes(N, K, G);
}

int launch_w4a16_prepack(const tce_w4a16_desc &d, void *out, hipStream_t stream, hipError_t *hip_err) {
    if (d.K % 128 != 0) return TCE_ERR_UNSUPPORTED_SHAPE;
    const int zw = zeros_width(d.K, d.group_size);
    PrepackArgs a{};
    a.qweight = static_cast<const unsigned *>(d.qweight);
    a.scales = static_cast<const half_t *>(d.scales);
    a.zeros = static_cast<const unsigned *>(d.zeros);
    unsigned char *base = static_cast<unsigned char *>(out);
    a.words = reinterpret_cast<unsigned *>(base);
    a.consts = reinterpret_cast<uint2_t *>(base + pk::consts_offset(d.N, d.K));
    a.last = reinterpret_cast<float *>(base + pk::last_offset(d.N, d.K, d.group_size));
    a.dscales = reinterpret_cast<half_t *>(base + pk::dscales_offset(d.N, d.K, d.group_size));
    a.dzeros = reinterpret_cast<unsigned *>(base + pk::dzeros_offset(d.N, d.K, d.group_size));
    a.N = d.N;
    a.K = d.K;
    a.log2g = d.group_size == 128 ? 7 : (d.group_size == 64 ? 6 : 5);
    a.scales_stride = d.scales_stride ? d.scales_stride : zw * 8;
    a.zeros_stride = d.zeros_stride ? d.zeros_stride : zw;
    const int nt = pk::nt16(d.N);
    hipLaunchKernelGGL(prepack_words_kernel, dim3(nt, d.K >> 7), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(prepack_consts_kernel, dim3((nt * 16 + 63) / 64), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(prepack_decode_kernel, dim3((unsigned)(((long long)nt * (d.K / d.group_size) + 63) / 64)), dim3(64), 0, stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

// Which form launch_w4a16_gemm_pk would run -- 1: 128x128 tiles, one quartet, two workgroups per CU; 2: 128x128, two quartets
// splitting K; 3: 128x256, two quartets side by side on one activation ring -- and a cost estimate in us for the dispatcher.
// A workgroup's time per k-block c grows with the load on its CU and on the chip (f = resident waves / 2048); fitted to
// profiles/r2/gemm_pk_sweep.jsonl: form 1 alone on its CU 1.06 us, sharing it 1.25 + 0.73 f; form 2 per PAIR of k-blocks
// 1.7 + 0.4 f; form 3 (every active CU carries eight waves) 1.5 + 0.4 f; + 3 us of launch, prologue and epilogue.
// Scratch for the K split across workgroups (form 4): [4 KiB of tile counters][kPkSplitMaxUnits partial tiles of 64 KiB] (32 MiB + 4 KiB; 288 units until round 4).
constexpr int kPkSplitMaxUnits = 512;

size_t gemm_pk_scratch_bytes() { return 4096 + (size_t)kPkSplitMaxUnits * 65536; }

// form 4 = form 1 with every tile's k-blocks cut into s runs on s workgroups (partial tiles added through the scratch area in a fixed
// order): for launches whose 128 x 128 tiles are too few for 256 CUs -- M = 512 at N = 4096 is 128 tiles.  One quartet alone on a CU
// walks a k-block in 1.06 us; the exchange costs what a dozen k-blocks cost, so the form pays from K ~ 6000 up (4096 x 11008 at
// M = 512: 69 -> 56 us) and is a tie at K = 4096.
static int pk_split_factor(long tiles1, int nkb, bool has_scratch, float *cost_out) {
    int best_s = 1;
    float best = 1e30f;
    if (has_scratch && tiles1 <= 1024)
        for (int s = 2; s <= 4; ++s) {
            if (tiles1 * s > kPkSplitMaxUnits || nkb / s < 4 || (g_pk_split_force && s != g_pk_split_force)) continue;
            // fitted to profiles/r2/gemm_pk_ksplit_sweep.jsonl: a run of k-blocks at the lone-quartet rate (units that have to share a CU:
            // two runs back to back) + 3 us of launch / prologue / epilogue + 5 + 1.3 s us for the exchange (64 KiB per unit written
            // through to memory -- 16 MB per launch --, the counter, the other partials read back)
            const float run = (float)((nkb + s - 1) / s);
            // (round 4, profiles/r4/gemm_pk_split_probe.jsonl: with room for 512 units in the scratch area, four runs per tile beat two on the long k ranges --
            //  512 x 4096 x 11008: 55.6 -> 52.0 us, x 14336: 70.7 -> 64.0 -- and lose on 4096 (26.9 -> 28.9); two units sharing a CU walk a k-block in ~1.8 us, + 1.5 us once)
            float c = (tiles1 * s <= 256 ? run * 1.06f : 1.8f * run + 1.5f) + 8.0f + 1.3f * (float)s;
            // (round 5) two runs as a directed hand-off: run 1 is a k-block longer than half, the exchange on its path is one read of run 0's tile
            if (s == 2 && g_pk_handoff && nkb >= 6) c = (tiles1 * s <= 256 ? (run + 1.f) * 1.06f : 1.8f * (run + 1.f) + 1.5f) + 3.0f + kPkHandoffUs;
            if (c < best) best = c, best_s = s;
        }
    if (cost_out) *cost_out = best;
    return best_s;
}

float gemm_pk_estimate_us(int M, int N, int K, int *form_out, bool has_scratch, int *split_out, int group_size, bool zero_point_8) {
    const long mt = (M + 127) / 128;
    const long tiles1 = mt * ((N + 127) / 128), tiles3 = mt * ((N + 255) / 256);
    const float nkb = (float)(K / 128);
    auto rounds = [](long tiles, long slots) {
        const float r = (float)tiles / (float)slots;
        return r <= 3.f ? (float)(int)(r + 0.999f) : r + 0.5f;
    };
    auto load = [](long waves) { return waves >= 2048 ? 1.0f : (float)waves / 2048.f; };
    // form 1 past one tile per CU: a CU walks its ceil(tiles / 256) tiles two at a time (688 tiles: three per CU = 1.5 pair-walks, measured
    // 93 us where "two rounds of 512" predicted 130 and sent 1024 x 11008 x 4096 to form 2 at 99; profiles/r2/gemm_pk_narrow_tile_sweep.jsonl)
    const float cost1 = (tiles1 <= 256 ? nkb * 1.06f : 0.5f * rounds(tiles1, 256) * nkb * (1.25f + 0.73f * load(tiles1 * 4))) + 3.0f;
    const float cost2 = rounds(tiles1, 256) * (nkb * 0.5f) * (1.7f + 0.4f * load(tiles1 * 8)) + 3.0f;
    const float cost3 = rounds(tiles3, 256) * nkb * (1.5f + 0.4f * load(tiles3 * 8)) + 3.0f;
    float cost4 = 1e30f;
    int split = pk_split_factor(tiles1, (int)nkb, has_scratch, &cost4);
    // form 5: 256 whole tiles, one per CU, and the k-blocks of the other tiles1 - 256 as s short runs beside them (each CU: a whole tile
    // and ~one run).  While a run shares the CU both quartets walk a k-block in ~1.9 us, afterwards the whole tile is alone again (1.06).
    float cost5 = 1e30f;
    int split5 = 1;
    if (has_scratch && tiles1 > 256 && tiles1 <= 256 + kPkSplitMaxUnits / 2) {
        const long tail = tiles1 - 256;
        for (int s5 = 2; s5 <= 4; ++s5) {
            // at most one run per CU: 344 tiles cut in 3 (264 runs, 520 workgroups for 512 slots) measured 60 us against 53 us cut in 2
            // and 56 us whole, while 320 and 336 tiles cut in 3 (192 / 240 runs) run 45-46 us (profiles/r2/gemm_pk_tail_cut_probe.jsonl)
            if ((tail + 7) / 8 * 8 * s5 > 256 || (int)nkb / s5 < 4 || (g_pk_split_force && s5 != g_pk_split_force)) continue;
            const float run = (float)(((int)nkb + s5 - 1) / s5);  // k-blocks during which a CU carries a run beside its whole tile
            const float c = run * 1.9f + (nkb - run) * 1.06f + 4.0f;
            if (c < cost5) cost5 = c, split5 = s5;
        }
    }
    int form = 1;
    float best = cost1;
    // Two quartets per workgroup (forms 2 / 3).  Round 5 took them away from groups of 64 / 32 after the groups-of-32 instantiation of form 2 returned, once in ~10^2-10^3
    // launches under load, one accumulator register's lanes 48-63 without the first quartet's history.  Round 6 found the cause -- ONE instruction form hipcc emitted in the
    // post-loop block (v_pk_mul_f32 with op_sel:[0,1]; isa_lint.py RULE 1, profiles/r6/pk_lost_lanes_rule.md, reproduced standalone) --, removed it from the sources and
    // made the build refuse it; the forms are offered for every group size again (profiles/r6: 0 of 1.7e5 launches of the fixed kernel under the pads that drove the old
    // one to 45-100 % failures).
    if (cost2 < best) best = cost2, form = 2;
    if (cost3 < best) best = cost3, form = 3;
    if (cost4 < best) best = cost4, form = 4;
    if (cost5 < best) best = cost5, form = 5;
    // forms 6 / 7 (round 5): 256-row wave tiles (one workgroup of four waves per CU on a 256 x 128 tile), whole tiles / every tile's k range cut into s runs.
    // A 256-row k-block carries twice the MFMAs of a 128-row one beside the same unpack: c6 us per k-block (fitted below)
    const long mt6 = (M + 255) / 256;
    const long tiles6 = mt6 * ((N + 127) / 128);
    float cost6 = 1e30f, cost7 = 1e30f;
    int split7 = 1;
    if (M > 128 && group_size == 128) {
        cost6 = rounds(tiles6, 256) * nkb * kPk256UsPerKBlock + 4.0f;
        if (has_scratch)
            for (int s7 = 2; s7 <= 4; ++s7) {
                if (tiles6 * s7 * 2 > kPkSplitMaxUnits || (int)nkb / s7 < 4 || (g_pk_split_force && s7 != g_pk_split_force)) continue;
                const float run = (float)(((int)nkb + s7 - 1) / s7);
                const float c = rounds(tiles6 * s7, 256) * run * kPk256UsPerKBlock + 9.0f + 2.0f * (float)s7;  // the exchange moves 128 KiB per unit
                if (c < cost7) cost7 = c, split7 = s7;
            }
    }
    // form 9 (round 5): 256 x 256 tiles, two quartets side by side on one ring
    float cost9 = 1e30f;
    const long tiles9 = mt6 * ((N + 255) / 256);
    if (M > 128 && group_size == 128) cost9 = rounds(tiles9, 256) * nkb * kPk256wUsPerKBlock + 5.0f;
    // form 8 (round 5): the 256 x 128 tile shared by two quartets that alternate its k-blocks (two waves per SIMD): kPk256x2UsPerPair per pair of k-blocks
    float cost8 = 1e30f;
    if (M > 128 && group_size == 128 && nkb >= 2.f && ((int)nkb & 1) == 0) cost8 = rounds(tiles6, 256) * (float)(((int)nkb + 1) / 2) * kPk256x2UsPerPair + 5.0f;
    // Only form 8 is offered to the dispatcher: forms 6 / 7 (one wave per SIMD: DMA issue, fragment reads and barriers serialise with the MFMA stream -- 73 against 64 us at
    // 2048 x 4096 x 4096) and form 9 (half the chip at that size, level at 4096 rows) lost every same-process comparison (profiles/r5/gemm_pk256_*.jsonl); they stay
    // reachable through tce_w4a16_set_debug_mode (66 / 67 / 672-674 / 2669) and in the parity tests.
    if (g_pk256_auto && cost8 < best) best = cost8, form = 8;
    // forms 10 / 11 / 12 (round 5): the wide form -- 128 x 256 tiles, 128 rows x 64 columns per wave.  10: one quartet per tile, two workgroups per CU; 11: two quartets
    // alternating the k-blocks of one tile (one workgroup per CU); 12: form 10 with every tile's k range cut into s runs on s workgroups
    const long tilesw = mt * ((N + 255) / 256);
    float cost10 = 1e30f, cost11 = 1e30f, cost12 = 1e30f;
    int split12 = 1;
    if (M > 128 && group_size == 128 && zero_point_8) {  // (the wide form reads no zero points: linears whose zero points are all 8 -- the flag the caller sets after tce_w4a16_check_zero_point_8)
        cost10 = (tilesw <= 256 ? nkb * kPkWideAloneUs : 0.5f * rounds(tilesw, 256) * nkb * kPkWidePairUs) + 3.5f;
        if (((int)nkb & 1) == 0 && nkb >= 2.f) cost11 = rounds(tilesw, 256) * (nkb * 0.5f) * kPkWideX2PairUs + 4.5f;
        if (has_scratch)
            for (int s = 2; s <= 4; ++s) {
                if (tilesw * s * 2 > kPkSplitMaxUnits || (int)nkb / s < 4 || (g_pk_split_force && s != g_pk_split_force)) continue;
                const float run = (float)(((int)nkb + s - 1) / s);
                const float c = (tilesw * s <= 256 ? run * kPkWideAloneUs : 0.5f * rounds(tilesw * s, 256) * run * kPkWidePairUs) + 9.0f + 2.0f * (float)s;  // the exchange moves 128 KiB per unit
                if (c < cost12) cost12 = c, split12 = s;
            }
    }
    // forms 13 / 14: 128 x 192 tiles (48 columns per wave): three quarters of a 256-wide tile's work per k-block
    const long tilesw3 = mt * ((N + 191) / 192);
    float cost13 = 1e30f, cost14 = 1e30f;
    if (M > 128 && group_size == 128 && zero_point_8) {
        cost13 = (tilesw3 <= 256 ? nkb * kPkWideAloneUs : 0.5f * rounds(tilesw3, 256) * nkb * kPkWidePairUs) * 0.78f + 3.5f;
        if (((int)nkb & 1) == 0 && nkb >= 2.f) cost14 = rounds(tilesw3, 256) * (nkb * 0.5f) * kPkWideX2PairUs * 0.78f + 4.5f;
    }
    // form 15: 128 x 512 tiles, eight waves on one ring, one workgroup per CU
    const long tilesw512 = mt * ((N + 511) / 512);
    float cost15 = 1e30f;
    if (M > 128 && group_size == 128 && zero_point_8) cost15 = rounds(tilesw512, 256) * nkb * kPkWide512Us + 5.0f;
    if (g_pk_wide_auto) {
        if (cost15 < best) best = cost15, form = 15;
        if (cost13 < best) best = cost13, form = 13;
        if (cost14 < best) best = cost14, form = 14;
        if (cost10 < best) best = cost10, form = 10;
        if (cost11 < best) best = cost11, form = 11;
        if (cost12 < best) best = cost12, form = 12;
    }
    // form 16 (round 6): form 2's tile -- two quartets alternating its k-blocks, two waves per SIMD -- with every tile's k range handed off between TWO workgroups: for the
    // launches whose 128 x 128 tiles are at most half the CUs (M = 512 at N = 4096: 128 tiles), where form 4 leaves one wave per SIMD and form 2 half the chip idle.
    // A run is ceil(nkb / 2) (+ 1: run 1 is the longer one) k-blocks, walked in pairs at form 2's rate
    float cost16 = 1e30f;
    if (has_scratch && tiles1 * 2 <= 256 && nkb >= 8.f && g_pk_handoff && group_size == 128)  // (measured with groups of 128 only)
        cost16 = (float)(((int)nkb / 2 + 1 + 1) / 2) * (1.7f + 0.4f * load(tiles1 * 2 * 8)) + 3.0f + kPkHandoffUs + kPk16ExtraUs;
    if (g_pk_form16_auto && cost16 < best) best = cost16, form = 16;
    if (g_pk_ks) {
        form = g_pk_ks;
        if (form == 16 && cost16 > 1e29f) form = 2;
        if (form == 4 && split == 1) form = split5 > 1 ? 5 : 1;  // "cut the k range": whichever of the two cut forms applies
        if (form == 7 && split7 == 1) form = 6;
        if ((form >= 6 && form <= 9) && (M <= 128 || group_size != 128)) form = 1;
        if (form == 8 && (nkb < 2.f || ((int)nkb & 1))) form = 6;
        if (form >= 10 && form <= 15 && (M <= 128 || group_size != 128 || !zero_point_8)) form = 1;
        if (form == 11 && (nkb < 2.f || ((int)nkb & 1))) form = 10;
        if (form == 12 && split12 == 1) form = 10;
        if (form == 14 && (nkb < 2.f || ((int)nkb & 1))) form = 13;
        if (form == 16) best = cost16;
        else if (form >= 10) best = form == 10 ? cost10 : (form == 11 ? cost11 : (form == 12 ? cost12 : (form == 13 ? cost13 : (form == 14 ? cost14 : cost15))));
        else
        best = form == 1 ? cost1 : (form == 2 ? cost2 : (form == 3 ? cost3 : (form == 4 ? cost4 : (form == 5 ? cost5 : (form == 6 ? cost6 : (form == 7 ? cost7 : (form == 8 ? cost8 : cost9)))))));
    }
    if (form_out) *form_out = form;
    if (split_out) *split_out = form == 4 ? split : (form == 5 ? split5 : (form == 7 ? split7 : (form == 12 ? split12 : (form == 16 ? 2 : 1))));
    return best;
}

int launch_w4a16_gemm_pk(const tce_w4a16_desc &d, const void *packed, hipStream_t stream, hipError_t *hip_err) {
    if (d.K % 128 != 0 || !packed) return TCE_ERR_UNSUPPORTED_SHAPE;
    const int lda = d.lda ? d.lda : d.K;
    if ((lda * 2) % 16 != 0 || (reinterpret_cast<uintptr_t>(d.A) & 15)) return TCE_ERR_UNSUPPORTED_SHAPE;  // 16-byte DMA pieces
    // the activation DMAs address rows as a 32-bit byte offset from a uniform base: a batch beyond 4 GiB of activations goes to the other GEMM kernel
    if ((unsigned long long)d.M * (unsigned long long)lda * 2ull >= (1ull << 32)) return TCE_ERR_UNSUPPORTED_SHAPE;
    PkGemmArgs g{};
    const unsigned char *base = static_cast<const unsigned char *>(packed);
    g.A = static_cast<const half_t *>(d.A);
    g.words = reinterpret_cast<const uint4_t *>(base);
    g.consts = reinterpret_cast<const uint2_t *>(base + pk::consts_offset(d.N, d.K));
    g.C = static_cast<half_t *>(d.C);
    g.M = d.M;
    g.N = d.N;
    g.K = d.K;
    g.lda = lda;
    g.pairs = (d.flags & TCE_W4_SILU_MUL_PAIRS) ? 1 : 0;
    g.ldc = d.ldc ? d.ldc : (g.pairs ? d.N / 2 : d.N);
    g.add_to_c = (d.flags & TCE_W4_ADD_TO_C) ? 1 : 0;
    int form = 1, split = 1;
    const bool has_scratch = d.scratch != nullptr && (reinterpret_cast<uintptr_t>(d.scratch) & 255) == 0;
    gemm_pk_estimate_us(d.M, d.N, d.K, &form, has_scratch, &split, d.group_size, (d.flags & TCE_W4_ZERO_POINT_IS_8) != 0);
    const bool cut_tail_only = form == 5;
    const bool rows256 = form >= 6 && form <= 9;
    const bool rows256x2 = form == 8, rows256w = form == 9;
    const bool two_quartets_cut = form == 16;
    const bool wide = form >= 10 && form <= 15, widex2 = form == 11, wide3 = form == 13 || form == 14, wide512 = form == 15;
    const bool handoff_form = (form == 4 || form == 16) && split == 2 && g_pk_handoff;
    // Two workgroups of the one-quartet wide form share a CU, a wave of each on every SIMD, with nothing tying their progress: half of them (a CU's first dispatch round) run at
    // s_setprio 2 for the whole kernel -- 2048 x 11008 x 4096 162 -> 156 us, 3072 rows 260.4 -> 255.9, 4096 x 14336 x 4096 384-391 -> 379 (profiles/r5/gemm_pk_prio_ab.jsonl: whichever half,
    // whichever level); nothing for the forms whose two quartets share barriers, nothing for the narrow forms.  Priorities do not touch the arithmetic.
    g.prio = g_pk_prio >= 0 ? g_pk_prio : ((form == 10 || form == 12) ? 1 : 0);
    if (form == 4 || form == 5 || form == 7 || form == 12 || form == 16) {
        form = two_quartets_cut ? 2 : 1;
        g.split_s = split;
        g.handoff = handoff_form ? 1 + g_pk_handoff_delta : 0;
        g.counters = static_cast<unsigned *>(d.scratch);
        g.partials = reinterpret_cast<float4_t *>(static_cast<unsigned char *>(d.scratch) + 4096);
    } else {
        g.split_s = 1;
    }
    const int bn = wide512 ? 512 : (wide3 ? 192 : ((form == 3 || form == 9 || wide) ? 256 : 128));
    g.n_blocks = (d.N + bn - 1) / bn;
    g.m_blocks = rows256 ? (d.M + 255) / 256 : (d.M + 127) / 128;
    int best_xm = 1;
    long best_grid = -1;
    for (int xm = 8; xm >= 1; xm >>= 1) {  // the XCD grid that wastes the fewest workgroup slots, larger xm on ties (w4a16_gemm_dma.hip)
        const long grid = 8L * ((g.m_blocks + xm - 1) / xm) * ((g.n_blocks + 8 / xm - 1) / (8 / xm));
        if (best_grid < 0 || grid < best_grid) {
            best_grid = grid;
            best_xm = xm;
        }
    }
    g.xm = g_pk_xm ? g_pk_xm : best_xm;
    g.m_per = (g.m_blocks + g.xm - 1) / g.xm;
    g.n_per = (g.n_blocks + 8 / g.xm - 1) / (8 / g.xm);
    if (g.split_s > 1) {
        g.full_slots = cut_tail_only ? 32 : 0;  // 32 slots x 8 XCDs = the first 256 workgroups
        const int cut_wgs = 8 * (g.m_per * g.n_per - g.full_slots);
        if (cut_wgs <= 0 || (long)cut_wgs * g.split_s * ((rows256 || wide) ? 2 : 1) > kPkSplitMaxUnits || cut_wgs > 1024) g.split_s = 1;  // does not fit the scratch area: whole tiles
    }
    const int ks = form;
    hipError_t e;
    const int lg = d.group_size == 128 ? 7 : (d.group_size == 64 ? 6 : 5);
#ifdef TCE_LAB
    if (wide && lg == 7 && g_pk_abl && !widex2 && !wide3 && !wide512) {  // timing experiments on the wide form (results meaningless)
        switch (g_pk_abl) {
#define TCE_ABL(X) case X: e = launch_pkw<7, X>(g, stream); break;
            TCE_ABL(1) TCE_ABL(2) TCE_ABL(4) TCE_ABL(8) TCE_ABL(16) TCE_ABL(32) TCE_ABL(7) TCE_ABL(55) TCE_ABL(128) TCE_ABL(256)
#undef TCE_ABL
            default: return TCE_ERR_BAD_ARG;
        }
        if (e != hipSuccess) {
            if (hip_err) *hip_err = e;
            return TCE_ERR_HIP;
        }
        return TCE_OK;
    }
#endif
#ifdef TCE_LAB
    if (g_pk_abl && lg == 7 && !rows256 && !wide) {
        switch (g_pk_abl) {
#define TCE_ABL(X) case X: e = launch_pk<1, 7, X>(g, stream); break;
            TCE_ABL(1) TCE_ABL(2) TCE_ABL(4) TCE_ABL(8) TCE_ABL(16) TCE_ABL(32) TCE_ABL(6) TCE_ABL(7) TCE_ABL(23) TCE_ABL(55) TCE_ABL(47) TCE_ABL(48) TCE_ABL(64)
#undef TCE_ABL
            default: return TCE_ERR_BAD_ARG;
        }
        if (e != hipSuccess) {
            if (hip_err) *hip_err = e;
            return TCE_ERR_HIP;
        }
        return TCE_OK;
    }
#endif
    if (wide512) e = launch_pkw512(g, stream);
    else if (wide3) e = form == 14 ? launch_pkw3<2>(g, stream) : launch_pkw3<1>(g, stream);
    else if (wide) e = widex2 ? launch_pkwx2<7>(g, stream) : launch_pkw<7>(g, stream);
    else if (rows256w) e = launch_pk256w<7>(g, stream);
    else if (rows256x2) e = launch_pk256x2<7>(g, stream);
#ifdef TCE_LAB
    else if (rows256 && g_pk_abl) {  // timing experiments on the 256-row form (results meaningless): tce_w4a16_set_debug_mode(66), then 600 + bits as for the 128-row form
        switch (g_pk_abl) {
#define TCE_ABL(X) case X: e = launch_pk256<7, X>(g, stream); break;
            TCE_ABL(1) TCE_ABL(2) TCE_ABL(4) TCE_ABL(8) TCE_ABL(16) TCE_ABL(32) TCE_ABL(7) TCE_ABL(55)
#undef TCE_ABL
            default: return TCE_ERR_BAD_ARG;
        }
    }
#endif
    else if (rows256) e = launch_pk256<7>(g, stream);  // (groups of 128 only: gemm_pk_estimate_us offers forms 6 / 7 for no other group size)
    else if (ks == 3) e = lg == 7 ? launch_pk<1, 7, 0, 2>(g, stream) : (lg == 6 ? launch_pk<1, 6, 0, 2>(g, stream) : launch_pk<1, 5, 0, 2>(g, stream));
    else if (ks == 2) e = lg == 7 ? launch_pk<2, 7>(g, stream) : (lg == 6 ? launch_pk<2, 6>(g, stream) : launch_pk<2, 5>(g, stream));
    else e = lg == 7 ? launch_pk<1, 7>(g, stream) : (lg == 6 ? launch_pk<1, 6>(g, stream) : launch_pk<1, 5>(g, stream));
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
