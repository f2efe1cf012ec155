// w4a16_gemv_i8.hip -- the decode GEMV (M <= 4) as an EXACT int8 contraction on the matrix pipe, on the pre-packed copy of a linear (q4_mfma words + the
// tile-major fp16 scales / zero points tce_w4a16_prepack writes beside them, w4a16_mfma_layout.hpp).  Round 4; replaces the unpack-to-halves GEMV
// (w4a16_gemv.hip) wherever a packed copy comes with the descriptor.  Same math as gemv_kernel_g128 / g64 (reference kernels/cuda/gemv_cuda.cu:140-194, 68-123):
//     C[m][n] = fp16( sum_g fp32(s[n][g]) * sum_{k in g} (q[n][k] - z[n][g]) * A[m][k] )
// Why another kernel: the fp16 form spends 7 vector instructions per 8 weights on the unpack alone and is issue-bound; measured issue rates
// (profiles/r4/valu_probe.jsonl): v_and_or / v_perm / v_cvt / v_dot8_i32_i4 4.2 cycles per wave instruction, plain VOP2 logic 2.4, an i8 16x16x64 MFMA 16, and
// MFMA time ADDS to VALU time on a SIMD.  Here:
//   * WEIGHTS  a lane's 16 bytes of the packed copy (row n16 of a 16-row tile, 32 codes of one 128-wide k unit) become the A operands of two
//              v_mfma_i32_16x16x64_i8 as signed bytes 16 * (q - 8): one v_bitop3 for the high nibbles, shift + v_bitop3 for the low ones -- 3 instructions per
//              8 weights, no zero-point term for the reference's zero point 8 (other zero points: (8 - z) * sum_k X_k from one more MFMA pair with A = 16).
//   * X        every wave converts the 128 * UW activations IT consumes: x * 2^sh truncated to a 31-bit integer under ONE block exponent per wave and row
//              (sh from the block's largest exponent: every element within 2^19 of the block maximum is represented exactly), split into four balanced
//              base-256 digit planes ((I + 0x808080) ^ 0x808080: 2 instructions), byte-transposed into the B operand image in the wave's OWN LDS region: no
//              workgroup barrier in front of the contraction.  The planes are COLUMNS of B: one MFMA contracts all four at once, products and sums exact.
//   * COLUMNS  M = 1 uses 4 of the 16 output columns per 128-k unit, so four units (four quantization groups) share one accumulator: unit c of a pass presents
//              its planes in columns 4c..4c+3 and zeros elsewhere (the B registers of the other lanes are never written).  One v_cvt_f32_i32 + one v_fma_mix
//              (fp16 scale) per output register then serve FOUR groups; M = 2 / 4 rows of A take the columns instead (2 / 1 units per pass).
//   * STREAM   a wave owns ROWS tiles x UW units: every weight byte it will use is requested at once (after the few x / scale bytes: loads retire in order,
//              a wait for x must not be a wait for the weights), 1 KiB of consecutive memory per load instruction, non-temporal.
//   * K        the waves of a workgroup split K (UW depends on K only: a row's bits do not depend on N, so column shards are bit-identical to the whole);
//              partial rows meet in LDS and are added in wave order.
// Numerics: integer part exact; the roundings are one fp32 fma per (row, group), the plane / group / wave additions in a fixed order, the fp16 store.
// Activations holding inf / NaN: every output of that row is NaN (the reference yields NaN or +-inf there).
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"
#include "w4a16_mfma_layout.hpp"

namespace tce {

namespace {

struct I8Seg {
    const void *words;       // u32 [NT16][U][64][4]
    const half_t *dscales;   // fp16 [NT16][NG][16]
    const unsigned *dzeros;  // u32 [NT16][NG][2]
    half_t *C;               // fp16 [M][ldc]
    int N, ldc, epilogue;
    int bytes_w, bytes_s, bytes_z;
    int block_begin;  // first blockIdx.x of this linear
};

struct I8Args {
    const half_t *A;  // fp16 [M][lda]
    int lda, M, K, U, NG;  // U = K / 128 units, NG = K / G groups
    int nseg;
    const float *gamma;  // non-null: the activation is the UN-normalised hidden state; the conversion stages RMSNorm(A) * gamma (generalT5LayerNorm arithmetic; M = 1)
    float eps;
    // RNORM (tce_w4a16_forward_residual_rmsnorm): the residual epilogue also produces the NEXT RMSNorm of the updated row -- out_gamma fp32 [N], xn_out fp16 [N],
    // ws = float slots[2048] (the row's piece sums) + unsigned counter at word 2048 (zero between launches)
    const float *out_gamma;
    float out_eps;
    half_t *xn_out;
    float *ws;
    // COMB (tce_w4a16_forward_deferred_attention, round 5): the activation row is the output of a deferred attention step -- per query head `comb_slots` partial
    // states (M, L, -, -, O[128]; `comb_stride` floats each) of which the first ceil((position + 1) / comb_chunk) are live; the prologue combines them exactly as the
    // attention kernel's own last workgroup would (attention_fast.hip) and rounds to binary16 -- the same row, bit for bit -- while the weights are in flight
    const float *comb_part;
    const int *comb_pos_dev;
    int comb_pos, comb_slots, comb_chunk, comb_stride, comb_heads;
    I8Seg seg[TCE_MAX_GROUP];
};

// ---- the MIXED launch (round 6): up to kI8MixMax decode linears that share NOTHING -- own activation, own K -- as one launch.
// What it is for: with the linears sharded over P ranks and the activations replicated (north_star's one-gather-per-block form) a rank's five linears of a block read
// independent inputs; issued one by one they are 4 launches of 2-6 MB at P = 8, each paying the same ~2.5 us of boundary and ramp as a 60 MB launch.
// Geometry: the workgroup has as many waves as the LONGEST K needs (K / 1024, e.g. 14 for down_proj of Llama-3-8B); a linear with a shorter K packs floor(waves / its own
// waves) tiles into one workgroup -- wave groups of ITS K-split width, each with its own LDS region, running the unchanged body.  A row's bits are therefore exactly those
// of the ordinary launch (same waves per tile, same order of every sum).  The workgroup barrier of the K reduction is shared by the wave groups (each reaches it once);
// waves that belong to no group end at once (a barrier counts the surviving waves only).
constexpr int kI8MixMax = 8;
struct I8MixSeg {
    const half_t *A;
    int lda, K;
    I8Seg seg;  // block_begin: first blockIdx.x of this linear
};
struct I8MixArgs {
    int nseg, M;
    // (what the body reads from I8Args whatever the form; the prologues / epilogues that use them are not compiled into the mixed form)
    const float *gamma = nullptr;
    float eps = 0.f;
    const float *out_gamma = nullptr;
    float out_eps = 0.f;
    half_t *xn_out = nullptr;
    float *ws = nullptr;
    const float *comb_part = nullptr;
    const int *comb_pos_dev = nullptr;
    int comb_pos = 0, comb_slots = 0, comb_chunk = 0, comb_stride = 0, comb_heads = 0;
    // the exchange of ONE linear's output slice with the other ranks inside this launch (tce_w4a16_forward_independent_gather): linear g_seg (-1: none), its g_tiles
    // 16-row tiles count themselves in on the slot's arrival counter
    int g_seg = -1;
    unsigned g_tiles = 0;
    PeerGatherEpi g;
    I8MixSeg s[kI8MixMax];
};

template <int DPP_CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, DPP_CTRL, ROW_MASK, 0xF, false);
    return v > t ? v : t;
}
template <int DPP_CTRL>
__device__ __forceinline__ float dpp_add_f32(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), DPP_CTRL, 0xF, 0xF, false);
    return v + __builtin_bit_cast(float, t);
}

// MB rows of A per pass over the weights (1 / 2 / 4), GPU groups per 128-k unit (1 / 2 / 4 for G = 128 / 64 / 32), ROWS tiles per wave, UW units per wave.
// Output column j of the MFMA = ((uu * GPU + gi) * MB + m) * 4 + p: unit-in-pass uu, group-in-unit gi, activation row m, digit plane p.
// NORM: the fused RMSNorm prologue (LlamaRMSNorm.cu:68-93 in front of q/k/v and gate/up, Int4llamaDecoderLayer.cu:78, 92-99): the waves' piece sums of x^2 meet in LDS
// (tce_common.hpp's shape-independent order: the same rs, bit for bit, as tce_rmsnorm_half and the fp16 GEMV's prologue), one barrier, then every wave normalises the
// activations it converts.  (A form that factors rs out of the row -- x * gamma kept in fp32, rs applied to the finished rows behind the barrier the K reduction needs
// anyway: no barrier and no second pass in front of the contraction -- ran the whole token 2.6 % faster and FAILED parity: the reference rounds the normalised
// activation to binary16 before the linear, and that rounding is 2.8e-4 of the output's rms, 18 x the tolerance floor of outputs near zero.  Round 4, not kept.)
// RNORM: o_proj / down_proj + residual add + the RMSNorm that FOLLOWS (post_attention_layernorm / the next layer's input_layernorm, Int4llamaDecoderLayer.cu:86-99,
// 107-108 + :78 of the next layer) as one launch: every workgroup stores its 16 updated residual values and their two piece sums of squares write-through (system
// scope: the per-XCD L2s are not coherent), counts itself in; the workgroup that arrives LAST forms rs in tce_common.hpp's order from all piece sums and writes the
// normalised row once -- instead of every workgroup of the NEXT launch normalising all of x behind a barrier (its fused prologue: +1.7 / +5.5 us on the q/k/v and
// gate/up launches; the next launch is then the plain kernel).  Bits: C as TCE_W4_ADD_TO_C, xn as tce_rmsnorm_half on the updated row.
// MEASURED (round 4): correct and SLOWER than the prologue it replaces -- the write-through stores must be acknowledged before a workgroup may count itself in and the last
// workgroup's pass is serial: +3.8 us per producer launch, a whole token 1.65 against 1.41 ms.  Kept as an entry point (tests pin its bits); DecoderBlock.step does not use it.
// ARGS = I8MixArgs: the mixed launch (above) -- the linear, its activation and k range per WORKGROUP, the tile per wave group
template <int MB, int GPU, int ROWS, int UW, bool Z8, int MAXT, bool NORM = false, bool RNORM = false, int COMB = 0, typename ARGS = I8Args>
__global__ __launch_bounds__(MAXT) void w4a16_gemv_i8_kernel(const ARGS args) {
    constexpr bool MIX = std::is_same<ARGS, I8MixArgs>::value;
    static_assert(!MIX || (MB == 1 && GPU == 1 && ROWS == 1 && UW == 8 && !NORM && !RNORM && !COMB), "the mixed launch: one decode row, groups of 128, one tile per wave group");
    static_assert(!COMB || (MB == 1 && ROWS == 1 && GPU == 1 && UW == 8 && !NORM && !RNORM), "the deferred-attention prologue: one decode row, groups of 128, K a multiple of 1024");
    static_assert(MB * GPU <= 4, "sixteen output columns: rows x groups-per-unit x 4 planes");
    static_assert(!RNORM || (MB == 1 && ROWS == 1 && !NORM), "the residual + next-norm epilogue: one decode row, one tile per workgroup");
    static_assert(!NORM || MB == 1, "the fused RMSNorm prologue is a decode (M = 1) feature");
    constexpr int UPP = 4 / (MB * GPU);  // units per pass
    static_assert(UW % UPP == 0 && UW % 4 == 0, "whole passes; whole 16-byte x loads per lane");
    constexpr int NP = UW / UPP;         // passes
    constexpr int XC = UW / 4;           // 8-element activation chunks per lane and row (64 lanes x 8 = 4 units)
    constexpr int SPG = 4 / GPU;         // dwords of a lane's B operand that belong to one group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wg[];
    unsigned char *smem = smem_wg;
    int tid = threadIdx.x;
    int WK = blockDim.x >> 6;
    int U, NG, K_, lda_, tile0;
    const half_t *A_;
    I8Seg seg;
    bool gathers = false;   // MIX: this workgroup's linear is the one whose slice is exchanged inside the launch
    unsigned g_epoch = 0;   // the exchange's epoch: the slot's count of completed exchanges + 1 (the same on every rank: they all run the same sequence per slot)
    if constexpr (MIX) {
        int si = 0;
#pragma unroll
        for (int s = 1; s < kI8MixMax; ++s)
            if (s < args.nseg && (int)blockIdx.x >= args.s[s].seg.block_begin) si = s;
        A_ = args.s[si].A;
        lda_ = args.s[si].lda;
        K_ = args.s[si].K;
        seg = args.s[si].seg;
        U = NG = K_ >> 7;
        const int wl = WK;                               // waves of the launch's workgroups
        WK = (U + UW - 1) / UW;                          // waves that split THIS linear's K
        const int nsub = wl / WK;                        // its tiles per workgroup
        const int sub = __builtin_amdgcn_readfirstlane(tid >> 6) / WK;
        tile0 = ((int)blockIdx.x - seg.block_begin) * nsub + sub;
        if (sub >= nsub || tile0 >= ((seg.N + 15) >> 4)) return;  // (whole wave groups: the barrier below counts the surviving waves)
        tid -= sub * WK * 64;
        smem += (size_t)sub * ((size_t)WK * UW * 512 + (size_t)WK * 16 * sizeof(float));
        gathers = si == args.g_seg;
        // (read before any tile of this launch can have counted itself in: the word changes only after ALL of them have)
        if (gathers) g_epoch = __hip_atomic_load(args.g.epochs + args.g.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    } else {
        U = args.U;
        NG = args.NG;
        K_ = args.K;
        lda_ = args.lda;
        A_ = args.A;
        int si = 0;
#pragma unroll
        for (int s = 1; s < TCE_MAX_GROUP; ++s)
            if (s < args.nseg && (int)blockIdx.x >= args.seg[s].block_begin) si = s;
        seg = args.seg[si];
        tile0 = ((int)blockIdx.x - seg.block_begin) * ROWS;
    }
    const int lane = tid & 63;
    const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = wk * UW;
    const int kq = lane >> 4, j = lane & 15;
    const int m0 = blockIdx.y * MB;
    const int ntiles = (seg.N + 15) >> 4;
    auto tile_of = [&](int r) { return tile0 + r < ntiles ? tile0 + r : ntiles - 1; };  // a surplus tile of the last workgroup re-reads the last one; its stores are masked

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(seg.words), 0, seg.bytes_w, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(seg.dscales), 0, seg.bytes_s, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(seg.dzeros), 0, seg.bytes_z, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(seg.words), 0, 0, 0x00020000);

    // column of this lane
    const int p_l = j & 3;
    const int m_l = (j >> 2) % MB;
    const int gi_l = ((j >> 2) / MB) % GPU;
    const int uu_l = (j >> 2) / (MB * GPU);

    // ---- 1. requests, in the order they are needed: activations, scales (zero points), then every weight byte of the wave ----
    uint4_t xv[MB][XC];
    // COMB: how many chunk slots of the deferred attention step are live (wave-uniform; one: that launch wrote the row itself and it is read like any other)
    int nact = 1;
    if constexpr (COMB) {
        const int pos = args.comb_pos_dev ? __builtin_amdgcn_readfirstlane(*args.comb_pos_dev) : args.comb_pos;
        nact = (pos + args.comb_chunk) / args.comb_chunk;  // ceil((pos + 1) / chunk)
        nact = nact < args.comb_slots ? nact : args.comb_slots;
    }
    constexpr int kSlots = COMB ? COMB : 1;              // COMB = the slots the prologue is compiled for: 4 or 8 (kAttnDeferMaxSlots); the launch's slots <= COMB
    uint2_t cml[COMB ? XC : 1][kSlots];                  // (M, L) of the lane's head per slot
    uint4_t co[COMB ? XC : 1][kSlots][2];                // O[d0 .. d0 + 7] per slot
    if constexpr (COMB) {
        if (nact > 1) {
            const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(args.comb_part), 0, args.comb_heads * args.comb_slots * args.comb_stride * 4, 0x00020000);
#pragma unroll
            for (int c = 0; c < XC; ++c) {
                const int kpos = u0 * 128 + (lane + 64 * c) * 8;       // the lane's 8 columns: head kpos / 128, dimensions kpos % 128 ..
                const int hbase = ((kpos >> 7) * args.comb_slots) * args.comb_stride * 4 + (kpos & 127) * 4;
#pragma unroll
                for (int i = 0; i < kSlots; ++i) {
                    // dead slots re-read the first one (their weights are never formed): plain loads -- every lane's addresses are valid, nothing needs a range check
                    const int ii = i < nact ? i : 0;
                    const unsigned char *src = reinterpret_cast<const unsigned char *>(args.comb_part) + hbase + ii * args.comb_stride * 4;
                    cml[c][i] = *reinterpret_cast<const uint2_t *>(src - (kpos & 127) * 4);
                    co[c][i][0] = *reinterpret_cast<const uint4_t *>(src + 16);
                    co[c][i][1] = *reinterpret_cast<const uint4_t *>(src + 32);
                }
                (void)rs_p;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        if (COMB && nact > 1) break;
        int mrow = m0 + m;
        mrow = mrow < args.M ? mrow : args.M - 1;
        // a descriptor per row, K halves long: chunks past K (a ragged last wave) read as zeros
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(A_ + (size_t)mrow * lda_), 0, K_ * 2, 0x00020000);
#pragma unroll
        for (int c = 0; c < XC; ++c) xv[m][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, u0 * 256 + (lane + 64 * c) * 16, 0, 0);  // (the scalar offset of a buffer load is NOT range-checked: everything that may run past K sits in the vector offset)
    }
    // the residual epilogue's old values (TCE_W4_ADD_TO_C): requested HERE, by the lanes that will add to them -- behind the barrier of the K reduction the load was a
    // memory round trip at the very end of every o_proj / down_proj launch (the row was written by another launch, possibly on another XCD: no cache holds it)
    half_t c_old = (half_t)0.0f;
    if constexpr (!RNORM) {
        if (wk == 0 && (seg.epilogue & TCE_W4_ADD_TO_C) && tid < ROWS * MB * 16) {
            const int i16 = tid & 15, m = (tid >> 4) % MB, r = tid / (16 * MB);
            const int row = (tile0 + r) * 16 + i16;
            if (m0 + m < args.M && row < seg.N) c_old = seg.C[(size_t)(m0 + m) * seg.ldc + row];
        }
    }
    float4_t gm[NORM ? XC : 1][2];  // gamma of the lane's 8 columns per chunk
    if constexpr (NORM) {
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(args.gamma), 0, K_ * 4, 0x00020000);
#pragma unroll
        for (int c = 0; c < XC; ++c) {
            gm[c][0] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_g, u0 * 512 + (lane + 64 * c) * 32, 0, 0));
            gm[c][1] = __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_g, u0 * 512 + (lane + 64 * c) * 32 + 16, 0, 0));
        }
    }
    uint2_t sc[ROWS][NP];
    unsigned zq[ROWS][(NP + 1) / 2];  // zero points other than 8: the lane's four rows' nibbles of pass ps in half (ps & 1) of word ps / 2 (two passes per register: the general-zero-point forms are the ones short of registers)
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int g = (ps * UPP + uu_l) * GPU + gi_l;  // group of this lane's column, relative to the wave's first group
            const int ga = (tile_of(r) * NG + u0 * GPU) + g;  // groups past K (a ragged last wave) read the next tile's values or, behind the last tile, zeros: their products are zero
            sc[r][ps] = __builtin_bit_cast(uint2_t, __builtin_amdgcn_raw_buffer_load_b64(rs_s, (ga * 16 + 4 * kq) * 2, 0, 0));
            if constexpr (!Z8) {
                const unsigned z16 = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs_z, ga * 8 + 2 * kq, 0, 0);
                zq[r][ps >> 1] = (ps & 1) ? (zq[r][ps >> 1] | (z16 << 16)) : z16;
            }
        }
    __builtin_amdgcn_sched_barrier(0);
    uint4_t w[ROWS][UW];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int t = 0; t < UW; ++t) {
            // units past K (a ragged last wave): an empty descriptor -- answered with zeros by the buffer unit, no memory traffic
            w[r][t] = __builtin_amdgcn_raw_buffer_load_b128(u0 + t < U ? rs_w : rs_none, lane * 16, (tile_of(r) * U + u0 + t) * 1024, /*nt*/ 2);
        }
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (COMB) {
        if (nact > 1) {
            // attention_fast.hip's combine, operation for operation: Mx = max M_i; w_i = exp(M_i - Mx); L = sum L_i w_i, O = sum O_i w_i in slot order (products and sums
            // rounded separately: this file is compiled without contraction); x = half(O / L)
#pragma unroll
            for (int c = 0; c < XC; ++c) {
                // (an element of a vector goes through a scalar before its bits are reinterpreted: __builtin_bit_cast applied to `v[k]` directly read v[0] for every k --
                //  found by tests/test_gpu_deferred_attention.py: M for L, O[0] for O[1..3])
                auto f32 = [](unsigned u) { return __builtin_bit_cast(float, u); };
                float Mx = -1.0e30f;
#pragma unroll
                for (int i = 0; i < kSlots; ++i)
                    if (i < nact) Mx = __builtin_fmaxf(Mx, f32(cml[c][i][0]));
                float Lx = 0.f, Ox[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) Ox[e] = 0.f;
#pragma unroll
                for (int i = 0; i < kSlots; ++i) {
                    if (i < nact) {  // (wave-uniform)
                        const float w = __expf(f32(cml[c][i][0]) - Mx);
                        Lx += f32(cml[c][i][1]) * w;
#pragma unroll
                        for (int e = 0; e < 8; ++e) Ox[e] += f32(co[c][i][e >> 2][e & 3]) * w;
                    }
                }
                half8_t y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (half_t)(Ox[e] / Lx);
                xv[0][c] = __builtin_bit_cast(uint4_t, y);
            }
        }
    }
    // ---- 2. the wave's activations -> digit planes, image [unit][lo/hi][m][plane][kq][word s] in its own LDS region ----
    unsigned *planes = reinterpret_cast<unsigned *>(smem) + wk * (UW * MB * 128);
    if constexpr (NORM) {
        // slot[p] = the sum of squares of the row's p-th 16-byte piece (fmaf chain over its 8 values); K <= 16384: at most two pieces per slot of the
        // 1024-slot order, and a two-term sum does not depend on which wave wrote which
        float *slots = reinterpret_cast<float *>(smem + (size_t)WK * UW * MB * 512 + (size_t)WK * ROWS * MB * 16 * sizeof(float));  // [2048]
        const int pieces = K_ >> 3;
#pragma unroll
        for (int c = 0; c < XC; ++c) {
            const int p = u0 * 16 + lane + 64 * c;
            if (p < pieces + 128) slots[p] = rmsnorm_piece_sum(__builtin_bit_cast(half8_t, xv[0][c]));  // (pieces past K: zeros were loaded)
        }
        lds_barrier();  // (LDS only: every weight byte of the wave stays in flight)
        // (the order's sixteen terms per lane; the terms past the row's pieces are zeros: only the first ceil(pieces / 64) are read -- 8 for K = 4096)
        float tot = 0.f;
        const int nc = pieces >= 1024 ? 16 : (pieces + 63) >> 6;
        if (pieces <= 1024 && (pieces & 63) == 0) {
            // whole 64-piece terms, no second piece per slot (K = 4096: 8 reads at fixed offsets and 8 adds; the same sum, without the bounds arithmetic of the
            // general form below -- measured 0.25 / 0.65 us per launch on norm + q/k/v and norm + gate/up of a 4096-wide layer)
            const float *sl = slots + lane;
            for (int c = 0; c < nc; ++c) tot += sl[c * 64];
        } else {
            for (int c = 0; c < nc; ++c) {
                const int q = c * 64 + lane;
                float slot = 0.f;
                slot += q < pieces ? slots[q] : 0.f;
                if (q + 1024 < pieces) slot += slots[q + 1024];
                tot += slot;
            }
        }
        tot = wave_sum_dpp_lane63(tot);
        tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 63));
        const float rs = 1.0f / sqrtf(tot / (float)K_ + args.eps);
#pragma unroll
        for (int c = 0; c < XC; ++c) {
            const half8_t v = __builtin_bit_cast(half8_t, xv[0][c]);
            half8_t y;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // rmsnorm_out's arithmetic -- half(clamp((x * rs) * gamma)) -- with the clamp as ONE v_med3_f32 (the same value for every finite product; a NaN / inf
                // activation poisons the row anyway): 3 vector instructions fewer per element, 48 per wave, in a kernel that is bound by issue
                const float f = ((float)v[e] * rs) * (e < 4 ? gm[c][0][e] : gm[c][1][e - 4]);
                y[e] = (half_t)__builtin_amdgcn_fmed3f(f, -(65504.f - 1000.f), 65504.f - 1000.f);
            }
            xv[0][c] = __builtin_bit_cast(uint4_t, y);
        }
    }
    int sh[MB];
    bool bad[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        unsigned mx = 0;
#pragma unroll
        for (int c = 0; c < XC; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned v = xv[m][c][q] & 0x7FFF7FFFu;
                const unsigned v2 = v << 16;
                const unsigned m2 = v > v2 ? v : v2;  // bits 30..26: the larger exponent field of the two halves
                mx = mx > m2 ? mx : m2;
            }
        // wave-wide maximum on the VALU: row_shr 1 2 4 8 inside rows of 16, then row_bcast15 / row_bcast31; the total sits in lane 63
        mx = dpp_max_u32<0x111>(mx);
        mx = dpp_max_u32<0x112>(mx);
        mx = dpp_max_u32<0x114>(mx);
        mx = dpp_max_u32<0x118>(mx);
        mx = dpp_max_u32<0x142, 0xA>(mx);
        mx = dpp_max_u32<0x143, 0xC>(mx);
        const int E = (int)((unsigned)__builtin_amdgcn_readlane((int)mx, 63) >> 26);  // exponent field of the block's largest magnitude: |x| < 2^(E - 14)
        bad[m] = E == 31;
        sh[m] = 44 - E;  // |x * 2^sh| < 2^30
        const float scale = __builtin_bit_cast(float, (unsigned)(127 + sh[m]) << 23);
#pragma unroll
        for (int c = 0; c < XC; ++c) {
            unsigned d[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const half_t lo = __builtin_bit_cast(half_t, (unsigned short)(xv[m][c][q] & 0xFFFFu));
                const half_t hi = __builtin_bit_cast(half_t, (unsigned short)(xv[m][c][q] >> 16));
                // exact product (11 significant bits), truncated to an integer; balanced digits: the bytes of (I + 0x808080) ^ 0x808080
                d[2 * q] = ((unsigned)(int)__builtin_fmaf((float)lo, scale, 0.0f) + 0x00808080u) ^ 0x00808080u;
                d[2 * q + 1] = ((unsigned)(int)__builtin_fmaf((float)hi, scale, 0.0f) + 0x00808080u) ^ 0x00808080u;
            }
            // chunk cc of the wave's block = the 8 activations of (unit tu, word s, k-quarter q): k = 128 tu + 32 s + 8 q + e
            const int cc = lane + 64 * c;
            const int tu = cc >> 4, s = (cc >> 2) & 3, q4 = cc & 3;
            // a packed word's low nibbles are the codes e = (0, 4, 1, 5), its high nibbles e = (2, 6, 3, 7) (pk::nibble_index)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned a0 = d[2 * h], a1 = d[2 * h + 4], a2 = d[2 * h + 1], a3 = d[2 * h + 5];
                const unsigned t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u);  // (a0.b0, a1.b0, a0.b1, a1.b1)
                const unsigned t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);  // (a0.b2, a1.b2, a0.b3, a1.b3)
                const unsigned t2 = __builtin_amdgcn_perm(a3, a2, 0x05010400u);
                const unsigned t3 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
                const int base = ((((tu * 2 + h) * MB + m) * 4) * 4 + q4) * 4 + s;  // + plane * 16
                planes[base + 0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
                planes[base + 16] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
                planes[base + 32] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
                planes[base + 48] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- 3. contraction.  B set uu: the operand of unit uu of the pass in the lanes of ITS columns, zeros in the others (never written) ----
    int4_t B[UPP][2];
#pragma unroll
    for (int uu = 0; uu < UPP; ++uu) B[uu][0] = B[uu][1] = int4_t{0, 0, 0, 0};
    auto read_b = [&](int ps) {
#pragma unroll
        for (int uu = 0; uu < UPP; ++uu)
#pragma unroll
            for (int gi = 0; gi < GPU; ++gi)
                if (uu_l == uu && gi_l == gi) {  // exec-masked reads
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned *src = planes + ((((((ps * UPP + uu) * 2 + h) * MB + m_l) * 4 + p_l) * 4 + kq) * 4 + gi * SPG);
                        if constexpr (GPU == 1) {
                            B[uu][h] = __builtin_bit_cast(int4_t, *reinterpret_cast<const uint4_t *>(src));
                        } else if constexpr (GPU == 2) {
                            const uint2_t v = *reinterpret_cast<const uint2_t *>(src);
                            B[uu][h][gi * 2] = (int)v[0];
                            B[uu][h][gi * 2 + 1] = (int)v[1];
                        } else {
                            B[uu][h][gi] = (int)*src;
                        }
                    }
                }
    };
    read_b(0);

    float acc[ROWS][4];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r][q] = 0.f;
    const int4_t zero4 = int4_t{0, 0, 0, 0};
    const int4_t sixteens = int4_t{0x10101010, 0x10101010, 0x10101010, 0x10101010};
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        // general zero points: the scheduler would otherwise unpack every pass's (8 - z) nibbles ahead of the first MFMA (NP x 4 registers: 19 spilled in the
        // M = 2 / G = 64 form under the 128-register bound of a 16-wave workgroup).  A pass's work stays inside the pass.
        if constexpr (!Z8) __builtin_amdgcn_sched_barrier(0);
        int4_t dd[ROWS][2];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) dd[r][0] = dd[r][1] = zero4;
        int4_t dz = zero4;  // 16 * sum of the digits of the column's group (zero points other than 8)
#pragma unroll
        for (int uu = 0; uu < UPP; ++uu) {
            const int t = ps * UPP + uu;
            if constexpr (!Z8) {
                dz = __builtin_amdgcn_mfma_i32_16x16x64_i8(sixteens, B[uu][0], dz, 0, 0, 0);
                dz = __builtin_amdgcn_mfma_i32_16x16x64_i8(sixteens, B[uu][1], dz, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                int4_t alo, ahi;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // signed bytes 16 (q - 8): (nibble in the high half of the byte) ^ 0x80.  Written as two independent and-xor forms so that each is ONE
                    // v_bitop3 (a shared w ^ 0x88888888 costs a fourth instruction per word)
                    const unsigned wq = w[r][t][q];
                    ahi[q] = (int)((wq & 0xF0F0F0F0u) ^ 0x80808080u);
                    alo[q] = (int)(((wq << 4) & 0xF0F0F0F0u) ^ 0x80808080u);
                }
                dd[r][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(alo, B[uu][0], dd[r][0], 0, 0, 0);
                if constexpr (Z8) dd[r][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ahi, B[uu][1], dd[r][1], 0, 0, 0);
                else dd[r][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ahi, B[uu][1], dd[r][0], 0, 0, 0);  // (one accumulator chain: integer sums, the same value; four registers fewer beside dz and the sixteens)
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const half_t s0 = __builtin_bit_cast(half_t, (unsigned short)(sc[r][ps][0] & 0xFFFFu));
            const half_t s1 = __builtin_bit_cast(half_t, (unsigned short)(sc[r][ps][0] >> 16));
            const half_t s2 = __builtin_bit_cast(half_t, (unsigned short)(sc[r][ps][1] & 0xFFFFu));
            const half_t s3 = __builtin_bit_cast(half_t, (unsigned short)(sc[r][ps][1] >> 16));
            int4_t tot = dd[r][0] + dd[r][1];
            if constexpr (!Z8) {
                // sum (q - z) X = sum (q - 8) X + (8 - z) sum X, in the same units of 16
#pragma unroll
                for (int q = 0; q < 4; ++q) tot[q] += __mul24(8 - (int)((zq[r][ps >> 1] >> (16 * (ps & 1) + 4 * q)) & 0xFu), dz[q]);
            }
            acc[r][0] = __builtin_fmaf((float)tot[0], (float)s0, acc[r][0]);
            acc[r][1] = __builtin_fmaf((float)tot[1], (float)s1, acc[r][1]);
            acc[r][2] = __builtin_fmaf((float)tot[2], (float)s2, acc[r][2]);
            acc[r][3] = __builtin_fmaf((float)tot[3], (float)s3, acc[r][3]);
        }
        if (ps + 1 < NP) read_b(ps + 1);
    }

    // ---- 4. planes and groups -> the wave's partial rows (fixed lane order), waves -> rows (wave order), store ----
    int sh_l = sh[0];
    bool bad_l = bad[0];
#pragma unroll
    for (int m = 1; m < MB; ++m) {
        sh_l = m_l == m ? sh[m] : sh_l;
        bad_l = m_l == m ? bad[m] : bad_l;
    }
    // 2^(8 p) * 2^(-sh) / 16 (the A operand carries 16 (q - 8))
    const float cj = bad_l ? __builtin_nanf("") : __builtin_bit_cast(float, (unsigned)(127 + 8 * p_l - sh_l - 4) << 23);
    float *red = reinterpret_cast<float *>(smem + (size_t)WK * UW * MB * 512);  // [WK][ROWS][MB][16]
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = acc[r][q] * cj;
            v = dpp_add_f32<0xB1>(v);  // quad_perm [1,0,3,2]: planes
            v = dpp_add_f32<0x4E>(v);  // quad_perm [2,3,0,1]
            if constexpr (MB <= 2) v = dpp_add_f32<0x128>(v);  // row_ror 8
            if constexpr (MB == 1) v = dpp_add_f32<0x124>(v);  // row_ror 4
            if (p_l == 0 && (j >> 2) < MB) red[((wk * ROWS + r) * MB + m_l) * 16 + kq * 4 + q] = v;
        }
    __syncthreads();
    if (tid < ROWS * MB * 16) {
        float part[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) part[k2] = red[(k2 < WK ? k2 : 0) * (ROWS * MB * 16) + tid];
        float v = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) v += k2 < WK ? part[k2] : 0.f;
        const int i16 = tid & 15, m = (tid >> 4) % MB, r = tid / (16 * MB);
        const int row = (tile0 + r) * 16 + i16;  // (unclamped: a surplus tile's rows are >= N)
        const half_t y = (half_t)v;
        // the (gate, up) neighbour of the pair epilogue: rows 2n, 2n + 1 are adjacent lanes
        const half_t y_other = __builtin_bit_cast(half_t, (unsigned short)__builtin_amdgcn_update_dpp(0, (int)__builtin_bit_cast(unsigned short, y), 0xB1, 0xF, 0xF, false));
        half_t hnew = (half_t)0.0f;
        (void)hnew;
        if (m0 + m < args.M && row < seg.N) {
            half_t *crow = seg.C + (size_t)(m0 + m) * seg.ldc;
            if (seg.epilogue & TCE_W4_SILU_MUL_PAIRS) {
                if ((i16 & 1) == 0) crow[row >> 1] = silu_mul_half(y, y_other);
            } else if (seg.epilogue & TCE_W4_ADD_TO_C) {
                if constexpr (RNORM) hnew = crow[row] + y;  // (stored below, write-through)
                else crow[row] = c_old + y;
            } else {
                crow[row] = y;
            }
            if constexpr (MIX) {
                if (gathers) {
                    // (a) the tile's 16 values into EVERY rank's window -- buffer (slot, parity of the epoch), this rank's slice -- past every cache (comm.hip's protocol)
                    const half_t outv = (seg.epilogue & TCE_W4_ADD_TO_C) ? (half_t)(c_old + y) : y;
                    const size_t off = ((size_t)args.g.slot * 2 + (g_epoch & 1u)) * args.g.vec_bytes + ((size_t)args.g.rank * args.g.slice_elems + (size_t)row) * 2;
#pragma unroll
                    for (int p = 0; p < kCommMaxRanks; ++p)
                        if (p < args.g.world)
                            __hip_atomic_store(reinterpret_cast<unsigned short *>(args.g.peer[p] + off), __builtin_bit_cast(unsigned short, outv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        if constexpr (RNORM) {
            // the tile's 16 updated values -> memory, past every cache (the last workgroup reads them from another XCD); their two piece sums likewise
            if (row < seg.N) __hip_atomic_store(reinterpret_cast<unsigned short *>(seg.C) + row, __builtin_bit_cast(unsigned short, hnew), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const float hf = row < seg.N ? (float)hnew : 0.f;
            float s0 = 0.f, s1 = 0.f;  // fmaf chains over the two pieces' 8 values, in order (rmsnorm_piece_sum), formed by every lane from broadcasts
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hf), e));
                const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hf), 8 + e));
                s0 = __builtin_fmaf(v0, v0, s0);
                s1 = __builtin_fmaf(v1, v1, s1);
            }
            if (tid < 2) __hip_atomic_store(args.ws + 2 * tile0 + tid, tid == 0 ? s0 : s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if constexpr (MIX) {
        if (gathers && tid < 64) {  // (the tile's first wave: the one that stored)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores above have been acknowledged before the tile counts itself in
            unsigned *const counter = args.g.epochs + args.g.slots + 1 + args.g.slot;
            unsigned last = 0;
            if (tid == 0) last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == args.g_tiles - 1u ? 1u : 0u;
            last = (unsigned)__builtin_amdgcn_readfirstlane((int)last);
            if (last) {
                // ---- the last tile of the slice: flags out, the other ranks' flags in, the complete vector to the consumer's buffer (allgather_peer_kernel's b, c, d) ----
                const unsigned e = g_epoch, par = e & 1u;
                __threadfence_system();
                if (tid < args.g.world) {
                    unsigned *flag = reinterpret_cast<unsigned *>(args.g.peer[tid] + args.g.flags_off) + (((size_t)args.g.slot * 2 + par) * kCommMaxRanks + args.g.rank) * kCommFlagStride;
                    __hip_atomic_store(flag, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (tid < args.g.world) {
                    const unsigned *flag = reinterpret_cast<const unsigned *>(args.g.peer[args.g.rank] + args.g.flags_off) + (((size_t)args.g.slot * 2 + par) * kCommMaxRanks + tid) * kCommFlagStride;
                    const unsigned long long t0 = wall_clock64();
                    const bool dead = __hip_atomic_load(args.g.epochs + args.g.slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                    while (!dead && (int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
                        __builtin_amdgcn_s_sleep(8);
                        if (wall_clock64() - t0 > args.g.timeout_ticks) {  // a rank that never arrives: give up, flag the communicator (tce_comm_status; tce_comm_reset re-arms)
                            __hip_atomic_store(args.g.epochs + args.g.slots, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __threadfence_system();  // acquire: the peers' data behind their flags
                const unsigned char *full = args.g.peer[args.g.rank] + ((size_t)args.g.slot * 2 + par) * args.g.vec_bytes;
                const unsigned total16 = args.g.slice_elems * (unsigned)args.g.world / 8u;
                for (unsigned base = 0; base < total16; base += 64 * 8) {
                    uint4_t v[8];  // eight pieces per lane in flight (one wave carries the whole vector: 8 KiB at 4096 halves)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const unsigned i = base + q * 64 + tid;
                        const unsigned ic = i < total16 ? i : total16 - 1;
                        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[q]) : "v"(full + (size_t)ic * 16) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const unsigned i = base + q * 64 + tid;
                        if (i < total16) reinterpret_cast<uint4_t *>(args.g.dst)[i] = v[q];
                    }
                }
                if (tid == 0) {
                    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next exchange on this slot (ordered by the kernel boundary)
                    __hip_atomic_store(args.g.epochs + args.g.slot, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    if constexpr (RNORM) {
        unsigned *flag = reinterpret_cast<unsigned *>(smem);  // (the planes are dead)
        if (tid < 64) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores above have left this wave before it counts itself in
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(args.ws) + 2048, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                flag[0] = old == gridDim.x - 1 ? 1u : 0u;
            }
        }
        __syncthreads();
        if (flag[0] == 0u) return;
        // ---- the last workgroup: rs, then the normalised row ----
        if (tid == 0) reinterpret_cast<unsigned *>(args.ws)[2048] = 0u;  // for the next launch on this workspace (ordered by the kernel boundary)
        const int n = seg.N, pieces = n >> 3;
        auto sys_load_f32 = [](const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
        float tot = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int q = c * 64 + lane;
            float slot = 0.f;
            slot += q < pieces ? sys_load_f32(args.ws + q) : 0.f;
            if (q + 1024 < pieces) slot += sys_load_f32(args.ws + q + 1024);
            tot += slot;
        }
        tot = wave_sum_dpp_lane63(tot);
        tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 63));
        const float rs = 1.0f / sqrtf(tot / (float)n + args.out_eps);
        const half_t *hrow = seg.C;
        for (int p = tid; p < pieces; p += (int)blockDim.x) {
            uint4_t raw;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(raw) : "v"(hrow + p * 8) : "memory");
            const half8_t hv = __builtin_bit_cast(half8_t, raw);
            const float4_t g0 = *reinterpret_cast<const float4_t *>(args.out_gamma + p * 8), g1 = *reinterpret_cast<const float4_t *>(args.out_gamma + p * 8 + 4);
            half8_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = rmsnorm_out(hv[e], rs, g0[e]);
                o[4 + e] = rmsnorm_out(hv[4 + e], rs, g1[e]);
            }
            *reinterpret_cast<half8_t *>(args.xn_out + p * 8) = o;
        }
    }
}

static int i8_units_per_wave(int U) { return U <= 128 ? 8 : (U <= 256 ? 16 : 0); }
thread_local int g_i8_mode = 0;  // 0 automatic (wherever the shape allows and a packed copy is given), 1 off
thread_local int g_i8_rows = 0;  // 0 the rule, 1 / 2 forced tiles per wave

template <int MB, int GPU, int ROWS, int UW, bool Z8, int MAXT, bool NORM = false, bool RNORM = false, int COMB = 0>
hipError_t launch_i8(const I8Args &a, int blocks, int m_blocks, int wk, hipStream_t stream) {
    const size_t lds = (size_t)wk * UW * MB * 512 + (size_t)wk * ROWS * MB * 16 * sizeof(float) + (NORM ? (size_t)(a.K >> 3) * sizeof(float) + 1024 : 0);  // the row's piece sums (+ the ragged last wave's zero pieces)
    auto kfn = w4a16_gemv_i8_kernel<MB, GPU, ROWS, UW, Z8, MAXT, NORM, RNORM, COMB>;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(blocks, m_blocks, 1), dim3(64 * wk, 1, 1), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

// rows of A per pass of the weights: 1 / 2 / 4, bounded by the sixteen columns of the MFMA and by the digit planes fitting LDS twice per CU (M = 4 at K = 11008 / 14336:
// one row per pass).  ONE rule for the launcher and for tce_w4a16_describe_dispatch.
int gemv_i8_rows_per_pass(int M, int K, int group_size) {
    const int U = K / 128, gpu = 128 / group_size;
    const int uw = i8_units_per_wave(U);
    const int wk = (U + uw - 1) / uw;
    int mb = M >= 3 ? 4 : M;
    while (mb > 1 && (mb * gpu > 4 || (size_t)wk * uw * mb * 512 > 72 * 1024)) mb >>= 1;
    return mb;
}

void set_gemv_i8_mode(int mode, int rows) {
    g_i8_mode = mode == 1 ? 1 : 0;
    g_i8_rows = rows == 1 || rows == 2 ? rows : 0;
}

// units per wave: a function of K (and of the rows / groups per pass) ONLY, so that a row's arithmetic does not depend on N

bool gemv_i8_supports(const tce_w4a16_desc *descs, int count, bool with_norm) {
    if (g_i8_mode == 1) return false;
    const tce_w4a16_desc &d0 = descs[0];
    if (d0.M < 1 || d0.M > 4 || d0.K % 128 != 0) return false;
    if ((d0.rmsnorm_gamma || with_norm) && (d0.M != 1 || d0.group_size != 128 || d0.K > 16384)) return false;  // the fused RMSNorm prologue: decode rows, groups of 128
    if (d0.group_size != 128 && d0.M > (d0.group_size == 64 ? 2 : 1)) return false;
    const int uw = i8_units_per_wave(d0.K / 128);
    if (uw == 0 || (uw == 16 && (d0.M > 1 || d0.group_size != 128))) return false;  // (very long K: single rows only -- the registers hold 16 KiB of weights per wave)
    for (int i = 0; i < count; ++i) {
        const tce_w4a16_desc &d = descs[i];
        if (!d.prepacked || (reinterpret_cast<uintptr_t>(d.prepacked) & 255)) return false;
        if ((long long)pk::nt16(d.N) * (d.K / 2) * 16 >= (1LL << 31)) return false;  // buffer descriptors address 32-bit byte offsets
        if (d.flags & TCE_W4_FORCE_GEMM) return false;
    }
    return true;
}

int launch_w4a16_gemv_i8(const tce_w4a16_desc *descs, int count, hipStream_t stream, hipError_t *hip_err, const float *gamma, float eps, const I8ResidualNorm *rn,
                         const AttnDeferred *comb, const int *comb_pos_dev, int comb_pos) {
    const tce_w4a16_desc &d0 = descs[0];
    if (!gamma && d0.rmsnorm_gamma) {
        gamma = static_cast<const float *>(d0.rmsnorm_gamma);
        eps = d0.rmsnorm_eps;
    }
    I8Args a{};
    a.gamma = gamma;
    a.eps = eps;
    a.A = static_cast<const half_t *>(d0.A);
    a.lda = d0.lda ? d0.lda : d0.K;
    a.M = d0.M;
    a.K = d0.K;
    a.U = d0.K / 128;
    a.NG = d0.K / d0.group_size;
    a.nseg = count;
    const int gpu = 128 / d0.group_size;
    const int uw = i8_units_per_wave(a.U);
    const int wk = (a.U + uw - 1) / uw;
    bool z8 = true;
    long long total_tiles = 0;
    for (int i = 0; i < count; ++i) {
        z8 = z8 && (descs[i].flags & TCE_W4_ZERO_POINT_IS_8);
        total_tiles += pk::nt16(descs[i].N);
    }
    const int mb = gemv_i8_rows_per_pass(d0.M, d0.K, d0.group_size);
    const int m_blocks = (d0.M + mb - 1) / mb;
    // tiles per wave: one.  Two (the conversion of x amortised over twice the bytes, one generation of workgroups for the gate+up launch) measured slower on every
    // launch of the token (profiles/r4/gemv_i8_ab.jsonl: qkv 6.6 -> 7.4 us, o 3.9 -> 4.5, gate+up 9.9 -> 10.2, down a tie); compiled, forceable (tce_w4a16_set_gemv_i8)
    // With the RMSNorm prologue -- x, gamma, the piece sums and the normalisation on top of the conversion, all per wave -- two tiles per wave win where one tile per
    // workgroup leaves a short second generation of workgroups behind the resident ones (256 CUs x 20 waves at five per SIMD: 1280 four-wave workgroups): norm + gate/up of
    // a 4096 x 11008 layer, 1376 tiles, 11.84 -> 11.31 us.  Elsewhere one tile per wave is level or ahead (scripts/probes/norm_tiles_ab.py: 768 tiles 7.1 / 8.3 us,
    // 1024: 8.65 / 9.04, 1280: 10.4 / 11.1, 1536: 11.6 / 11.75, 1792: 12.9 / 13.5, 2000: 14.2 / 14.0, 8016: 43.3 / 43.8).  (The tile count per wave does not enter
    // the arithmetic: same bits.)
    int rows = 1;
    const bool two_ok = mb == 1 && gpu == 1 && uw == 8 && wk <= 8;
    const long long resident = 256LL * (20 / wk);
    if (gamma && two_ok && total_tiles > resident && total_tiles * 5 <= resident * 6) rows = 2;
    if (g_i8_rows && two_ok) rows = g_i8_rows;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const tce_w4a16_desc &d = descs[i];
        I8Seg &s = a.seg[i];
        const unsigned char *base = static_cast<const unsigned char *>(d.prepacked);
        s.words = base;
        s.dscales = reinterpret_cast<const half_t *>(base + pk::dscales_offset(d.N, d.K, d.group_size));
        s.dzeros = reinterpret_cast<const unsigned *>(base + pk::dzeros_offset(d.N, d.K, d.group_size));
        s.C = static_cast<half_t *>(d.C);
        s.N = d.N;
        s.epilogue = d.flags & (TCE_W4_SILU_MUL_PAIRS | TCE_W4_ADD_TO_C);
        s.ldc = d.ldc ? d.ldc : ((s.epilogue & TCE_W4_SILU_MUL_PAIRS) ? d.N / 2 : d.N);
        s.bytes_w = (int)pk::words_bytes(d.N, d.K);
        s.bytes_s = (int)pk::dscales_bytes(d.N, d.K, d.group_size);
        s.bytes_z = (int)pk::dzeros_bytes(d.N, d.K, d.group_size);
        s.block_begin = blocks;
        blocks += (pk::nt16(d.N) + rows - 1) / rows;
    }
    for (int i = count; i < TCE_MAX_GROUP; ++i) a.seg[i] = a.seg[0];
    hipError_t e = hipErrorInvalidConfiguration;
    if (rn) {  // residual add + the next RMSNorm: one decode row, one linear, groups of 128, N <= 16384 (the 2048 slots of the order)
        if (count != 1 || gamma || mb != 1 || gpu != 1 || d0.M != 1 || !(d0.flags & TCE_W4_ADD_TO_C) || d0.N > 16384 || d0.N % 8 != 0) return TCE_ERR_UNSUPPORTED_SHAPE;
        a.out_gamma = rn->gamma;
        a.out_eps = rn->eps;
        a.xn_out = static_cast<half_t *>(rn->xn_out);
        a.ws = static_cast<float *>(rn->workspace);
        if (uw == 8) e = z8 ? launch_i8<1, 1, 1, 8, true, 1024, false, true>(a, blocks, m_blocks, wk, stream) : launch_i8<1, 1, 1, 8, false, 1024, false, true>(a, blocks, m_blocks, wk, stream);
        else if (z8) e = launch_i8<1, 1, 1, 16, true, 1024, false, true>(a, blocks, m_blocks, wk, stream);
        else return TCE_ERR_UNSUPPORTED_SHAPE;  // K > 16384 with general zero points: the epilogue's registers on top of 16 KiB of weights per wave and the zero-point chain do not fit 128 (the form spilled; no instantiation spills: build.py NO_VGPR_SPILL)
        if (e != hipSuccess) {
            if (hip_err) *hip_err = e;
            return TCE_ERR_HIP;
        }
        return TCE_OK;
    }
    if (comb && comb->slots > 1) {  // the activation row = a deferred attention step's partial states (one decode row, one linear, groups of 128, K = heads x 128 in whole 1024-k waves)
        if (count != 1 || gamma || mb != 1 || gpu != 1 || uw != 8 || d0.M != 1 || d0.K != comb->heads * 128 || d0.K % 1024 != 0 || comb->slots > kAttnDeferMaxSlots || comb->stride != 132)
            return TCE_ERR_UNSUPPORTED_SHAPE;
        a.comb_part = comb->part;
        a.comb_pos_dev = comb_pos_dev;
        a.comb_pos = comb_pos;
        a.comb_slots = comb->slots;
        a.comb_chunk = comb->chunk;
        a.comb_stride = comb->stride;
        a.comb_heads = comb->heads;
        // (compiled for at most four waves -- K <= 4096, the 7B / 8B widths --: eight slots of partial states in flight beside the wave's weights need more registers than
        //  sixteen waves per CU leave)
        if (wk > 4) return TCE_ERR_UNSUPPORTED_SHAPE;
        if (comb->slots <= 4) e = z8 ? launch_i8<1, 1, 1, 8, true, 256, false, false, 4>(a, blocks, m_blocks, wk, stream) : launch_i8<1, 1, 1, 8, false, 256, false, false, 4>(a, blocks, m_blocks, wk, stream);
        else e = z8 ? launch_i8<1, 1, 1, 8, true, 256, false, false, 8>(a, blocks, m_blocks, wk, stream) : launch_i8<1, 1, 1, 8, false, 256, false, false, 8>(a, blocks, m_blocks, wk, stream);
        if (e != hipSuccess) {
            if (hip_err) *hip_err = e;
            return TCE_ERR_HIP;
        }
        return TCE_OK;
    }
    if (gamma) {
        if (mb != 1 || gpu != 1 || uw != 8) return TCE_ERR_UNSUPPORTED_SHAPE;
        if (rows == 2) e = z8 ? launch_i8<1, 1, 2, 8, true, 512, true>(a, blocks, m_blocks, wk, stream) : launch_i8<1, 1, 2, 8, false, 512, true>(a, blocks, m_blocks, wk, stream);
        else e = z8 ? launch_i8<1, 1, 1, 8, true, 1024, true>(a, blocks, m_blocks, wk, stream) : launch_i8<1, 1, 1, 8, false, 1024, true>(a, blocks, m_blocks, wk, stream);
        if (e != hipSuccess) {
            if (hip_err) *hip_err = e;
            return TCE_ERR_HIP;
        }
        return TCE_OK;
    }
#define TCE_I8(MB_, GPU_, ROWS_, UW_, MAXT_)                                                                                        \
    if (mb == MB_ && gpu == GPU_ && rows == ROWS_ && uw == UW_) {                                                                   \
        e = z8 ? launch_i8<MB_, GPU_, ROWS_, UW_, true, MAXT_>(a, blocks, m_blocks, wk, stream)                                     \
               : launch_i8<MB_, GPU_, ROWS_, UW_, false, MAXT_>(a, blocks, m_blocks, wk, stream);                                   \
    } else
    TCE_I8(1, 1, 1, 8, 1024)
    TCE_I8(1, 1, 2, 8, 512)
    TCE_I8(1, 1, 1, 16, 1024)
    TCE_I8(2, 1, 1, 8, 1024)
    TCE_I8(4, 1, 1, 8, 1024)
    TCE_I8(1, 2, 1, 8, 1024)
    TCE_I8(2, 2, 1, 8, 1024)
    TCE_I8(1, 4, 1, 8, 1024)
    return TCE_ERR_UNSUPPORTED_SHAPE;
#undef TCE_I8
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

// ---- the mixed launch: up to TCE_MAX_INDEPENDENT decode linears with their own activations and K as ONE launch (the kernel with ARGS = I8MixArgs) ----
static_assert(kI8MixMax == TCE_MAX_INDEPENDENT, "include/tce_matmul.h");

bool gemv_i8_mixed_supports(const tce_w4a16_desc *descs, int count) {
    if (g_i8_mode == 1 || count < 1 || count > kI8MixMax) return false;
    for (int i = 0; i < count; ++i) {
        const tce_w4a16_desc &d = descs[i];
        // one decode row, groups of 128, eight units per wave (K <= 16384), no prologue: the <1, 1, 1, 8> body
        if (d.M != 1 || d.group_size != 128 || d.K % 128 != 0 || d.K > 16384 || d.rmsnorm_gamma) return false;
        if (!d.prepacked || (reinterpret_cast<uintptr_t>(d.prepacked) & 255)) return false;
        if ((long long)pk::nt16(d.N) * (d.K / 2) * 16 >= (1LL << 31)) return false;
        if (d.flags & TCE_W4_FORCE_GEMM) return false;
    }
    return true;
}

// workgroups of the mixed launch and the waves of each (what tce_w4a16_forward_independent reports through describe; ONE rule with the launcher)
thread_local int g_i8_mix_waves = 0;  // tuning: waves per workgroup of the mixed launch forced (>= what the longest K needs; 0: the rule)
void set_gemv_i8_mixed_waves(int w) { g_i8_mix_waves = w >= 1 && w <= 16 ? w : 0; }

void gemv_i8_mixed_geometry(const tce_w4a16_desc *descs, int count, int *waves, int *workgroups) {
    // (a host issues the same block shapes launch after launch: the last answer is kept per thread)
    struct Memo {
        int count = 0, forced = -1, n[kI8MixMax], k[kI8MixMax], waves = 0, workgroups = 0;
    };
    thread_local Memo memo;
    bool hit = memo.count == count && memo.forced == g_i8_mix_waves;
    for (int i = 0; i < count && hit; ++i) hit = memo.n[i] == descs[i].N && memo.k[i] == descs[i].K;
    if (hit) {
        *waves = memo.waves;
        *workgroups = memo.workgroups;
        return;
    }
    int need = 1;
    for (int i = 0; i < count; ++i) need = need > (descs[i].K / 128 + 7) / 8 ? need : (descs[i].K / 128 + 7) / 8;
    auto blocks_at = [&](int wl) {
        int blocks = 0;
        for (int i = 0; i < count; ++i) {
            const int nsub = wl / ((descs[i].K / 128 + 7) / 8);
            blocks += (pk::nt16(descs[i].N) + nsub - 1) / nsub;
        }
        return blocks;
    };
    // The workgroup is at least as wide as the longest K needs and at most 16 waves.  What a width costs is the most loaded CU -- workgroups go to the 256 CUs in
    // index order, a workgroup's bytes are its live waves' 8 KiB each -- and, between widths that load it alike (within 5 %), the number of workgroups.  Fitted to
    // profiles/r6/mixed_block_waves_sweep.jsonl (a Llama-3-8B block's shards at 8 / 4 / 2 / 1 ranks: 14 waves 6.29 / 11.08 / 16.45 / 26.26 us, 16 waves 6.66 / 7.89 / 13.73 /
    // 24.71; Llama-2-13B: 14 waves 7.01 / 11.4 / 20.07 / 36.3, 15 waves 6.98 / 12.0 / 18.03 / 33.4): the rule picks the faster (or a tie) in all eight.
    // (workgroups b0 .. b0 + n - 1 of a linear go to CUs b0 % 256 ...: every CU gets n / 256 of them, the first n % 256 from b0 on one more; the linear's last workgroup may
    //  carry fewer tiles -- O(256) per linear, not O(tiles): this runs on the launch path)
    auto max_load = [&](int wl) {
        int load[256] = {0};
        int b0 = 0;
        for (int i = 0; i < count; ++i) {
            const int wk = (descs[i].K / 128 + 7) / 8, nsub = wl / wk, tiles = pk::nt16(descs[i].N);
            const int n = (tiles + nsub - 1) / nsub, units = nsub * wk, last_units = (tiles - (n - 1) * nsub) * wk;
            const int every = n / 256, extra = n % 256;
            for (int c = 0; c < 256; ++c) load[c] += every * units;
            for (int e = 0; e < extra; ++e) load[(b0 + every * 256 + e) & 255] += units;
            load[(b0 + n - 1) & 255] -= units - last_units;
            b0 += n;
        }
        int worst = 0;
        for (int c = 0; c < 256; ++c) worst = worst > load[c] ? worst : load[c];
        return worst;
    };
    int best_load = 1 << 30;
    for (int w = need; w <= 16; ++w) best_load = best_load < max_load(w) ? best_load : max_load(w);
    int wl = need;
    long best_blocks = 1L << 40;
    for (int w = need; w <= 16; ++w)
        if (max_load(w) * 20 <= best_load * 21 && blocks_at(w) < best_blocks) best_blocks = blocks_at(w), wl = w;
    if (g_i8_mix_waves >= need) wl = g_i8_mix_waves;
    *waves = wl;
    *workgroups = blocks_at(wl);
    memo.count = count;
    memo.forced = g_i8_mix_waves;
    for (int i = 0; i < count; ++i) memo.n[i] = descs[i].N, memo.k[i] = descs[i].K;
    memo.waves = wl;
    memo.workgroups = *workgroups;
}

int launch_w4a16_gemv_i8_mixed(const tce_w4a16_desc *descs, int count, hipStream_t stream, hipError_t *hip_err, const PeerGatherEpi *gather, int gathered) {
    if (!gemv_i8_mixed_supports(descs, count)) return TCE_ERR_UNSUPPORTED_SHAPE;
    I8MixArgs a{};
    a.nseg = count;
    a.M = 1;
    if (gather) {  // linear `gathered`'s slice is exchanged inside the launch: plain or residual-add store (the pair epilogue halves the row index), the slice is the whole linear
        if (gathered < 0 || gathered >= count) return TCE_ERR_BAD_ARG;
        const tce_w4a16_desc &dg = descs[gathered];
        if ((dg.flags & TCE_W4_SILU_MUL_PAIRS) || (unsigned)dg.N != gather->slice_elems) return TCE_ERR_UNSUPPORTED_SHAPE;
        a.g = *gather;
        a.g_seg = gathered;
        a.g_tiles = (unsigned)pk::nt16(dg.N);
    }
    int wl = 1, total = 0;
    gemv_i8_mixed_geometry(descs, count, &wl, &total);
    bool z8 = true;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const tce_w4a16_desc &d = descs[i];
        z8 = z8 && (d.flags & TCE_W4_ZERO_POINT_IS_8);
        I8MixSeg &m = a.s[i];
        m.A = static_cast<const half_t *>(d.A);
        m.lda = d.lda ? d.lda : d.K;
        m.K = d.K;
        I8Seg &s = m.seg;
        const unsigned char *base = static_cast<const unsigned char *>(d.prepacked);
        s.words = base;
        s.dscales = reinterpret_cast<const half_t *>(base + pk::dscales_offset(d.N, d.K, d.group_size));
        s.dzeros = reinterpret_cast<const unsigned *>(base + pk::dzeros_offset(d.N, d.K, d.group_size));
        s.C = static_cast<half_t *>(d.C);
        s.N = d.N;
        s.epilogue = d.flags & (TCE_W4_SILU_MUL_PAIRS | TCE_W4_ADD_TO_C);
        s.ldc = d.ldc ? d.ldc : ((s.epilogue & TCE_W4_SILU_MUL_PAIRS) ? d.N / 2 : d.N);
        s.bytes_w = (int)pk::words_bytes(d.N, d.K);
        s.bytes_s = (int)pk::dscales_bytes(d.N, d.K, d.group_size);
        s.bytes_z = (int)pk::dzeros_bytes(d.N, d.K, d.group_size);
        s.block_begin = blocks;
        const int nsub = wl / ((d.K / 128 + 7) / 8);
        blocks += (pk::nt16(d.N) + nsub - 1) / nsub;
    }
    for (int i = count; i < kI8MixMax; ++i) a.s[i] = a.s[0];
    const size_t lds = (size_t)wl * 8 * 512 + (size_t)wl * 16 * sizeof(float);  // the wave groups' regions side by side: never more than `wl` waves' worth
    auto launch = [&](auto kfn) -> hipError_t {
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kfn, dim3(blocks, 1, 1), dim3(64 * wl, 1, 1), lds, stream, a);
        return hipGetLastError();
    };
    const hipError_t e = z8 ? launch(w4a16_gemv_i8_kernel<1, 1, 1, 8, true, 1024, false, false, 0, I8MixArgs>) : launch(w4a16_gemv_i8_kernel<1, 1, 1, 8, false, 1024, false, false, 0, I8MixArgs>);
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
