// w4a16_gemv.hip -- W4A16 dequant-GEMV for gfx950 (MI355X), M <= a few rows: the per-token decode kernel.
//
// Replaces gemv_kernel_g128 / gemv_kernel_g64 (reference kernels/cuda/gemv_cuda.cu:140-194, 68-123) behind
// MatmulOperator::gemv_forward_cuda.  Same data layout (q4_6), same math
//     C[m][n] = fp16( sum_k fp32(s[n][k/G]) * (q[n][k] - z[n][k/G]) * fp32(A[m][k]) )      (fp32 accumulate)
// designed for CDNA4 rather than translated.  Measurements (profiles/) showed the first version of this kernel was
// bound by VALU ISSUE, not by HBM: a wave64 VALU instruction occupies its SIMD for 4 cycles, which caps a CU at one
// vector instruction per cycle, i.e. ~85 instructions per KiB of weights at 6.5 TB/s.  So the design minimises vector
// instructions per weight and moves everything it can to other pipes:
//
//   * STREAM   one wavefront streams ROWS weight rows at a time; lane l of step t owns the 16-byte chunk
//              c = t*64*WK + wk*64 + l of each row (32 int4 weights), so every buffer_load_dwordx4 of a wave covers
//              1 KiB of consecutive HBM bytes.  Loads are non-temporal (each byte is read once by one CU), addressed
//              through buffer descriptors (row offset in an SGPR, chunk offset in one VGPR shared by all rows: no
//              per-load 64-bit address arithmetic), and DEPTH steps are kept in flight per wave.
//   * X        the activation vector is staged once per workgroup into LDS, permuted into the pair order the int4
//              extraction yields, in a lane-linear image (every ds_read_b128 is bank-conflict free); one read
//              feeds ROWS rows.
//   * UNPACK   7 VALU instructions per 8 weights: each nibble pair is moved to mantissa bits 4..7 of the half
//              1024.0 (3 shifts + 4 v_and_or_b32 with the mask in a VGPR -- gfx9 VOP3 reads one scalar only), giving the
//              halves 1024 + 16q.  The zero point is NOT removed per element:
//                  sum_k (q_k - z) x_k = ( sum_k (1024 + 16 q_k) x_k  -  (1024 + 16 z) * sum_k x_k ) / 16
//              where sum_k x_k per 32-weight chunk is row independent (computed once per step, on the matrix pipe).
//              The bias is only ~14x the signal at this nibble position, so the fp32 accumulation noise it adds is
//              ~2e-6 of the output rms (DESIGN.md "GEMV numerics"), three orders below the 1e-3 budget.
//   * DOT      on the otherwise idle matrix pipe: v_mfma_f32_4x4x4_16b_f16 computes, per group of 4 lanes,
//              D[i][j] = sum_k A_i[k] B_j[k] over the 4 halves each lane supplies; with A = a lane's weights and
//              B = the same lane's activations the DIAGONAL D[l%4][l%4] is that lane's own dot product, accumulated
//              in fp32 across calls (the off-diagonal cross terms are discarded at the end with a one-hot multiply).
//   * SCALE    the fp16 group scale is applied in fp32 once per 32 weights (4 fma for the accumulator, 2 for the
//              zero-point term).
//   * REDUCE   64-lane shuffle tree (+ an LDS hop when WK waves split K).
//   * GROUP    up to TCE_MAX_GROUP linears that share the activation (q/k/v, gate/up) run as ONE launch.
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

#include <type_traits>

namespace tce {

namespace {

// MODE: 0 = the GEMV.  3 = the GEMV with x staged before the weight stream starts.  2 = 0 + per-wave timestamps.  1 = diagnostics: stream the weights only (no unpack / dot) -- roofline experiments, output unused.
// The second __launch_bounds__ argument (minimum waves per SIMD) caps the register allocation: left alone, hipcc hoists
// every row's unpack ahead of the MFMAs and spends > 256 VGPRs, i.e. one wave per SIMD and nothing to hide HBM latency.
// NORM: the activation staging applies generalT5LayerNorm (LlamaRMSNorm.cu:68-93) on the fly -- the workgroup reads the
// un-normalised hidden state, sums its squares, and writes half(clamp((x * rs) * gamma)) into the x image (MB = 1).
template <int MB, int ROWS, int WN, int WK, int DEPTH, int XB, int MODE = 0, bool Z8 = false, bool NORM = false>
__global__ __launch_bounds__(64 * WN * WK, (MB * ROWS * DEPTH >= 8 ? 2 : (MB * ROWS * DEPTH >= 4 ? (MB == 1 && ROWS == 4 && DEPTH == 1 ? 5 : 3) : 4))) void w4a16_gemv_kernel(const GemvArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NTHREADS = 64 * WN * WK;
    constexpr int LS = 64 * WK;  // lanes that split K
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if constexpr (MODE == 2) ts0 = wall_clock64();  // 100 MHz
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % WK;
    const int wn = wave / WK;

    const int K = args.K;
    const int nchunks = K >> 5;             // 16-byte chunks per weight row
    const int T = (nchunks + LS - 1) / LS;  // steps
    const int gshift = args.log2g - 5;      // chunk -> group
    const int m0 = blockIdx.y * MB;

    // ---- which linear of the group does this workgroup belong to? (wave-uniform) ----
    int si = 0;
#pragma unroll
    for (int s = 1; s < TCE_MAX_GROUP; ++s)
        if (s < args.nseg && (int)blockIdx.x >= args.seg[s].block_begin) si = s;
    const GemvSeg seg = args.seg[si];
    const int row_base = (((int)blockIdx.x - seg.block_begin) * WN + wn) * ROWS;

    uint4_t *xs = reinterpret_cast<uint4_t *>(smem);  // [MB][T][WK][4][64] pieces of 16 bytes
    const int pieces_per_m = T * LS * 4;
    const int total_pieces = MB * pieces_per_m;

    // ---- activation staging, part 1: issue the first batch of x loads (addresses clamped, never predicated) ----
    const half_t *A = args.A;
    auto x_src = [&](int p, bool &valid) -> const uint4_t * {
        // LDS piece index p -> (m, t*WK + wk', j, l): the image is lane-linear for the readers
        const int pc = p < total_pieces ? p : 0;
        const int m = MB == 1 ? 0 : pc / pieces_per_m;
        const int r = pc - m * pieces_per_m;
        const int c = (r >> 8) * 64 + (r & 63);
        const int j = (r >> 6) & 3;
        int mrow = m0 + m;
        mrow = mrow < args.M ? mrow : args.M - 1;
        valid = (p < total_pieces) && (c < nchunks);
        const int cc = c < nchunks ? c : 0;
        return reinterpret_cast<const uint4_t *>(A + (size_t)mrow * args.lda + (cc * 32 + j * 8));
    };
    uint4_t xv[XB];
    bool xok[XB];
#pragma unroll
    for (int i = 0; i < XB; ++i) xv[i] = *x_src(tid + i * NTHREADS, xok[i]);
    // fused RMSNorm prologue: gamma of the same 8 columns, requested now (their latency overlaps the weight prologue below)
    float4_t gv[NORM ? XB : 1][2];
    if constexpr (NORM) {
#pragma unroll
        for (int i = 0; i < XB; ++i) {
            const float *gp = args.gamma + (reinterpret_cast<const half_t *>(x_src(tid + i * NTHREADS, xok[i])) - A);  // same column offset as the x piece (M = 1)
            gv[i][0] = *reinterpret_cast<const float4_t *>(gp);
            gv[i][1] = *reinterpret_cast<const float4_t *>(gp + 4);
        }
    }

    // ---- weight stream: buffer descriptors + scalar row offsets ----
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4_t *>(seg.qweight), 0, seg.bytes_w, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(seg.scales), 0, seg.bytes_s, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(seg.zeros), 0, seg.bytes_z, 0x00020000);
    int so_w[ROWS], so_s[ROWS], so_z[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        int r = row_base + i;
        r = r < seg.N ? r : seg.N - 1;  // clamped; stores are masked
        so_w[i] = r * (nchunks * 16);
        so_s[i] = r * (seg.scales_stride * 2);
        so_z[i] = r * (seg.zeros_stride * 4);
    }
    struct Step {
        uint4_t w[ROWS];
        unsigned short s[ROWS];
        unsigned z[ROWS];
        int g;
    };
    auto issue = [&](Step &st, int t) {
        const int c = t * LS + wk * 64 + lane;
        const int cc = c < nchunks ? c : nchunks - 1;  // tail lanes re-read the last chunk; their x image is zero
        const int g = cc >> gshift;
        st.g = g;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            if constexpr (MODE == 4) {  // diagnostics: no weight traffic, the unpack/MFMA work runs on synthetic registers
                const unsigned v = (unsigned)(cc * 16 + so_w[i]) * 2654435761u;
                st.w[i] = uint4_t{v, v ^ 0x9E3779B9u, v * 3u, v * 5u};
                st.s[i] = (unsigned short)0x2000;
                st.z[i] = 0x88888888u;
                asm volatile("" : "+v"(st.w[i]));
                continue;
            }
            st.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, cc * 16, so_w[i], /*nt*/ 2);
            if constexpr (MODE == 1) {
                st.s[i] = 0x2000;
                st.z[i] = 0x88888888u;
            } else {
                st.s[i] = __builtin_amdgcn_raw_buffer_load_b16(rs_s, g * 2, so_s[i], 0);
                // Z8: the caller vouches (TCE_W4_ZERO_POINT_IS_8) that every zero point is 8 -- what the reference
                // quantizer always writes (quantize_methods.py:436-440) -- so the packed zeros are not streamed at all
                if constexpr (Z8) st.z[i] = 0x88888888u;
                else st.z[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_z, (g >> 3) * 4, so_z[i], 0);
            }
        }
    };
    // Prologue: DEPTH steps issued unconditionally (the host only picks variants with DEPTH <= T), so every wait the
    // compiler places is an exact counted vmcnt.
    Step st[DEPTH];
    constexpr bool XFIRST = MODE == 3;  // stage x completely before the first weight load is issued (see launch_variant)
    if constexpr (!XFIRST) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(st[d], d);
    }
    // pin the issue order: x loads, then the weight stream, and only then the first use of x -- so the wait in front
    // of the LDS writes is a counted vmcnt that leaves every weight load in flight
    __builtin_amdgcn_sched_barrier(0);

    // ---- activation staging, part 2: permute + write.  Batch 0 is straight-line code (counted waits); XB is chosen by
    // the host so that one batch covers the whole image whenever a compiled XB is large enough. ----
    auto x_write = [&](int base) {
#pragma unroll
        for (int i = 0; i < XB; ++i) {
            const int p = base + tid + i * NTHREADS;
            uint4_t v = pair_permute(xv[i]);
            if (!xok[i]) v = uint4_t{0u, 0u, 0u, 0u};
            // unconditional store (surplus pieces land in a per-thread trash slot behind the image): a predicated
            // store makes hipcc sink the matching load into the branch and drain vmcnt(0) there
            xs[p < total_pieces ? p : total_pieces + tid] = v;
        }
    };
    if constexpr (NORM) {
        static_assert(MB == 1, "the fused RMSNorm prologue is a decode (M = 1) feature");
        // rs in the shape-independent order of tce_common.hpp (same bits as tce_rmsnorm_half and as the persistent kernel's prologue), formed
        // from the pieces this thread ALREADY holds: piece sums -> the 1024 slots (4 KiB in the trash-slot area behind the image) -> the
        // fixed lane / DPP tree.  Image piece p' is row piece lin(p') = 4 * ((p' >> 8) * 64 + (p' & 63)) + ((p' >> 6) & 3), a permutation
        // inside every run of 256, so thread tid's pieces p' = tid + i * NTHREADS fall into slots lin(tid + r * NTHREADS), r = i mod R,
        // in increasing row order -- and the threads' R slots each cover all 1024 exactly once.
        static_assert(NTHREADS >= 256 && 1024 % NTHREADS == 0, "the trash-slot area must hold 16 x 64 floats; slots are dealt out per thread");
        constexpr int R = 1024 / NTHREADS;
        float *part = reinterpret_cast<float *>(smem + (size_t)total_pieces * 16);
        float sl[R];
#pragma unroll
        for (int r = 0; r < R; ++r) sl[r] = 0.f;
#pragma unroll
        for (int i = 0; i < XB; ++i) sl[i % R] += xok[i] ? rmsnorm_piece_sum(__builtin_bit_cast(half8_t, xv[i])) : 0.f;
        for (int base = XB * NTHREADS, b = 1; base < total_pieces; base += XB * NTHREADS, ++b) {  // long K only: the rest of the row
#pragma unroll
            for (int i = 0; i < XB; ++i) {
                bool ok;
                const uint4_t v = *x_src(base + tid + i * NTHREADS, ok);
                const float sp = ok ? rmsnorm_piece_sum(__builtin_bit_cast(half8_t, v)) : 0.f;
                const int r = (XB * b + i) & (R - 1);
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
                    if (rr == r) sl[rr] += sp;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int pp = tid + r * NTHREADS;
            part[(((pp >> 8) * 64 + (pp & 63)) << 2) + ((pp >> 6) & 3)] = sl[r];
        }
        const float rs = rmsnorm_slots_finish(part, K, args.eps, lane);
        // normalise, permute, write the image: the first batch from registers, the rest of a long row re-read (L1 / L2 hits)
        auto norm_piece = [&](uint4_t raw, const float4_t &g0, const float4_t &g1) -> uint4_t {
            const half8_t v = __builtin_bit_cast(half8_t, raw);
            half8_t y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = rmsnorm_out(v[e], rs, g0[e]);
                y[4 + e] = rmsnorm_out(v[4 + e], rs, g1[e]);
            }
            return pair_permute(__builtin_bit_cast(uint4_t, y));
        };
#pragma unroll
        for (int i = 0; i < XB; ++i) {
            const int p = tid + i * NTHREADS;
            uint4_t o = norm_piece(xv[i], gv[i][0], gv[i][1]);
            if (!xok[i]) o = uint4_t{0u, 0u, 0u, 0u};
            xs[p < total_pieces ? p : total_pieces + tid] = o;
        }
        for (int p = XB * NTHREADS + tid; p < total_pieces; p += NTHREADS) {
            bool ok;
            const uint4_t *src = x_src(p, ok);
            const float *gp = args.gamma + (reinterpret_cast<const half_t *>(src) - A);
            uint4_t o = norm_piece(*src, *reinterpret_cast<const float4_t *>(gp), *reinterpret_cast<const float4_t *>(gp + 4));
            if (!ok) o = uint4_t{0u, 0u, 0u, 0u};
            xs[p] = o;
        }
    } else {
        x_write(0);
        for (int base = XB * NTHREADS; base < total_pieces; base += XB * NTHREADS) {  // only for long K and/or MB > 1
#pragma unroll
            for (int i = 0; i < XB; ++i) xv[i] = *x_src(base + tid + i * NTHREADS, xok[i]);
            x_write(base);
        }
    }
    lds_barrier();  // (publishes the x image; the weight loads issued above stay in flight -- a __syncthreads would wait for all of them)
    if constexpr (XFIRST) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(st[d], d);
    }
    if constexpr (MODE == 2) ts1 = wall_clock64();
    // The sum of a chunk's 32 activations (the zero-point term's factor) does not depend on the row: all WN waves along N used to form it for every
    // step with 8 MFMAs (A = ones) -- a third of the launch's MFMAs at two rows per wave.  With `shared_xsum` wave wn forms it for the steps
    // t = wn, wn + WN, ... only (the same 8 MFMAs in the same order: the same bits), the table goes through LDS behind the trash slots.
    float *sxt = reinterpret_cast<float *>(smem + (size_t)total_pieces * 16 + (size_t)NTHREADS * 16);  // [MB][T][WK][64]
    if (args.shared_xsum) {
        const half4_t ones_t = half4_t{(half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};
        for (int t = wn; t < T; t += WN) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float4_t xs4 = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint4_t xp = xs[(((m * T + t) * WK + wk) * 4 + j) * 64 + lane];
                    xs4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones_t, __builtin_bit_cast(half4_t, uint2_t{xp.x, xp.y}), xs4, 0, 0, 0);
                    xs4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones_t, __builtin_bit_cast(half4_t, uint2_t{xp.z, xp.w}), xs4, 0, 0, 0);
                }
                sxt[((m * T + t) * WK + wk) * 64 + lane] = xs4[0];
            }
        }
        lds_barrier();
    }

    float acc[ROWS][MB][4];  // the 4 accumulator registers of the 4x4x4 MFMA; the lane's own dot product is [lane & 3]
    float corr[ROWS][MB];    // sum over chunks of s * (1024 + 16 z) * sum_k x_k
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            corr[i][m] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][m][r] = 0.f;
        }
    // nibble mask in a VGPR (see tce_common.hpp: one scalar operand per VOP3 on gfx9)
    unsigned mask_hi;
    asm volatile("v_mov_b32 %0, 0x00F000F0" : "=v"(mask_hi));
    const unsigned magic = 0x64006400u;  // (1024.0h, 1024.0h)
    const half4_t ones = half4_t{(half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};

    auto compute = [&](const Step &st, int t) {
        if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < ROWS; ++i) acc[i][0][0] += (float)((st.w[i].x ^ st.w[i].y ^ st.w[i].z ^ st.w[i].w) & 0xFFu);
            return;
        }
        // activations of this lane's chunk, as MFMA B operands, and their sum (A = ones) on the matrix pipe
        half4_t xb[MB][8];
        float xsum[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4_t xp = xs[(((m * T + t) * WK + wk) * 4 + j) * 64 + lane];
                xb[m][2 * j] = __builtin_bit_cast(half4_t, uint2_t{xp.x, xp.y});      // (x0,x4,x1,x5) of word j
                xb[m][2 * j + 1] = __builtin_bit_cast(half4_t, uint2_t{xp.z, xp.w});  // (x2,x6,x3,x7)
            }
            if (args.shared_xsum) {  // wave-uniform
                xsum[m] = sxt[((m * T + t) * WK + wk) * 64 + lane];
            } else {
                float4_t xs4 = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xs4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, xb[m][2 * j], xs4, 0, 0, 0);
                    xs4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, xb[m][2 * j + 1], xs4, 0, 0, 0);
                }
                xsum[m] = xs4[0];  // D[i][j] = sum_k B_j[k] for every i: all four registers hold this lane's sum
            }
        }
        const int zsh = (st.g & 7) * 4;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            // keep each row's unpack -> MFMA -> scale sequence together: without the fence the scheduler hoists all
            // rows' unpacks ahead of the first MFMA and the register allocation explodes
            __builtin_amdgcn_sched_barrier(0);
            float4_t blk[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) blk[m] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned w = st.w[i][j];
                const unsigned t0 = ((w << 4) & mask_hi) | magic;  // (1024+16 q0, 1024+16 q4)
                const unsigned t1 = (w & mask_hi) | magic;         // (q1, q5)
                const unsigned t2 = ((w >> 4) & mask_hi) | magic;  // (q2, q6)
                const unsigned t3 = ((w >> 8) & mask_hi) | magic;  // (q3, q7)
                const half4_t a0 = __builtin_bit_cast(half4_t, uint2_t{t0, t1});
                const half4_t a1 = __builtin_bit_cast(half4_t, uint2_t{t2, t3});
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    blk[m] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, xb[m][2 * j], blk[m], 0, 0, 0);
                    blk[m] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, xb[m][2 * j + 1], blk[m], 0, 0, 0);
                }
            }
            // lanes 4b..4b+3 share a quantization group (32 weights per lane, groups of >= 32), so scaling all four
            // accumulator registers by this lane's scale is consistent; the diagonal is picked once, at the end
            const float s = (float)__builtin_bit_cast(half_t, st.s[i]);
            const float cz = Z8 ? 1152.0f : __builtin_fmaf((float)((st.z[i] >> zsh) & 0xFu), 16.0f, 1024.0f);  // 1024 + 16 z
            const float scz = s * cz;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][m][r] = __builtin_fmaf(s, blk[m][r], acc[i][m][r]);
                corr[i][m] = __builtin_fmaf(scz, xsum[m], corr[i][m]);
            }
        }
    };

    // Steady state, unrolled by DEPTH so that ring slots are compile-time registers (a register-to-register
    // rotation would force a wait on the in-flight loads): consume step t from slot t % DEPTH, refill it with step
    // t + DEPTH.  DEPTH-1 steps ((DEPTH-1)*ROWS 16-byte loads per lane + their scales/zeros) stay in flight.
    const int S = T - DEPTH;  // steps that still have a refill behind them
    int t = 0;
    for (; t + DEPTH <= S; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            compute(st[d], t + d);
            issue(st[d], t + d + DEPTH);
        }
    }
    // r = S - t leftover refilling steps (0 <= r < DEPTH), then the DEPTH-step drain; t is a multiple of DEPTH here
    auto tail = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            compute(st[i], t + i);
            issue(st[i], t + i + DEPTH);
        }
#pragma unroll
        for (int i = R; i < R + DEPTH; ++i) compute(st[i % DEPTH], t + i);
    };
    const int r = S - t;
    if (r == 0) tail(std::integral_constant<int, 0>{});
    if constexpr (DEPTH > 1) { if (r == 1) tail(std::integral_constant<int, 1>{}); }
    if constexpr (DEPTH > 2) { if (r == 2) tail(std::integral_constant<int, 2>{}); }
    static_assert(DEPTH <= 3, "add tail cases");

    if constexpr (MODE == 2) ts2 = wall_clock64();
    // ---- K reduction: pick the lane's diagonal accumulator, 64-lane shuffle tree, then across the WK waves via LDS ----
    float diag[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) diag[q] = (lane & 3) == q ? 0.0625f : 0.0f;  // one-hot pick and the final /16 in one factor
    float red[ROWS][MB];
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            // diagonal pick as multiply-adds with a one-hot lane mask: a select chain on (lane & 3) is turned by hipcc
            // into a dynamically indexed private array, i.e. scratch memory traffic (off-diagonal terms are finite)
            float v = corr[i][m] * -0.0625f;
            v = __builtin_fmaf(acc[i][m][0], diag[0], v);
            v = __builtin_fmaf(acc[i][m][1], diag[1], v);
            v = __builtin_fmaf(acc[i][m][2], diag[2], v);
            v = __builtin_fmaf(acc[i][m][3], diag[3], v);
            if constexpr (MODE == 1) v = acc[i][m][0];
            red[i][m] = wave_sum_dpp_lane63(v);  // total in lane 63
        }

    if constexpr (WK > 1) {
        __syncthreads();  // everyone is done reading the x image; reuse the front of LDS
        float *redbuf = reinterpret_cast<float *>(smem);  // [WN][WK][ROWS*MB]
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < ROWS; ++i)
#pragma unroll
                for (int m = 0; m < MB; ++m) redbuf[((wn * WK + wk) * ROWS + i) * MB + m] = red[i][m];
        }
        __syncthreads();
        if (wk != 0) return;
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float v = 0.f;
#pragma unroll
                for (int k2 = 0; k2 < WK; ++k2) v += redbuf[((wn * WK + k2) * ROWS + i) * MB + m];
                red[i][m] = v;
            }
    }

    if constexpr (MODE == 2) {
        if (lane == 63 && args.dbg) {
            unsigned long long *d = args.dbg + ((size_t)blockIdx.x * (WN * WK) + wave) * 4;
            d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = wall_clock64();
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m0 + m >= args.M) continue;
            half_t *crow = seg.C + (size_t)(m0 + m) * seg.ldc;
            if (seg.epilogue & TCE_W4_SILU_MUL_PAIRS) {  // rows (2n, 2n+1) = (gate n, up n); the host picked an even ROWS
                if constexpr (ROWS % 2 == 0) {
#pragma unroll
                    for (int i = 0; i < ROWS; i += 2)
                        if (row_base + i + 1 < seg.N) crow[(row_base + i) >> 1] = silu_mul_half((half_t)red[i][m], (half_t)red[i + 1][m]);
                }
            } else if (seg.epilogue & TCE_W4_ADD_TO_C) {
#pragma unroll
                for (int i = 0; i < ROWS; ++i)
                    if (row_base + i < seg.N) crow[row_base + i] = crow[row_base + i] + (half_t)red[i][m];
            } else {
#pragma unroll
                for (int i = 0; i < ROWS; ++i)
                    if (row_base + i < seg.N) crow[row_base + i] = (half_t)red[i][m];
            }
        }
    }
}

struct Variant {
    int rows, wn, wk, depth;
};

thread_local int g_debug_mode = 0;
thread_local int g_order_force = 0;  // tuning (tce_w4a16_set_debug_mode 10 / 11 / 12): 0 the rule in launch_variant, 1 x first always, 2 weights first always
unsigned long long *g_debug_buf = nullptr;

template <int MB, int ROWS, int WN, int WK, int DEPTH, int XB, int MODE = 0, bool Z8 = false, bool NORM = false>
hipError_t launch_one(const GemvArgs &a, int total_blocks, int m_blocks, hipStream_t stream) {
    const int nchunks = a.K >> 5;
    const int LS = 64 * WK;
    const int T = (nchunks + LS - 1) / LS;
    size_t lds = (size_t)MB * T * LS * 64 + (size_t)64 * WN * WK * 16 + (size_t)MB * T * LS * sizeof(float);  // x image + trash slots + the shared activation-sum table
    const size_t red = (size_t)WN * WK * ROWS * MB * sizeof(float);
    if (lds < red) lds = red;
    auto kfn = w4a16_gemv_kernel<MB, ROWS, WN, WK, DEPTH, XB, MODE, Z8, NORM>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(total_blocks, m_blocks, 1), dim3(64 * WN * WK, 1, 1), lds, stream, a);
    return hipGetLastError();
}

template <int MB, int ROWS, int WN, int WK, int DEPTH>
hipError_t launch_variant(const GemvArgs &a, int total_blocks, int m_blocks, hipStream_t stream) {
    // x pieces per thread: MB * T * 4 / WN; one straight-line batch when it fits a compiled batch size
    const int nchunks = a.K >> 5;
    const int T = (nchunks + 64 * WK - 1) / (64 * WK);
    const int need = (MB * T * 4 + WN - 1) / WN;
    // register budget: the widest batches run with a shallower pipeline instead of spilling
    if constexpr (MB * ROWS * DEPTH > 16 && DEPTH > 1) {
        return launch_variant<MB, ROWS, WN, WK, DEPTH - 1>(a, total_blocks, m_blocks, stream);
    } else {
        if constexpr (MB == 1) {
            if (a.gamma) {
                if constexpr ((64 * WN * WK) < 256 || 1024 % (64 * WN * WK) != 0) {
                    return hipErrorInvalidConfiguration;  // (the fused prologue's slot scheme needs 256 / 512 / 1024 threads; the dispatcher never asks for it here)
                } else {
                // like the plain launches below: "x first" when the whole grid is one generation of workgroups -- round 2 found the fused
                // prologue 2.6 us slower than the plain launch for exactly this reason (its x loads queued behind every wave's weights)
                const int wps = (ROWS == 4 && DEPTH == 1) ? 5 : (ROWS * DEPTH <= 2 ? 8 : 4);
                if (total_blocks <= 256 * (4 * wps / (WN * WK)))
                    return a.zeros_are_8 ? launch_one<1, ROWS, WN, WK, DEPTH, 2, 3, true, true>(a, total_blocks, m_blocks, stream)
                                         : launch_one<1, ROWS, WN, WK, DEPTH, 2, 3, false, true>(a, total_blocks, m_blocks, stream);
                return a.zeros_are_8 ? launch_one<1, ROWS, WN, WK, DEPTH, 2, 0, true, true>(a, total_blocks, m_blocks, stream)
                                     : launch_one<1, ROWS, WN, WK, DEPTH, 2, 0, false, true>(a, total_blocks, m_blocks, stream);
                }
            }
#ifdef TCE_LAB  // (diagnostic instantiations -- stream only / timestamps / arithmetic only --: the lab build, build.py --lab)
            if (g_debug_mode == 1) return launch_one<1, ROWS, WN, WK, DEPTH, 8, 1>(a, total_blocks, m_blocks, stream);
            if (g_debug_mode == 2) return launch_one<1, ROWS, WN, WK, DEPTH, 2, 2>(a, total_blocks, m_blocks, stream);
            if (g_debug_mode == 4) return launch_one<1, ROWS, WN, WK, DEPTH, 2, 4>(a, total_blocks, m_blocks, stream);
#endif
            if (g_debug_mode == 3) return launch_one<1, ROWS, WN, WK, DEPTH, 2, 3>(a, total_blocks, m_blocks, stream);
        }
        // MODE 3 = "x first": the workgroup stages x completely before it issues its first weight load.  When the whole
        // grid is resident at once (one generation of workgroups) this keeps the x loads from queueing behind HBM-bound
        // weight loads in the CU's miss queue (median wave start -> barrier 2.5 us otherwise, profiles/r1/timeline*.jsonl);
        // with several generations the late workgroups queue anyway and the delayed weight issue only costs.
        if constexpr (MB == 1) {
            const int waves_per_simd = (ROWS == 4 && DEPTH == 1) ? 5 : (ROWS * DEPTH <= 2 ? 8 : 4);
            const int capacity = 256 * (4 * waves_per_simd / (WN * WK));
            if (g_debug_mode == 0 && (g_order_force == 1 || (g_order_force == 0 && total_blocks <= capacity))) {
                if (a.zeros_are_8) {
                    if (need <= 2) return launch_one<MB, ROWS, WN, WK, DEPTH, 2, 3, true>(a, total_blocks, m_blocks, stream);
                    if (need <= 4) return launch_one<MB, ROWS, WN, WK, DEPTH, 4, 3, true>(a, total_blocks, m_blocks, stream);
                    return launch_one<MB, ROWS, WN, WK, DEPTH, 8, 3, true>(a, total_blocks, m_blocks, stream);
                }
                if (need <= 2) return launch_one<MB, ROWS, WN, WK, DEPTH, 2, 3>(a, total_blocks, m_blocks, stream);
                if (need <= 4) return launch_one<MB, ROWS, WN, WK, DEPTH, 4, 3>(a, total_blocks, m_blocks, stream);
                return launch_one<MB, ROWS, WN, WK, DEPTH, 8, 3>(a, total_blocks, m_blocks, stream);
            }
            if (g_debug_mode == 0 && a.zeros_are_8) {
                if (need <= 2) return launch_one<MB, ROWS, WN, WK, DEPTH, 2, 0, true>(a, total_blocks, m_blocks, stream);
                if (need <= 4) return launch_one<MB, ROWS, WN, WK, DEPTH, 4, 0, true>(a, total_blocks, m_blocks, stream);
                return launch_one<MB, ROWS, WN, WK, DEPTH, 8, 0, true>(a, total_blocks, m_blocks, stream);
            }
        }
        if (need <= 2) return launch_one<MB, ROWS, WN, WK, DEPTH, 2>(a, total_blocks, m_blocks, stream);
        if (need <= 4) return launch_one<MB, ROWS, WN, WK, DEPTH, 4>(a, total_blocks, m_blocks, stream);
        return launch_one<MB, ROWS, WN, WK, DEPTH, 8>(a, total_blocks, m_blocks, stream);
    }
}

template <int MB>
hipError_t launch_mb(const Variant &v, const GemvArgs &a, int total_blocks, int m_blocks, hipStream_t s, bool &found) {
    found = true;
#define TCE_V(R, N_, K_, D_) \
    if (v.rows == R && v.wn == N_ && v.wk == K_ && v.depth == D_) \
        return launch_variant<MB, R, N_, K_, D_>(a, total_blocks, m_blocks, s);
    TCE_GEMV_VARIANTS(TCE_V)
#undef TCE_V
    found = false;
    return hipSuccess;
}

}  // namespace

void set_gemv_debug_mode(int mode) { g_debug_mode = mode; }
void set_gemv_order(int force) { g_order_force = force >= 0 && force <= 2 ? force : 0; }
// 0: the rule -- ON for a decode launch (M = 1): 2-3 % on each of the token's four launch shapes (7.31 -> 7.09, 4.22 -> 4.09, 10.79 -> 10.47, 7.73 -> 7.60 us,
// profiles/r3/gemv_shared_xsum_ab.jsonl; the round's first A/B of this switch compared a kernel with itself -- its debug mode was shadowed -- and is withdrawn);
// 1 on, 2 off
thread_local int g_shared_xsum = 0;
void set_gemv_shared_xsum(int on) { g_shared_xsum = on == 1 || on == 2 ? on : 0; }
void set_gemv_debug_buffer(void *p) { g_debug_buf = static_cast<unsigned long long *>(p); }

bool gemv_variant_exists(int rows, int wn, int wk, int depth) {
#define TCE_V(R, N_, K_, D_) \
    if (rows == R && wn == N_ && wk == K_ && depth == D_) return true;
    TCE_GEMV_VARIANTS(TCE_V)
#undef TCE_V
    return false;
}

// Host-side launch: fills GemvArgs, picks MB and the geometry, launches.  `forced_*` may be 0.
int launch_w4a16_gemv(const tce_w4a16_desc *descs, int count, int forced_rows, int forced_wn, int forced_wk,
                      int forced_depth, hipStream_t stream, hipError_t *hip_err, const float *gamma, float eps) {
    const tce_w4a16_desc &d0 = descs[0];
    if (!gamma && d0.rmsnorm_gamma) {
        gamma = static_cast<const float *>(d0.rmsnorm_gamma);
        eps = d0.rmsnorm_eps;
    }
    if (gamma && d0.M != 1) return TCE_ERR_UNSUPPORTED_SHAPE;
    GemvArgs a{};
    a.gamma = gamma;
    a.eps = eps;
    a.A = static_cast<const half_t *>(d0.A);
    a.lda = d0.lda ? d0.lda : d0.K;
    a.M = d0.M;
    a.K = d0.K;
    a.log2g = d0.group_size == 128 ? 7 : (d0.group_size == 64 ? 6 : 5);
    a.nseg = count;
    a.dbg = g_debug_buf;
    a.zeros_are_8 = 1;
    for (int i = 0; i < count; ++i)
        if (!(descs[i].flags & TCE_W4_ZERO_POINT_IS_8)) a.zeros_are_8 = 0;
    a.shared_xsum = g_shared_xsum == 1 || (g_shared_xsum == 0 && d0.M == 1);

    int total_n = 0;
    for (int i = 0; i < count; ++i) total_n += descs[i].N;
    Variant v{forced_rows, forced_wn, forced_wk, forced_depth};
    const int nchunks = d0.K >> 5;
    if (v.rows == 0) {
        // Enough workgroups to cover 256 CUs a few times; split K across waves only when there are too few rows to
        // fill the chip and K is long enough.  (Tuned on MI355X; see DESIGN.md "GEMV geometry".)
        // Re-fitted in round 2 (scripts/tune.py --only gemv with the zero-point-8 kernels the bench and the adapter run,
        // profiles/r2/gemv_geometry_sweep.jsonl; run-to-run noise is ~3 %): two rows per wave instead of four between 8k and 24k rows
        // (gate+up 2 x 11008: 10.9-11.3 vs 11.7 us; qkv 12288: a tie), four rows and FOUR waves per workgroup from 24k rows up (lm_head
        // 32000: 14.0-14.2 vs 14.6 us with eight waves; 128256: 45.5-46.1 us, level with the persistent kernel's 46.7), and few rows
        // over a long K split K four ways (512 x 11008: 3.9 vs 4.3 us).
        // A/B in one process (scripts/gemv_ab.py, profiles/r2/gemv_geometry_ab.jsonl; best / median of 7 rounds): gate+up 22016 rows:
        // (2 rows, 4 waves, depth 2) 11.35 / 11.40 us, (2, 4, 1) 11.48 / 12.68 (bimodal), (4, 4, 1) 11.94 / 12.10; qkv 12288 rows: (4, 4, 1)
        // 7.02 / 7.09, (2, 4, 1) 7.16 / 7.17; Llama-3 gate+up 28672 rows: (4, 4, 1) 13.18, (2, 4, 2) 13.64.
        // Round 3: three rows per wave (compiled: X(3, 4, 1, 1 / 2), forceable) were A/B'd on the grouped gate+up launches and NOT taken: on one box
        // (3, 4, 2) ran 11.18 / 11.23 us against (2, 4, 2) 11.47 / 11.71, on the next 10.73 / 10.80 against 10.34 / 10.50 (profiles/r3/gemv_rows3_ab.jsonl).
        if (total_n >= 24000) v = {4, 4, 1, 1};
        else if (total_n >= 16384) v = {2, 4, 1, 2};
        else if (total_n >= 8192) v = {4, 4, 1, 1};
        else if (total_n >= 3072) v = nchunks >= 256 && d0.M == 1 ? Variant{2, 8, 1, 2} : Variant{2, 4, 1, 2};  // long K (down_proj): 7.5 vs 7.8 us, 8.2 vs 8.5
        else if (total_n >= 1536) v = {1, 4, 1, 2};
        else if (nchunks >= 256 && d0.M == 1) v = {1, 2, 4, 1};
        else if (nchunks >= 128) v = {1, 2, 2, 1};
        else v = {1, 4, 1, 1};
    }
    bool pairs = false;
    for (int i = 0; i < count; ++i) pairs = pairs || (descs[i].flags & TCE_W4_SILU_MUL_PAIRS);
    if (pairs && (v.rows & 1)) {  // a (gate, up) row pair must sit in one wave: the 2-row sibling of the chosen geometry
        v.rows = 2;
        if (v.wk == 4) v.wk = 2;
        if (!gemv_variant_exists(v.rows, v.wn, v.wk, v.depth)) v.depth = 1;
        if (!gemv_variant_exists(v.rows, v.wn, v.wk, v.depth)) v = {2, 4, 1, 1};
    }
    {
        // the kernel's prologue needs DEPTH <= T; shrink the pipeline for short K
        const int T = (nchunks + 64 * v.wk - 1) / (64 * v.wk);
        while (v.depth > 1 && (v.depth > T || !gemv_variant_exists(v.rows, v.wn, v.wk, v.depth))) --v.depth;
        if (v.depth > T) return TCE_ERR_UNSUPPORTED_SHAPE;
    }
    const int rows_per_block = v.rows * v.wn;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const tce_w4a16_desc &d = descs[i];
        GemvSeg &s = a.seg[i];
        const int zw = zeros_width(d.K, d.group_size);
        s.qweight = static_cast<const uint4_t *>(d.qweight);
        s.scales = static_cast<const half_t *>(d.scales);
        s.zeros = static_cast<const unsigned *>(d.zeros);
        s.C = static_cast<half_t *>(d.C);
        s.N = d.N;
        s.epilogue = d.flags & (TCE_W4_SILU_MUL_PAIRS | TCE_W4_ADD_TO_C);
        s.ldc = d.ldc ? d.ldc : ((s.epilogue & TCE_W4_SILU_MUL_PAIRS) ? d.N / 2 : d.N);
        s.scales_stride = d.scales_stride ? d.scales_stride : zw * 8;
        s.zeros_stride = d.zeros_stride ? d.zeros_stride : zw;
        // buffer descriptors address 32-bit byte offsets
        const long long bw = (long long)d.N * (d.K / 2), bs = (long long)d.N * s.scales_stride * 2, bz = (long long)d.N * s.zeros_stride * 4;
        if (bw >= (1LL << 31) || bs >= (1LL << 31) || bz >= (1LL << 31)) return TCE_ERR_UNSUPPORTED_SHAPE;
        s.bytes_w = (int)bw;
        s.bytes_s = (int)bs;
        s.bytes_z = (int)bz;
        s.block_begin = blocks;
        blocks += (d.N + rows_per_block - 1) / rows_per_block;
    }
    for (int i = count; i < TCE_MAX_GROUP; ++i) a.seg[i] = a.seg[0];

    // M rows are processed MB at a time by gridDim.y; MB in {1,2,4}
    int mb = d0.M >= 4 ? 4 : (d0.M >= 2 ? 2 : 1);
    {
        // the x image of mb rows must fit the CU's 160 KiB of LDS (launch_one's formula): fewer rows per pass for a long K,
        // and a clear refusal -- not a generic HIP launch error -- when even one row does not fit (K > ~80k)
        const int LS = 64 * v.wk, T = (nchunks + LS - 1) / LS;
        auto need = [&](int m) { return (size_t)m * T * LS * 64 + (size_t)64 * v.wn * v.wk * 16 + (size_t)m * T * LS * 4; };  // (launch_one's formula: x image, trash slots, sum table)
        while (mb > 1 && need(mb) > 160 * 1024) mb >>= 1;
        if (need(mb) > 160 * 1024) return TCE_ERR_UNSUPPORTED_SHAPE;
    }
    const int m_blocks = (d0.M + mb - 1) / mb;
    bool found = false;
    hipError_t e;
    if (mb == 4) e = launch_mb<4>(v, a, blocks, m_blocks, stream, found);
    else if (mb == 2) e = launch_mb<2>(v, a, blocks, m_blocks, stream, found);
    else e = launch_mb<1>(v, a, blocks, m_blocks, stream, found);
    if (!found) return TCE_ERR_BAD_ARG;
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
