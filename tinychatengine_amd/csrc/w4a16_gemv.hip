// w4a16_gemv.hip -- bandwidth-bound W4A16 dequant-GEMV for gfx950 (MI355X), M <= a few rows.
//
// Replaces gemv_kernel_g128 / gemv_kernel_g64 (reference kernels/cuda/gemv_cuda.cu:140-194, 68-123) behind
// MatmulOperator::gemv_forward_cuda.  Same data layout (q4_6), same math
//     C[m][n] = fp16( sum_k fp32(s[n][k/G]) * (q[n][k] - z[n][k/G]) * fp32(A[m][k]) )      (fp32 accumulate)
// but designed for CDNA4 rather than translated:
//   * one 64-lane wavefront streams ROWS weight rows at a time; lane l of step t owns the 16-byte chunk
//     c = t*64*WK + wk*64 + l of each row (32 int4 weights), so every global_load_dwordx4 of a wave covers
//     1 KiB of consecutive HBM bytes; loads are non-temporal (each byte is read once by one CU) and
//     2*ROWS of them are in flight per lane (double-buffered steps) to cover HBM latency;
//   * the activation vector is staged once per workgroup into LDS, pre-permuted into the pair order that the
//     magic-number int4->fp16 conversion yields (tce_common.hpp), in a lane-linear image so every ds_read_b128
//     is bank-conflict free; one read feeds ROWS rows;
//   * (q - z) is formed exactly in packed fp16, multiplied with the activations by v_dot2c_f32_f16 (fp32
//     accumulate), and the fp16 group scale is applied once per 32 weights in fp32;
//   * the K reduction is a 64-lane shuffle tree (+ an LDS hop when WK waves split K);
//   * up to TCE_MAX_GROUP linears that share the activation (q/k/v, gate/up) run as ONE launch.
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

#include <type_traits>

namespace tce {

namespace {

// MODE is a diagnostics switch for roofline experiments (scripts/tune.py); only MODE 0 computes the GEMV.
//   1: stream only (weights loaded, no dequant / dot)   2: no scale / zero-point loads   3: plain instead of non-temporal loads
//   4: dot products on the VALU (v_dot2c_f32_f16) instead of the matrix pipe -- a correct GEMV, kept for A/B measurements
template <int MB, int ROWS, int WN, int WK, int DEPTH, int MODE = 0>
__global__ __launch_bounds__(64 * WN * WK) void w4a16_gemv_kernel(const GemvArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NTHREADS = 64 * WN * WK;
    constexpr int LS = 64 * WK;  // lanes that split K
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % WK;
    const int wn = wave / WK;

    const int K = args.K;
    const int nchunks = K >> 5;                // 16-byte chunks per weight row
    const int T = (nchunks + LS - 1) / LS;     // steps
    const int gshift = args.log2g - 5;         // chunk -> group
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
    constexpr int XB = 8;  // pieces per thread per batch
    const half_t *A = args.A;
    auto x_src = [&](int p, bool &valid) -> const uint4_t * {
        // LDS piece index p -> (m, t, wk', j, l): image is lane-linear for the readers
        const int pc = p < total_pieces ? p : 0;
        const int m = MB == 1 ? 0 : pc / pieces_per_m;
        const int r = pc - m * pieces_per_m;
        const int l = r & 63;
        const int j = (r >> 6) & 3;
        const int tw = r >> 8;  // t*WK + wk'
        const int c = tw * 64 + l;
        const int mrow = (m0 + m) < args.M ? (m0 + m) : (args.M - 1);
        valid = (p < total_pieces) && (c < nchunks);
        const int cc = c < nchunks ? c : 0;
        return reinterpret_cast<const uint4_t *>(A + (size_t)mrow * args.lda + (size_t)cc * 32 + j * 8);
    };
    uint4_t xv[XB];
    bool xok[XB];
#pragma unroll
    for (int i = 0; i < XB; ++i) xv[i] = *x_src(tid + i * NTHREADS, xok[i]);

    // ---- weight stream ----
    int rows[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int r = row_base + i;
        rows[i] = r < seg.N ? r : seg.N - 1;  // clamped; stores are masked
    }
    struct Step {
        uint4_t w[ROWS];
        half_t s[ROWS];
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
            const uint4_t *wp = seg.qweight + (size_t)rows[i] * nchunks + cc;
            if constexpr (MODE == 3) st.w[i] = *wp;
            else st.w[i] = load_nt(wp);
            if constexpr (MODE == 1 || MODE == 2) {
                st.s[i] = (half_t)0.01f;
                st.z[i] = 0x88888888u;
            } else {
                st.s[i] = seg.scales[(size_t)rows[i] * seg.scales_stride + g];
                st.z[i] = seg.zeros[(size_t)rows[i] * seg.zeros_stride + (g >> 3)];
            }
        }
    };
    // Prologue: DEPTH steps issued unconditionally (the host only picks variants with DEPTH <= T), so every wait
    // the compiler places is an exact counted vmcnt.
    Step st[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(st[d], d);
    // pin the issue order: x loads, then the weight stream, and only then the first use of x -- so the wait in
    // front of the LDS writes is a counted vmcnt that leaves every weight load in flight
    __builtin_amdgcn_sched_barrier(0);

    // ---- activation staging, part 2: permute + write.  Batch 0 is straight-line code so that its wait is a
    // counted vmcnt covering only the x loads (they were issued first; the weight loads stay in flight). ----
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
    x_write(0);
    for (int base = XB * NTHREADS; base < total_pieces; base += XB * NTHREADS) {  // only for long K and/or MB > 1
#pragma unroll
        for (int i = 0; i < XB; ++i) xv[i] = *x_src(base + tid + i * NTHREADS, xok[i]);
        x_write(base);
    }
    __syncthreads();

    // Dot products.  MODE 4 uses v_dot2c_f32_f16 on the VALU.  The default puts them on the otherwise idle matrix
    // pipe: v_mfma_f32_4x4x4_16b_f16 computes, for each group of 4 lanes, D[i][j] = sum_k A_i[k] * B_j[k] over the 4
    // halves each lane supplies; with A = a lane's dequantized weights and B = the same lane's activations, the
    // DIAGONAL element D[l%4][l%4] is exactly that lane's own 4-term dot product, accumulated in fp32 across calls
    // (the 12 off-diagonal cross terms per group are discarded).  That removes 4 of the 13 VALU instructions per 8
    // weights -- measured: the kernel was VALU-issue-bound, not HBM-bound (profiles/).
    constexpr bool kMfma = (MODE != 4);
    constexpr int ACCW = kMfma ? 4 : 1;
    float acc[ROWS][MB][ACCW];
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < ACCW; ++r) acc[i][m][r] = 0.f;
    const NibbleMasks nmask = make_nibble_masks();

    auto compute = [&](const Step &st, int t) {
        if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < ROWS; ++i) acc[i][0][0] += (float)((st.w[i].x ^ st.w[i].y ^ st.w[i].z ^ st.w[i].w) & 0xFFu);
            return;
        }
        uint4_t x[MB][4];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[m][j] = xs[(((m * T + t) * WK + wk) * 4 + j) * 64 + lane];
        const int zsh = (st.g & 7) * 4;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const ZeroPair zp = make_zero_pair((st.z[i] >> zsh) & 0xFu);
            const float s = (float)st.s[i];
            if constexpr (kMfma) {
                float4_t blk[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) blk[m] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    half2_t d[4];
                    dequant_word(st.w[i][j], zp, nmask, d);
                    const half4_t a0 = __builtin_bit_cast(half4_t, uint2_t{as_u32(d[0]), as_u32(d[1])});
                    const half4_t a1 = __builtin_bit_cast(half4_t, uint2_t{as_u32(d[2]), as_u32(d[3])});
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const half4_t b0 = __builtin_bit_cast(half4_t, uint2_t{x[m][j].x, x[m][j].y});
                        const half4_t b1 = __builtin_bit_cast(half4_t, uint2_t{x[m][j].z, x[m][j].w});
                        blk[m] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, b0, blk[m], 0, 0, 0);
                        blk[m] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, b1, blk[m], 0, 0, 0);
                    }
                }
                // lanes 4b..4b+3 share a quantization group (32 weights per lane, groups of >= 32), so scaling all four
                // accumulator registers by this lane's scale is consistent; the diagonal is picked once, at the end
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][m][r] = __builtin_fmaf(s, blk[m][r], acc[i][m][r]);
            } else {
                float p[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) p[m] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    half2_t d[4];
                    dequant_word(st.w[i][j], zp, nmask, d);
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        p[m] = __builtin_amdgcn_fdot2(d[0], as_half2(x[m][j].x), p[m], false);
                        p[m] = __builtin_amdgcn_fdot2(d[1], as_half2(x[m][j].y), p[m], false);
                        p[m] = __builtin_amdgcn_fdot2(d[2], as_half2(x[m][j].z), p[m], false);
                        p[m] = __builtin_amdgcn_fdot2(d[3], as_half2(x[m][j].w), p[m], false);
                    }
                }
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[i][m][0] = __builtin_fmaf(s, p[m], acc[i][m][0]);
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

    // ---- K reduction: pick the lane's diagonal accumulator, 64-lane shuffle tree, then across the WK waves via LDS ----
    float diag[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) diag[r] = (lane & 3) == r ? 1.0f : 0.0f;  // off-diagonal terms are finite, so x0 is exact
    float red[ROWS][MB];
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float v = acc[i][m][0];
            if constexpr (kMfma) {
                // diagonal pick as a multiply-add with a one-hot lane mask: a select chain on (lane & 3) is turned by
                // hipcc into a dynamically indexed private array, i.e. scratch memory traffic
                v = acc[i][m][0] * diag[0];
                v = __builtin_fmaf(acc[i][m][1], diag[1], v);
                v = __builtin_fmaf(acc[i][m][2], diag[2], v);
                v = __builtin_fmaf(acc[i][m][3], diag[3], v);
            }
            red[i][m] = wave_sum(v);
        }

    if constexpr (WK > 1) {
        __syncthreads();  // everyone is done reading the x image; reuse the front of LDS
        float *redbuf = reinterpret_cast<float *>(smem);  // [WN][WK][ROWS*MB]
        if (lane == 0) {
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

    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m0 + m >= args.M) continue;
#pragma unroll
            for (int i = 0; i < ROWS; ++i)
                if (row_base + i < seg.N) seg.C[(size_t)(m0 + m) * seg.ldc + row_base + i] = (half_t)red[i][m];
        }
    }
}

struct Variant {
    int rows, wn, wk, depth;
};

int g_debug_mode = 0;

template <int MB, int ROWS, int WN, int WK, int DEPTH, int MODE = 0>
hipError_t launch_variant(const GemvArgs &a, int total_blocks, int m_blocks, hipStream_t stream) {
    const int nchunks = a.K >> 5;
    const int LS = 64 * WK;
    const int T = (nchunks + LS - 1) / LS;
    size_t lds = (size_t)MB * T * LS * 64 + (size_t)64 * WN * WK * 16;  // x image + trash slots
    const size_t red = (size_t)WN * WK * ROWS * MB * sizeof(float);
    if (lds < red) lds = red;
    auto kfn = w4a16_gemv_kernel<MB, ROWS, WN, WK, DEPTH, MODE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(total_blocks, m_blocks, 1), dim3(64 * WN * WK, 1, 1), lds, stream, a);
    return hipGetLastError();
}

template <int MB>
hipError_t launch_mb(const Variant &v, const GemvArgs &a, int total_blocks, int m_blocks, hipStream_t s, bool &found) {
    found = true;
    if constexpr (MB == 1) {
        if (g_debug_mode != 0) {  // diagnostics builds exist for three geometries only
#define TCE_DBG(R, N_, K_, D_)                                                                                   \
    if (v.rows == R && v.wn == N_ && v.wk == K_ && v.depth == D_) {                                              \
        if (g_debug_mode == 1) return launch_variant<1, R, N_, K_, D_, 1>(a, total_blocks, m_blocks, s);          \
        if (g_debug_mode == 2) return launch_variant<1, R, N_, K_, D_, 2>(a, total_blocks, m_blocks, s);          \
        if (g_debug_mode == 3) return launch_variant<1, R, N_, K_, D_, 3>(a, total_blocks, m_blocks, s);          \
        if (g_debug_mode == 4) return launch_variant<1, R, N_, K_, D_, 4>(a, total_blocks, m_blocks, s);          \
    }
            TCE_DBG(4, 4, 1, 1)
            TCE_DBG(4, 4, 1, 2)
            TCE_DBG(2, 4, 1, 2)
#undef TCE_DBG
            found = false;
            return hipSuccess;
        }
    }
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

bool gemv_variant_exists(int rows, int wn, int wk, int depth) {
#define TCE_V(R, N_, K_, D_) \
    if (rows == R && wn == N_ && wk == K_ && depth == D_) return true;
    TCE_GEMV_VARIANTS(TCE_V)
#undef TCE_V
    return false;
}

// Host-side launch: fills GemvArgs, picks MB and the geometry, launches.  `forced` may be {0,0,0}.
int launch_w4a16_gemv(const tce_w4a16_desc *descs, int count, int forced_rows, int forced_wn, int forced_wk,
                      int forced_depth, hipStream_t stream, hipError_t *hip_err) {
    const tce_w4a16_desc &d0 = descs[0];
    GemvArgs a{};
    a.A = static_cast<const half_t *>(d0.A);
    a.lda = d0.lda ? d0.lda : d0.K;
    a.M = d0.M;
    a.K = d0.K;
    a.log2g = d0.group_size == 128 ? 7 : (d0.group_size == 64 ? 6 : 5);
    a.nseg = count;

    // ---- geometry: enough workgroups to cover 256 CUs several times, enough bytes in flight per CU ----
    int total_n = 0, min_n = 1 << 30;
    for (int i = 0; i < count; ++i) {
        total_n += descs[i].N;
        if (descs[i].N < min_n) min_n = descs[i].N;
    }
    Variant v{forced_rows, forced_wn, forced_wk, forced_depth};
    const int nchunks = d0.K >> 5;
    if (v.rows == 0) {
        // Enough workgroups to cover 256 CUs a few times; split K across waves only when there are too few rows to
        // fill the chip and K is long enough.  (Tuned on MI355X; see DESIGN.md "GEMV geometry".)
        if (total_n >= 8192) v = {4, 4, 1, 2};
        else if (total_n >= 3072) v = {2, 4, 1, 2};
        else if (total_n >= 1536) v = {1, 4, 1, 2};
        else if (nchunks >= 128) v = {1, 2, 2, 1};
        else v = {1, 4, 1, 1};
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
        s.ldc = d.ldc ? d.ldc : d.N;
        s.scales_stride = d.scales_stride ? d.scales_stride : zw * 8;
        s.zeros_stride = d.zeros_stride ? d.zeros_stride : zw;
        s.block_begin = blocks;
        blocks += (d.N + rows_per_block - 1) / rows_per_block;
    }
    for (int i = count; i < TCE_MAX_GROUP; ++i) a.seg[i] = a.seg[0];

    // M rows are processed MB at a time by gridDim.y; MB in {1,2,4}
    const int mb = d0.M >= 4 ? 4 : (d0.M >= 2 ? 2 : 1);
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
