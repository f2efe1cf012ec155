// w4a16_gemv_stream.hip -- the persistent ("streaming") form of the W4A16 decode GEMV for gfx950.
//
// Same math, layout and unpack/dot scheme as w4a16_gemv.hip (read its header first); what changes is the work
// distribution, because per-wave timestamps of that kernel (DESIGN.md "GEMV timeline", profiles/r1/timeline.jsonl)
// showed where a large launch loses its time:
//   * resident capacity is 4 waves/SIMD = 1024 four-wave workgroups; 1376 workgroups therefore run as a full first
//     generation and a 34 %-full second one that still costs a whole workgroup lifetime (~5 us);
//   * every workgroup re-stages the activation vector, and those x loads queue behind the weight loads already in the
//     CU's memory pipeline: median 1.7 us (p90 4.6 us) from wave start to the barrier.
// Here ONE workgroup per CU (up to 16 waves) stages x once, then each wave walks its own sequence of row groups
//     rg = gw, gw + W, gw + 2W, ...        (gw = global wave index, W = waves in the grid)
// with a ring of DEPTH units (one unit = ROWS rows x one 1-KiB step) always in flight, so loads, unpack and MFMAs of
// different units overlap inside a wave and across the 4 waves of a SIMD.  The host picks ROWS in {1,2} and the wave
// count per workgroup (<= 16) that make ceil(n_rg / W) * W closest to n_rg.  Units past the end of a wave's list are
// issued with an out-of-range buffer offset -- the hardware returns zeros without touching memory -- so the loop body
// has no divergent or conditional loads and every s_waitcnt the compiler places is an exact count.
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

namespace {

struct StreamArgs {
    const half_t *A;
    int lda, M, K, log2g;
    int nseg;
    int n_rg;  // total row groups over all segments
    unsigned long long *dbg;  // MODE 2 only
    GemvSeg seg[TCE_MAX_GROUP];  // block_begin = first row group of the segment
};

template <typename T>
__device__ __forceinline__ T pick4(int i, T a, T b, T c, T d) {
    return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d));
}

template <int MB, int ROWS, int DEPTH, int MODE = 0>
__global__ __launch_bounds__(1024) void w4a16_gemv_stream_kernel(const StreamArgs args) {
    unsigned long long ts0 = 0, ts1 = 0;
    if constexpr (MODE == 2) ts0 = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = nthreads >> 6;
    const int gw = blockIdx.x * NW + wave;  // global wave index
    const int W = gridDim.x * NW;

    const int K = args.K;
    const int nchunks = K >> 5;
    const int T = (nchunks + 63) >> 6;
    const int gshift = args.log2g - 5;
    const int m0 = blockIdx.y * MB;
    const int rowbytes = nchunks * 16;

    // ---- cursors (all wave-uniform) ----
    struct Cursor {
        int rg, t;
    };
    Cursor ci{gw, 0};  // next unit to issue

    struct Step {
        uint4_t w[ROWS];
        unsigned short s[ROWS];
        unsigned z[ROWS];
        int g;       // per lane: quantization group of the lane's chunk
        // wave-uniform bookkeeping of the unit (lives in SGPRs)
        int t;       // step within the row group
        bool live;   // false: a padding unit past the end of this wave's list (loads were answered with zeros)
        half_t *C;   // output pointer / extent of the unit's segment, needed when the row group completes
        int segN, ldc, row0;
    };
    // "Sticky" segment state: the descriptors of the linear the issue cursor is in.  A wave's row groups only move
    // forward, so this is re-read from the kernel arguments a handful of times per launch, not per unit (re-deriving it
    // per unit cost ~175 scalar instructions and two dozen branches per unit in the first version of this kernel).
    int cur_si = 0;
    int cur_end = args.nseg > 1 ? args.seg[1].block_begin : args.n_rg;  // first row group past the current segment
    const uint4_t *cur_qw = args.seg[0].qweight;
    const half_t *cur_sc = args.seg[0].scales;
    const unsigned *cur_zp = args.seg[0].zeros;
    half_t *cur_C = args.seg[0].C;
    int cur_N = args.seg[0].N, cur_ldc = args.seg[0].ldc, cur_sstr = args.seg[0].scales_stride * 2,
        cur_zstr = args.seg[0].zeros_stride * 4, cur_rgb = 0;
    auto issue = [&](Step &st) {
        const int rg = ci.rg, t = ci.t;
        const bool live = rg < args.n_rg;
        if (live && rg >= cur_end) {  // uniform, rare: advance to the segment that contains rg
            do {
                ++cur_si;
                cur_end = cur_si + 1 < args.nseg ? args.seg[cur_si + 1].block_begin : args.n_rg;
            } while (rg >= cur_end);
            const GemvSeg &sg = args.seg[cur_si];
            cur_qw = sg.qweight;
            cur_sc = sg.scales;
            cur_zp = sg.zeros;
            cur_C = sg.C;
            cur_N = sg.N;
            cur_ldc = sg.ldc;
            cur_sstr = sg.scales_stride * 2;
            cur_zstr = sg.zeros_stride * 4;
            cur_rgb = sg.block_begin;
        }
        // num_records is a constant just below 2 GiB: real extents were checked on the host, and a padding unit uses an
        // offset above it, which the buffer unit answers with zeros and no memory request
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4_t *>(cur_qw), 0, 0x7FFFFFF0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(cur_sc), 0, 0x7FFFFFF0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(cur_zp), 0, 0x7FFFFFF0, 0x00020000);
        const int c = t * 64 + lane;
        const int cc = c < nchunks ? c : nchunks - 1;  // tail lanes re-read the last chunk; their x image is zero
        const int g = cc >> gshift;
        st.g = g;
        st.t = t;
        st.live = live;
        st.C = cur_C;
        st.segN = cur_N;
        st.ldc = cur_ldc;
        const int row0 = live ? (rg - cur_rgb) * ROWS : 0;
        st.row0 = row0;
        const int oob = live ? 0 : (int)0x7FFFFFF0;
        const int vo_w = cc * 16 + oob, vo_s = g * 2 + oob, vo_z = (g >> 3) * 4 + oob;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            int r = row0 + i;
            r = r < cur_N ? r : cur_N - 1;  // clamped; the store is masked
            st.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo_w, r * rowbytes, /*nt*/ 2);
            st.s[i] = __builtin_amdgcn_raw_buffer_load_b16(rs_s, vo_s, r * cur_sstr, 0);
            st.z[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_z, vo_z, r * cur_zstr, 0);
        }
        // advance
        ci.t = t + 1;
        if (ci.t == T) {
            ci.t = 0;
            ci.rg = rg + W;
        }
    };

    // ---- prologue: the one-time x staging FIRST (with empty memory queues it takes ~1 us; issued behind the first
    // weight units its data cannot be consumed before theirs -- vmcnt retires in order -- and every wave idled 3.3-5 us,
    // profiles/r1/timeline_stream.jsonl), then the weight stream ----
    Step st[DEPTH];

    uint4_t *xs = reinterpret_cast<uint4_t *>(smem);  // [MB][T][4][64] pieces of 16 bytes (pair-permuted, lane-linear)
    {
        const int pieces_per_m = T * 256;
        const int total_pieces = MB * pieces_per_m;
        // one piece per thread per round; threads without a piece load nothing (clamping surplus threads onto piece 0,
        // as the row-block kernel does for its counted waits, made 3584 of 4096 threads hammer one L2 line here)
        for (int p = tid; p < total_pieces; p += nthreads) {
            const int m = MB == 1 ? 0 : p / pieces_per_m;
            const int r = p - m * pieces_per_m;
            const int c = (r >> 8) * 64 + (r & 63);
            const int j = (r >> 6) & 3;
            int mrow = m0 + m;
            mrow = mrow < args.M ? mrow : args.M - 1;
            uint4_t q = uint4_t{0u, 0u, 0u, 0u};
            if (c < nchunks) q = pair_permute(*reinterpret_cast<const uint4_t *>(args.A + (size_t)mrow * args.lda + (c * 32 + j * 8)));
            xs[p] = q;
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(st[d]);
    if constexpr (MODE == 2) ts1 = wall_clock64();

    float acc[ROWS][MB][4];
    float corr[ROWS][MB];
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            corr[i][m] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][m][r] = 0.f;
        }
    unsigned mask_hi;
    asm volatile("v_mov_b32 %0, 0x00F000F0" : "=v"(mask_hi));
    const unsigned magic = 0x64006400u;
    const half4_t ones = half4_t{(half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};
    float diag[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) diag[q] = (lane & 3) == q ? 0.0625f : 0.0f;

    auto compute = [&](const Step &st) {
        if constexpr (MODE == 1) {  // diagnostics: consume the loads, skip the math
#pragma unroll
            for (int i = 0; i < ROWS; ++i) acc[i][0][0] += (float)((st.w[i].x ^ st.w[i].y ^ st.w[i].z ^ st.w[i].w ^ st.s[i] ^ st.z[i]) & 0xFFu);
            if (st.t == T - 1 && st.live && lane == 63) st.C[st.row0] = (half_t)acc[0][0][0];
            return;
        }
        const int t = st.t;
        half4_t xb[MB][8];
        float xsum[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float4_t xs4 = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4_t xp = xs[((m * T + t) * 4 + j) * 64 + lane];
                xb[m][2 * j] = __builtin_bit_cast(half4_t, uint2_t{xp.x, xp.y});
                xb[m][2 * j + 1] = __builtin_bit_cast(half4_t, uint2_t{xp.z, xp.w});
                xs4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, xb[m][2 * j], xs4, 0, 0, 0);
                xs4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, xb[m][2 * j + 1], xs4, 0, 0, 0);
            }
            xsum[m] = xs4[0];
        }
        const int zsh = (st.g & 7) * 4;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            float4_t blk[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) blk[m] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned w = st.w[i][j];
                const unsigned t0 = ((w << 4) & mask_hi) | magic;
                const unsigned t1 = (w & mask_hi) | magic;
                const unsigned t2 = ((w >> 4) & mask_hi) | magic;
                const unsigned t3 = ((w >> 8) & mask_hi) | magic;
                const half4_t a0 = __builtin_bit_cast(half4_t, uint2_t{t0, t1});
                const half4_t a1 = __builtin_bit_cast(half4_t, uint2_t{t2, t3});
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    blk[m] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, xb[m][2 * j], blk[m], 0, 0, 0);
                    blk[m] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, xb[m][2 * j + 1], blk[m], 0, 0, 0);
                }
            }
            // a padding unit must contribute nothing even if an out-of-range load were not answered with zeros
            const float s = st.live ? (float)__builtin_bit_cast(half_t, st.s[i]) : 0.0f;
            const float cz = __builtin_fmaf((float)((st.z[i] >> zsh) & 0xFu), 16.0f, 1024.0f);
            const float scz = s * cz;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][m][r] = __builtin_fmaf(s, blk[m][r], acc[i][m][r]);
                corr[i][m] = __builtin_fmaf(scz, xsum[m], corr[i][m]);
            }
        }
        // ---- end of a row group: reduce over the 64 lanes, store, reset (wave-uniform branch) ----
        if (t == T - 1) {
            const bool live = st.live;
            half_t *Cp = st.C;
            const int segN = st.segN, ldc = st.ldc, row0 = st.row0;
#pragma unroll
            for (int i = 0; i < ROWS; ++i)
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    float v = corr[i][m] * -0.0625f;
                    v = __builtin_fmaf(acc[i][m][0], diag[0], v);
                    v = __builtin_fmaf(acc[i][m][1], diag[1], v);
                    v = __builtin_fmaf(acc[i][m][2], diag[2], v);
                    v = __builtin_fmaf(acc[i][m][3], diag[3], v);
                    v = wave_sum_dpp_lane63(v);  // total in lane 63
                    if (lane == 63 && live && row0 + i < segN && m0 + m < args.M) Cp[(size_t)(m0 + m) * ldc + row0 + i] = (half_t)v;
                    corr[i][m] = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][m][r] = 0.f;
                }
        }
    };

    // ---- the stream: every wave runs the same number of ring rounds; units past its list are zero-traffic padding ----
    const int iters = (args.n_rg + W - 1) / W;             // row groups per wave (last one may be padding)
    const int units = iters * T;
    for (int u = 0; u < units; u += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (u + d < units) compute(st[d]);  // uniform
            issue(st[d]);                       // past the end: padding (no memory traffic)
        }
    }
    if constexpr (MODE == 2) {
        if (lane == 63 && args.dbg) {
            unsigned long long *d = args.dbg + (size_t)gw * 4;
            d[0] = ts0; d[1] = ts1; d[2] = wall_clock64(); d[3] = d[2];
        }
    }
}

template <int MB, int ROWS, int DEPTH, int MODE = 0>
hipError_t launch_stream(const StreamArgs &a, int blocks, int nw, int m_blocks, hipStream_t stream) {
    const int nchunks = a.K >> 5;
    const int T = (nchunks + 63) / 64;
    const size_t lds = (size_t)MB * T * 4096 + (size_t)64 * nw * 16;
    auto kfn = w4a16_gemv_stream_kernel<MB, ROWS, DEPTH, MODE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(blocks, m_blocks, 1), dim3(64 * nw, 1, 1), lds, stream, a);
    return hipGetLastError();
}

int g_num_cus = 0;
int g_stream_rows = 0, g_stream_nw = 0, g_stream_depth = 0;  // forced geometry (0 = automatic)
int g_stream_mode = 0;
unsigned long long *g_stream_dbg = nullptr;

}  // namespace

void set_gemv_stream_debug(int mode, void *buf) {
    g_stream_mode = mode;
    g_stream_dbg = static_cast<unsigned long long *>(buf);
}

void set_gemv_stream_config(int rows, int nw, int depth) {
    g_stream_rows = rows;
    g_stream_nw = nw;
    g_stream_depth = depth;
}

// Returns TCE_ERR_UNSUPPORTED_SHAPE when the persistent form does not apply (the caller then uses the workgroup-per-
// row-block kernel of w4a16_gemv.hip).
int launch_w4a16_gemv_stream(const tce_w4a16_desc *descs, int count, hipStream_t stream, hipError_t *hip_err) {
    const tce_w4a16_desc &d0 = descs[0];
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return TCE_ERR_HIP;
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    long total_n = 0;
    for (int i = 0; i < count; ++i) total_n += descs[i].N;
    const int nchunks = d0.K >> 5;
    const int T = (nchunks + 63) / 64;
    const int mb = d0.M >= 4 ? 4 : (d0.M >= 2 ? 2 : 1);
    if (mb == 4) return TCE_ERR_UNSUPPORTED_SHAPE;  // 4 activation rows per pass do not fit 128 VGPRs here: row-block kernel

    // ---- geometry: ROWS in {1,2}, nw waves per workgroup (one workgroup per CU), DEPTH units in flight ----
    int rows = g_stream_rows, nw = g_stream_nw, depth = g_stream_depth;
    if (rows == 0) {
        double best = -1.0;
        for (int r = 2; r >= 1; --r) {
            long n_rg = 0;
            for (int i = 0; i < count; ++i) n_rg += (descs[i].N + r - 1) / r;
            for (int w = 16; w >= 8; --w) {
                const long Wt = (long)g_num_cus * w;
                const long it = (n_rg + Wt - 1) / Wt;
                // balance x a mild preference for more waves (latency hiding) and for 2 rows per unit (x reuse)
                const double eff = (double)n_rg / (double)(Wt * it) * (0.90 + 0.10 * w / 16.0) * (r == 2 ? 1.0 : 0.97);
                if (eff > best) {
                    best = eff;
                    rows = r;
                    nw = w;
                }
            }
        }
    }
    if (depth == 0 || (mb * rows >= 4 && depth == 3)) depth = (mb * rows >= 4) ? 2 : 3;
    if (rows != 1 && rows != 2) return TCE_ERR_BAD_ARG;
    if (nw < 1 || nw > 16 || depth < 2 || depth > 3) return TCE_ERR_BAD_ARG;

    StreamArgs a{};
    a.A = static_cast<const half_t *>(d0.A);
    a.lda = d0.lda ? d0.lda : d0.K;
    a.M = d0.M;
    a.K = d0.K;
    a.log2g = d0.group_size == 128 ? 7 : (d0.group_size == 64 ? 6 : 5);
    a.nseg = count;
    int n_rg = 0;
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
        const long long bw = (long long)d.N * (d.K / 2), bs = (long long)d.N * s.scales_stride * 2, bz = (long long)d.N * s.zeros_stride * 4;
        if (bw >= 0x7FFFFFF0LL || bs >= 0x7FFFFFF0LL || bz >= 0x7FFFFFF0LL) return TCE_ERR_UNSUPPORTED_SHAPE;
        s.block_begin = n_rg;
        n_rg += (d.N + rows - 1) / rows;
    }
    for (int i = count; i < TCE_MAX_GROUP; ++i) a.seg[i] = a.seg[0];
    a.n_rg = n_rg;
    a.dbg = g_stream_dbg;
    const size_t lds = (size_t)mb * T * 4096 + (size_t)64 * nw * 16;
    if (lds > 160 * 1024) return TCE_ERR_UNSUPPORTED_SHAPE;
    int blocks = g_num_cus;
    if ((long)blocks * nw > n_rg) blocks = (n_rg + nw - 1) / nw;  // fewer row groups than waves
    const int m_blocks = (d0.M + mb - 1) / mb;

    hipError_t e = hipSuccess;
    bool found = true;
    if (mb == 1 && g_stream_mode != 0) {
        if (rows == 2 && depth == 2) e = g_stream_mode == 1 ? launch_stream<1, 2, 2, 1>(a, blocks, nw, m_blocks, stream) : launch_stream<1, 2, 2, 2>(a, blocks, nw, m_blocks, stream);
        else if (rows == 2 && depth == 3) e = g_stream_mode == 1 ? launch_stream<1, 2, 3, 1>(a, blocks, nw, m_blocks, stream) : launch_stream<1, 2, 3, 2>(a, blocks, nw, m_blocks, stream);
        else return TCE_ERR_BAD_ARG;
        if (e != hipSuccess) { if (hip_err) *hip_err = e; return TCE_ERR_HIP; }
        return TCE_OK;
    }
#define TCE_S(MB_, R_, D_) \
    if (mb == MB_ && rows == R_ && depth == D_) e = launch_stream<MB_, R_, D_>(a, blocks, nw, m_blocks, stream); else
    TCE_S(1, 1, 2) TCE_S(1, 1, 3) TCE_S(1, 2, 2) TCE_S(1, 2, 3) TCE_S(2, 1, 2) TCE_S(2, 1, 3) TCE_S(2, 2, 2)
    found = false;
#undef TCE_S
    if (!found) return TCE_ERR_BAD_ARG;
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
