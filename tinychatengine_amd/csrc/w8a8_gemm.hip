// w8a8_gemm.hip -- SmoothQuant W8A8 int8 GEMM on MFMA for gfx950, bit-exact with kernels/ref/matmul_ref_int8.cc.
//
// One kernel serves the eight int8 MatmulOperator methods (kernels/ref/matmul_ref_int8.cc:161-192); they differ only
// in the epilogue:
//     acc   = sum_k (int32)A[m][k] * (int32)B[n][k]                                  exact, v_mfma_i32_16x16x64_i8
//     int8 out:  r = (int32) round( (float)acc * alpha [+ (float)bias_i8[n] * beta] ); clamp(q_min,q_max)   (:29-32,:54-57)
//     fp32 out:  (float)acc * alpha [+ bias_f32[n]]                                                     (:108,:132)
// Bit-exactness rules: int->float conversion is RNE (v_cvt_f32_i32), the two products and the sum are
// each rounded separately (this file is compiled with -ffp-contract=off AND uses __fmul_rn/__fadd_rn), round() is
// half-away-from-zero, the clamp happens on the rounded value before narrowing.
//
// Layout: A int8 [M][K], B int8 [N][K] -- both K-contiguous, which is the MFMA i8 fragment shape: lane l feeds 16
// consecutive k of row (l & 15), k-block (l >> 4).  (Any assignment of k to MFMA slots is legal as long as A and B agree,
// so no swizzle of the DATA is needed.)  The MFMA kernel below nevertheless loads row-major 64-byte row pieces (four
// lanes per row: full cache lines) and transposes them to fragment order through a per-wave 1 KiB LDS slot; the
// direct fragment-shaped load (16 bytes per lane, 16 rows x 4 k-blocks apart) touched 16 cache lines per request.
#include "tce_common.hpp"

namespace tce {

namespace {

struct W8A8Args {
    const int8_t *A;
    const int8_t *B;
    const void *bias;
    void *C;
    long long strideA, strideB, strideC;
    int M, N, K;
    int lda, ldb, ldc;  // row strides in elements (>= K, K, N): a head's slice of a [rows][heads * 64] matrix is an operand as it lies
    float alpha, beta;
    int q_min, q_max;
    int bias_kind, out_kind, b_per_row, vec_ok;
    int accumulate;     // fp32 output only: C = fadd_rn(C, result) -- the residual add behind out_proj / fc2 (Int8OPTDecoderLayer.cc:39, 54)
    // round 5: a tile's k-steps cut into `xs` runs on as many WORKGROUPS (few tiles with a long chain: 512 x 768 x 3072 is 96 tiles of 48 steps, each bound by what ONE CU
    // pulls through its L1).  int32 partial tiles are exact: written through to `xpart`, the workgroup that arrives last at the tile's counter adds the others' in any
    // order and runs the epilogue -- the same integers as the unsplit kernel, bit-exact.  xcnt is zero between launches.
    int xs;
    int4_t *xpart;      // [tiles][xs][4 accumulator tiles][256 threads]
    unsigned *xcnt;     // [tiles]
};

// The additive term of output column n: fmul_rn(bias[n], beta) for the int8 form, bias[n] for the fp32 form, none.
__device__ __forceinline__ float bias_term(const W8A8Args &a, int n) {
    if (a.bias_kind == TCE_BIAS_INT8) return __fmul_rn((float)static_cast<const int8_t *>(a.bias)[n], a.beta);
    if (a.bias_kind == TCE_BIAS_FP32) return static_cast<const float *>(a.bias)[n];
    return 0.0f;
}

// the int8 output of one accumulator (`u` = bias_term of its column)
__device__ __forceinline__ int8_t epilogue_i8(const W8A8Args &a, int acc, float u) {
    const float f = (float)acc;  // v_cvt_f32_i32: round-to-nearest-even, like the host cast
    float v = __fmul_rn(f, a.alpha);
    if (a.bias_kind == TCE_BIAS_INT8) v = __fadd_rn(v, u);
    float r = roundf(v);  // half away from zero (std::round)
    // the reference narrows to int32 first and clamps the integer (matmul_ref_int8.cc:32-34); clamping the rounded float and
    // narrowing afterwards gives the same int8 for every |v| < 2^31 -- beyond that the reference's float -> int32 cast is UB
    r = fmaxf(r, (float)a.q_min);
    r = fminf(r, (float)a.q_max);
    return (int8_t)(int)r;
}

// the fp32 output of one accumulator BEFORE `accumulate` (`u` = bias_term of its column)
__device__ __forceinline__ float epilogue_f32(const W8A8Args &a, int acc, float u) {
    float v = __fmul_rn((float)acc, a.alpha);
    if (a.bias_kind == TCE_BIAS_FP32) v = __fadd_rn(v, u);
    return v;
}
// Staged outputs (round 6): a full-width tile [rows][TN] of int8_t or float outputs, written to LDS by the lanes that hold the accumulators, leaves as 16-byte row pieces --
// the accumulator layout gives a lane ONE column of four rows, i.e. 16 .. 64 one-element stores (and, with `accumulate`, as many dependent one-element loads) per lane.
// fp32 with `accumulate`: C = fadd_rn(C, result), one rounding, as epilogue_store.  staged_ok: the launch-uniform part of the condition.
__device__ __forceinline__ bool staged_ok(const W8A8Args &a, const void *Cb) {
    return (reinterpret_cast<uintptr_t>(Cb) & 15) == 0 && (a.out_kind == TCE_OUT_INT8 ? (a.ldc & 15) == 0 : (a.ldc & 3) == 0);
}
template <int TN, typename T>
__device__ __forceinline__ void flush_tile(const W8A8Args &a, void *Cb, const T *tile, int rows, int m0, int n0, int tid, int nthreads) {
    constexpr int EPP = 16 / (int)sizeof(T), PPR = TN / EPP;  // elements per piece, pieces per row
    for (int e = tid; e < rows * PPR; e += nthreads) {
        const int row = e / PPR, piece = e % PPR;
        if (m0 + row >= a.M) continue;
        T *dst = static_cast<T *>(Cb) + (size_t)(m0 + row) * a.ldc + n0 + piece * EPP;
        if constexpr (sizeof(T) == 4) {
            float4_t v = *reinterpret_cast<const float4_t *>(tile + row * TN + piece * EPP);
            if (a.accumulate) {
                const float4_t old = *reinterpret_cast<const float4_t *>(dst);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = __fadd_rn(old[r], v[r]);
            }
            *reinterpret_cast<float4_t *>(dst) = v;
        } else {
            *reinterpret_cast<int4_t *>(dst) = *reinterpret_cast<const int4_t *>(tile + row * TN + piece * EPP);
        }
    }
}

// `u` = bias_term(a, n), loaded by the caller once per column (the MFMA kernel's 16 outputs per lane share two columns:
// a bias load per element put 16 dependent memory round trips, ~10 us, behind a ~1 us contraction).
__device__ __forceinline__ void epilogue_store(const W8A8Args &a, void *Cb, int m, int n, int acc, float u) {
    if (a.out_kind == TCE_OUT_INT8) {
        static_cast<int8_t *>(Cb)[(size_t)m * a.ldc + n] = epilogue_i8(a, acc, u);
    } else {
        const float f = (float)acc;
        float v = __fmul_rn(f, a.alpha);
        if (a.bias_kind == TCE_BIAS_FP32) v = __fadd_rn(v, u);
        float *dst = static_cast<float *>(Cb) + (size_t)m * a.ldc + n;
        if (a.accumulate) v = __fadd_rn(*dst, v);  // add(a, b, c): c = a + b, one rounding (Int8OPTDecoderLayer.cc:14-22)
        *dst = v;
    }
}

// 4 waves as 2(M) x 2(N); each wave 2x2 MFMA tiles of 16x16 -> 64x64 per workgroup.
//
// Operand fragments: the MFMA wants row (lane % 16) and k-slice (lane / 16) in a lane.  Loading that straight from a
// K-contiguous matrix makes the 16 lanes of every quarter-wave read 16 different rows -- 64 cache-line look-ups per wave
// load, and the texture-address unit, not HBM or the matrix pipe, bounded the kernel (512x3072x768: 15.7 us; with a
// coalesced but wrong lane mapping 7.9 us).  So a lane loads row (lane / 4), chunk (lane % 4) -- 4 consecutive lanes read
// one row's 64 bytes -- and the 16x64-byte fragment is transposed through a per-wave 1 KiB LDS slot: lane-linear
// ds_write_b128 with an XOR swizzle (slot = row*4 + (chunk ^ (row/4 % 4))), one ds_read_b128 back in MFMA order
// (conflict-free both ways).  A wave's LDS operations complete in order, so one slot per fragment is enough and no
// barrier is involved.
// KS > 1: the K range is cut between KS wave quartets of the same 64x64 tile (the OPT shapes give 24-96 tiles, each a serial
// chain of K/64 steps: 512x768x3072 15.4 us with 96 workgroups).  int32 partial sums are exact, so the quartets' tiles are
// added through LDS in any order and the epilogue sees the same integers as the unsplit kernel: still bit-exact.
// MT (round 6): MFMA row tiles per wave -- 2: the 64 x 64 tile above; 1: a 32 x 64 tile (each wave 16 rows x 32 columns) for launches whose 64-row tiles leave most of the
// chip without a workgroup (OPT-125M's fc2 at 512 rows: 96 tiles of 64 x 64 on 256 CUs, VERDICT r5 weak 6 -> 192 tiles).  Same integers, same epilogue: bit-exact.
template <int KS, bool XS = false, int MT = 2>
__global__ __launch_bounds__(256 * KS) void w8a8_mfma_kernel(const W8A8Args a) {
    static_assert(MT == 2 || (MT == 1 && !XS), "32-row tiles: not combined with the cut across workgroups");
    extern __shared__ __attribute__((aligned(16))) int4_t lds_dyn[];  // [4 * KS waves][fragment: A0 A1 B0 B1][64 slots]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, grp = wave8 >> 2;
    int4_t(*lds_t)[64] = reinterpret_cast<int4_t(*)[64]>(lds_dyn + (size_t)wave8 * 4 * 64);  // this wave's four slots
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;   // MFMA view of the lane
    const int lrow = lane >> 2, lchunk = lane & 3;  // load view of the lane
    const int wslot = lrow * 4 + (lchunk ^ ((lrow >> 2) & 3));
    const int rslot = r16 * 4 + (kq ^ ((r16 >> 2) & 3));
    const int batch = XS ? 0 : blockIdx.z;        // (the cut across workgroups: one problem per launch, blockIdx.z = the run)
    const int part = XS ? (int)blockIdx.z : 0, xs = XS ? a.xs : 1;
    const int8_t *A = a.A + (size_t)batch * a.strideA;
    const int8_t *B = a.B + (size_t)batch * a.strideB;
    const size_t c_off = (size_t)batch * a.strideC;
    void *Cb = a.out_kind == TCE_OUT_INT8 ? static_cast<void *>(static_cast<int8_t *>(a.C) + c_off)
                                          : static_cast<void *>(static_cast<float *>(a.C) + c_off);
    const int m_base = blockIdx.y * (32 * MT) + wm * (16 * MT);
    const int n_base = blockIdx.x * 64 + wn * 32;

    const int8_t *pa[MT], *pb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i < MT) {
            int m = m_base + i * 16 + lrow;
            m = m < a.M ? m : a.M - 1;
            pa[i] = A + (size_t)m * a.lda + lchunk * 16;
        }
        int n = n_base + i * 16 + lrow;
        n = n < a.N ? n : a.N - 1;
        pb[i] = B + (size_t)n * a.ldb + lchunk * 16;
    }
    int4_t acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = int4_t{0, 0, 0, 0};
    float bterm[2];  // requested now, used after the contraction
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n_base + j * 16 + r16;
        bterm[j] = bias_term(a, n < a.N ? n : a.N - 1);
    }

    // one k-step (64 k): raw fragments -> LDS -> MFMA order -> 4 MFMAs
    auto contract = [&](const int4_t (&ra)[MT], const int4_t (&rb)[2]) {
        int4_t fa[MT], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < MT) lds_t[i][wslot] = ra[i];
            lds_t[2 + i][wslot] = rb[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < MT) fa[i] = lds_t[i][rslot];
            fb[i] = lds_t[2 + i][rslot];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };

    // this quartet's full k-steps: [k_begin, k_end)
    const int nfull = a.K >> 6;
    const int s_lo = part * nfull / xs, nloc = (part + 1) * nfull / xs - s_lo;  // this workgroup's run of k-steps (all of them unless XS)
    const int spg = (nloc + KS - 1) / KS;
    const int k_begin = (s_lo + (grp * spg < nloc ? grp * spg : nloc)) * 64;
    const int k_end = (s_lo + ((grp + 1) * spg < nloc ? (grp + 1) * spg : nloc)) * 64;
    // the next k-step's raw fragments are requested before the current one goes through LDS and the MFMAs (two static
    // register sets; requests past the end are clamped re-reads)
    auto load_raw = [&](int4_t (&ra)[MT], int4_t (&rb)[2], int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < MT) ra[i] = *reinterpret_cast<const int4_t *>(pa[i] + k0);
            rb[i] = *reinterpret_cast<const int4_t *>(pb[i] + k0);
        }
    };
    if (k_end > k_begin) {
        // (three steps of loads in flight through four static register sets, unrolled by four, measured SLOWER: hipcc answers
        // the ring with vmcnt(0) at most uses -- 512x768x3072 with two quartets 10.7 -> 12.5 us)
        const int k_last = k_end - 64;
        int4_t ra0[MT], rb0[2], ra1[MT], rb1[2];
        load_raw(ra0, rb0, k_begin);
        for (int k0 = k_begin; k0 < k_end; k0 += 128) {
            load_raw(ra1, rb1, k0 + 64 <= k_last ? k0 + 64 : k_last);
            __builtin_amdgcn_sched_barrier(0);  // (without it the scheduler sinks these loads to their use and both "sets" share registers: no load is in flight during a contraction)
            contract(ra0, rb0);
            if (k0 + 64 < k_end) {
                load_raw(ra0, rb0, k0 + 128 <= k_last ? k0 + 128 : k_last);
                __builtin_amdgcn_sched_barrier(0);
                contract(ra1, rb1);
            }
        }
    }
    const int k_full = nfull * 64;
    if (k_full < a.K && grp == KS - 1 && part == xs - 1) {  // K % 64 in {16,32,48}: chunks past the end contribute zeros (K % 16 == 0 is guaranteed)
        const bool live = k_full + lchunk * 16 < a.K;
        int4_t ra[MT], rb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int koff = live ? k_full : 0;  // dead chunks re-read a valid address and are zeroed
            if (i < MT) ra[i] = *reinterpret_cast<const int4_t *>(pa[i] + koff);
            rb[i] = *reinterpret_cast<const int4_t *>(pb[i] + koff);
            if (!live) {
                if (i < MT) ra[i] = int4_t{0, 0, 0, 0};
                rb[i] = int4_t{0, 0, 0, 0};
            }
        }
        contract(ra, rb);
    }

    if constexpr (KS > 1) {  // quartets 1.. hand their int32 tiles to quartet 0: [quartet - 1][register][thread of the quartet]
        __syncthreads();  // every wave is done with its transpose slots
        int4_t *red = lds_dyn;
        const int t4 = tid & 255;
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) red[((grp - 1) * 4 + i * 2 + j) * 256 + t4] = acc[i][j];
        }
        __syncthreads();
        if (!XS && grp > 0) return;  // (with the cut across workgroups every wave stays for the barriers below)
        if (grp == 0)
        for (int g2 = 0; g2 < KS - 1; ++g2)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int4_t o = red[(g2 * 4 + i * 2 + j) * 256 + t4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += o[r];
                }
    }
    if constexpr (XS) {
        // the run's partial tile, written through; the counter elects the last workgroup; it adds the other runs' tiles (any order: integers) -- visibility as in the
        // W4A16 GEMM's k-range cut and the attention step's merge: acknowledged device-scope stores, then the counter, then coherent loads
        const int tile = blockIdx.y * gridDim.x + blockIdx.x;
        const int t4x = tid & 255;
        const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(a.xpart + (size_t)tile * xs * 4 * 256, 0, xs * 4 * 256 * 16, 0x00020000);
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, acc[i][j]), rs_p, ((part * 4 + i * 2 + j) * 256 + t4x) * 16, 0, /*sc0|sc1*/ 17);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned *flag = reinterpret_cast<unsigned *>(lds_dyn);
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.xcnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = old == (unsigned)xs - 1 ? 1u : 0u;
            if (last) __hip_atomic_store(a.xcnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0u || grp > 0) return;
        for (int p = 0; p < xs; ++p) {
            if (p == part) continue;  // (workgroup-uniform)
            uint4_t o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, ((p * 4 + u) * 256 + t4x) * 16, 0, /*sc0|sc1*/ 17);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] += __builtin_bit_cast(int4_t, o[i * 2 + j]);
        }
    }

    // the outputs of a full-width tile leave through LDS as 16-byte row pieces (flush_tile above; 512 x 3072 x 768 writes 1.5 MB of int8 that way)
    if constexpr (!XS) {
        if (staged_ok(a, Cb) && (int)(blockIdx.x * 64 + 64) <= a.N) {  // workgroup-uniform
            __syncthreads();  // (every wave of the quartet is done with its transposition slots; quartets 1.. have left)
            const int trow = wm * (16 * MT) + kq * 4, tcol = wn * 32 + r16;
            if (a.out_kind == TCE_OUT_INT8) {
                int8_t *tile = reinterpret_cast<int8_t *>(lds_dyn);  // [32 * MT][64]
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) tile[(trow + i * 16 + r) * 64 + tcol + j * 16] = epilogue_i8(a, acc[i][j][r], bterm[j]);
                __syncthreads();
                flush_tile<64>(a, Cb, tile, 32 * MT, (int)blockIdx.y * (32 * MT), (int)blockIdx.x * 64, tid & 255, 256);
            } else {
                float *tile = reinterpret_cast<float *>(lds_dyn);  // [32 * MT][64]: 16 KiB, the first quartet's slots
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) tile[(trow + i * 16 + r) * 64 + tcol + j * 16] = epilogue_f32(a, acc[i][j][r], bterm[j]);
                __syncthreads();
                flush_tile<64>(a, Cb, tile, 32 * MT, (int)blockIdx.y * (32 * MT), (int)blockIdx.x * 64, tid & 255, 256);
            }
            return;
        }
    }
    // D[row = 4*(lane>>4) + r][col = lane & 15]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n_base + j * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + i * 16 + kq * 4 + r;
                if (m < a.M && n < a.N) epilogue_store(a, Cb, m, n, acc[i][j][r], bterm[j]);
            }
        }
}

// Round 6, second form for few tiles with a long chain: the WHOLE tile in every wave, the k-steps dealt to the waves.  In the kernel above a quartet's waves sit 2 x 2 on
// the tile, so every operand row is fetched by two waves: a 32 x 64 tile's k-step pulls 12 KiB through the CU's L1 for 6 KiB of operands, and 512 x 768 x 3072 (192
// workgroups x 48 steps) is bound by that path, not by the matrix pipe or the chain (two and four quartets tie).  Here a wave owns a run of k-steps and contracts the full
// (16 TM16) x (16 TN16) tile on them -- TM16 + TN16 fragments requested for TM16 x TN16 MFMAs, every operand byte through L1 ONCE per workgroup, the same per-wave LDS
// transposition, no barrier in the loop -- and the W waves' int32 tiles are added through LDS at the end (any order: integers, the same epilogue: bit-exact).  A 32 x 48
// tile gives 512 x 768 x 3072 exactly 256 workgroups; four waves (a wave walks its run three steps at a time) measured level with eight and sixteen.
template <int TM16, int TN16, int W>
__global__ __launch_bounds__(64 * W) void w8a8_kslice_kernel(const W8A8Args a) {
    constexpr int NF = TM16 + TN16;          // fragments of a k-step: A row tiles, then B column tiles
    constexpr int TILE4 = TM16 * TN16 * 64;  // the tile as int4 positions: [MFMA tile][lane]
    constexpr int D = 3;                     // k-steps requested at once
    constexpr int NP = (TILE4 + 64 * W - 1) / (64 * W);  // positions a thread finishes
    extern __shared__ __attribute__((aligned(16))) int4_t lds_ks[];  // [W waves][NF fragments][64 slots]; afterwards the sums' [W][TILE4]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int4_t(*lds_t)[64] = reinterpret_cast<int4_t(*)[64]>(lds_ks + (size_t)wave * NF * 64);
    const int r16 = lane & 15, kq = lane >> 4;
    const int lrow = lane >> 2, lchunk = lane & 3;
    const int wslot = lrow * 4 + (lchunk ^ ((lrow >> 2) & 3));
    const int rslot = r16 * 4 + (kq ^ ((r16 >> 2) & 3));
    const int batch = blockIdx.z;
    const int8_t *A = a.A + (size_t)batch * a.strideA;
    const int8_t *B = a.B + (size_t)batch * a.strideB;
    const size_t c_off = (size_t)batch * a.strideC;
    void *Cb = a.out_kind == TCE_OUT_INT8 ? static_cast<void *>(static_cast<int8_t *>(a.C) + c_off)
                                          : static_cast<void *>(static_cast<float *>(a.C) + c_off);
    const int m_base = blockIdx.y * (16 * TM16), n_base = blockIdx.x * (16 * TN16);

    const int8_t *p[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f < TM16) {
            int m = m_base + f * 16 + lrow;
            m = m < a.M ? m : a.M - 1;
            p[f] = A + (size_t)m * a.lda + lchunk * 16;
        } else {
            int n = n_base + (f - TM16) * 16 + lrow;
            n = n < a.N ? n : a.N - 1;
            p[f] = B + (size_t)n * a.ldb + lchunk * 16;
        }
    }
    // the additive terms of the positions this thread finishes, requested now
    float bterm[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pos = tid + q * 64 * W;
        const int n = n_base + ((pos >> 6) % TN16) * 16 + (pos & 15);
        bterm[q] = pos < TILE4 ? bias_term(a, n < a.N ? n : a.N - 1) : 0.0f;
    }
    int4_t acc[TM16][TN16];
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
        for (int j = 0; j < TN16; ++j) acc[i][j] = int4_t{0, 0, 0, 0};

    auto contract = [&](const int4_t (&raw)[NF]) {
        int4_t fr[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) lds_t[f][wslot] = raw[f];
#pragma unroll
        for (int f = 0; f < NF; ++f) fr[f] = lds_t[f][rslot];
#pragma unroll
        for (int i = 0; i < TM16; ++i)
#pragma unroll
            for (int j = 0; j < TN16; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fr[i], fr[TM16 + j], acc[i][j], 0, 0, 0);
    };

    const int nfull = a.K >> 6;
    const int s_begin = wave * nfull / W, s_end = (wave + 1) * nfull / W;  // this wave's k-steps (none when there are fewer steps than waves)
    for (int s = s_begin; s < s_end; s += D) {
        int4_t raw[D][NF];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int k0 = (s + d < s_end ? s + d : s_end - 1) * 64;  // (past the run: a clamped re-read, not contracted)
#pragma unroll
            for (int f = 0; f < NF; ++f) raw[d][f] = *reinterpret_cast<const int4_t *>(p[f] + k0);
        }
        __builtin_amdgcn_sched_barrier(0);  // (every request of the D steps is out before the first fragment goes through LDS: the scheduler otherwise sinks the loads to their uses)
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (s + d < s_end) contract(raw[d]);
    }
    if ((nfull << 6) < a.K && wave == W - 1) {  // K % 64 in {16, 32, 48}: chunks past the end contribute zeros (K % 16 == 0 is guaranteed)
        const bool live = (nfull << 6) + lchunk * 16 < a.K;
        int4_t raw[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            raw[f] = *reinterpret_cast<const int4_t *>(p[f] + (live ? (nfull << 6) : 0));  // dead chunks re-read a valid address and are zeroed
            if (!live) raw[f] = int4_t{0, 0, 0, 0};
        }
        contract(raw);
    }

    __syncthreads();  // every wave is done with its transposition slots
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
        for (int j = 0; j < TN16; ++j) lds_ks[(size_t)wave * TILE4 + (i * TN16 + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    int4_t sum[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pos = tid + q * 64 * W;
        sum[q] = int4_t{0, 0, 0, 0};
        if (pos >= TILE4) break;
        sum[q] = lds_ks[pos];
#pragma unroll
        for (int w = 1; w < W; ++w) {
            const int4_t o = lds_ks[(size_t)w * TILE4 + pos];
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[q][r] += o[r];
        }
    }
    // position = [MFMA tile i * TN16 + j][lane]: D[row = 4 * (lane >> 4) + r][col = lane & 15]
    // int8 outputs of a full-width tile leave through LDS as 16-byte row pieces (as in w8a8_mfma_kernel)
    constexpr int TNB = 16 * TN16;
    if (staged_ok(a, Cb) && n_base + TNB <= a.N && m_base + 16 < a.M) {  // workgroup-uniform (a tile of at most 16 live rows: the two extra barriers cost more than the stores save -- 16 x 768 x 3072 6.47 -> 6.65 us)
        __syncthreads();  // every thread has taken its sums
        if (a.out_kind == TCE_OUT_INT8) {
            int8_t *tile = reinterpret_cast<int8_t *>(lds_ks);  // [16 TM16][TNB]
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int pos = tid + q * 64 * W;
                if (pos >= TILE4) break;
                const int t = pos >> 6, l = pos & 63;
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[((t / TN16) * 16 + (l >> 4) * 4 + r) * TNB + (t % TN16) * 16 + (l & 15)] = epilogue_i8(a, sum[q][r], bterm[q]);
            }
            __syncthreads();
            flush_tile<TNB>(a, Cb, tile, 16 * TM16, m_base, n_base, tid, 64 * W);
        } else {
            float *tile = reinterpret_cast<float *>(lds_ks);  // [16 TM16][TNB]: at most 16 KiB
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int pos = tid + q * 64 * W;
                if (pos >= TILE4) break;
                const int t = pos >> 6, l = pos & 63;
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[((t / TN16) * 16 + (l >> 4) * 4 + r) * TNB + (t % TN16) * 16 + (l & 15)] = epilogue_f32(a, sum[q][r], bterm[q]);
            }
            __syncthreads();
            flush_tile<TNB>(a, Cb, tile, 16 * TM16, m_base, n_base, tid, 64 * W);
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pos = tid + q * 64 * W;
        if (pos >= TILE4) break;
        const int t = pos >> 6, l = pos & 63;
        const int n = n_base + (t % TN16) * 16 + (l & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m_base + (t / TN16) * 16 + (l >> 4) * 4 + r;
            if (m < a.M && n < a.N) epilogue_store(a, Cb, m, n, sum[q][r], bterm[q]);
        }
    }
}

// The same 64 x 64 tile for launches whose k chain is long and whose tiles are few (OPT-125M's 512 x 768 x 3072 is 96 tiles of 48 k-steps): the kernel above keeps ONE
// k-step of loads in flight per wave, and a step is 4 MFMAs -- 64 cycles of work against a memory round trip of ~1500 -- so the chain runs at memory LATENCY
// (0.3 us per step).  Here a quartet stages its panels cooperatively (a k-step is 64 + 64 rows of 64 bytes = ONE 16-byte piece of A and one of B per thread, half
// of what four waves fetching their own fragments move) and keeps P steps in flight in registers (P x 8 registers), written to a double-buffered LDS stage in
// the slot order of the kernel above one step ahead of the contraction; one LDS-only barrier per step (the requests stay in flight).  Requests past the end
// are clamped re-reads, so every wave issues the same loads and the wait counts are exact.  int32 sums: bit-exact in any order, like every other path.
template <int KS, int P>
__global__ __launch_bounds__(256 * KS) void w8a8_mfma_deep_kernel(const W8A8Args a) {
    constexpr int STAGE_SLOTS = 512;  // A: 64 rows x 4 | B: 64 rows x 4 (16-byte slots)
    extern __shared__ __attribute__((aligned(16))) int4_t lds_deep[];  // [KS quartets][2 stages][512 slots]
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3, grp = wave8 >> 2;
    const int tid = threadIdx.x & 255;  // within the quartet
    int4_t *ring = lds_deep + grp * (2 * STAGE_SLOTS);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int batch = blockIdx.z;
    const int8_t *A = a.A + (size_t)batch * a.strideA;
    const int8_t *B = a.B + (size_t)batch * a.strideB;
    const size_t c_off = (size_t)batch * a.strideC;
    void *Cb = a.out_kind == TCE_OUT_INT8 ? static_cast<void *>(static_cast<int8_t *>(a.C) + c_off) : static_cast<void *>(static_cast<float *>(a.C) + c_off);
    const int m_tile = blockIdx.y * 64, n_tile = blockIdx.x * 64;
    const int lrow = tid >> 2, lchunk = tid & 3;
    int m = m_tile + lrow, n = n_tile + lrow;
    m = m < a.M ? m : a.M - 1;
    n = n < a.N ? n : a.N - 1;
    const int8_t *pa = A + (size_t)m * a.lda + lchunk * 16, *pb = B + (size_t)n * a.ldb + lchunk * 16;
    const int wslot = lrow * 4 + (lchunk ^ ((lrow >> 2) & 3));
    const int rslot = r16 * 4 + (kq ^ ((r16 >> 2) & 3));
    const int fa0 = (wm * 32) * 4 + rslot, fb0 = 256 + (wn * 32) * 4 + rslot;
    int4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = int4_t{0, 0, 0, 0};
    const int Tall = a.K >> 6;                  // K % 64 == 0 (the host's condition)
    const int Tmax = (Tall + KS - 1) / KS;      // every wave passes Tmax barriers
    const int t_begin = grp * Tmax < Tall ? grp * Tmax : Tall;
    const int T = (t_begin + Tmax < Tall ? t_begin + Tmax : Tall) - t_begin;  // this quartet's k-steps (the last quartet's may be fewer, or none)
    const int t_last = T > 0 ? t_begin + T - 1 : 0;
    int4_t ra[P], rb[P];
    auto request = [&](int u, int t) {  // step t of this quartet (clamped: a re-read of its last step)
        const int tt = t_begin + t <= t_last ? t_begin + t : t_last;
        ra[u] = *reinterpret_cast<const int4_t *>(pa + (size_t)tt * 64);
        rb[u] = *reinterpret_cast<const int4_t *>(pb + (size_t)tt * 64);
    };
#pragma unroll
    for (int u = 0; u < P; ++u) request(u, u);
    __builtin_amdgcn_sched_barrier(0);
    // (the body covers three rounds of the register ring: hipcc's wait-count pass answers the loop HEADER with vmcnt(0) -- a drained pipeline once per body)
    constexpr int BODY = 3 * P;
    for (int t0 = 0; t0 < Tmax; t0 += BODY) {
#pragma unroll
        for (int v = 0; v < BODY; ++v) {
            const int u = v % P;
            const int t = t0 + v;
            if (t >= Tmax) break;
            int4_t *stage = ring + (t & 1) * STAGE_SLOTS;
            stage[wslot] = ra[u];
            stage[256 + wslot] = rb[u];
            __builtin_amdgcn_sched_barrier(0);
            request(u, t + P);
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            if (t < T) {
                int4_t fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i] = stage[fa0 + i * 64];
                    fb[i] = stage[fb0 + i * 64];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    if constexpr (KS > 1) {  // quartets 1.. hand their int32 tiles to quartet 0: [quartet - 1][register][thread of the quartet]
        lds_barrier();       // every wave is done with the stages
        int4_t *red = lds_deep;
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) red[((grp - 1) * 4 + i * 2 + j) * 256 + tid] = acc[i][j];
        }
        lds_barrier();
        if (grp > 0) return;
        for (int g2 = 0; g2 < KS - 1; ++g2)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int4_t o = red[(g2 * 4 + i * 2 + j) * 256 + tid];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += o[r];
                }
    }
    const int m_base = m_tile + wm * 32, n_base = n_tile + wn * 32;
    float bterm[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nn = n_base + j * 16 + r16;
        bterm[j] = bias_term(a, nn < a.N ? nn : a.N - 1);
    }
    // int8 outputs of a full-width tile leave through LDS as 16-byte row pieces (as in w8a8_mfma_kernel)
    if (staged_ok(a, Cb) && n_tile + 64 <= a.N) {  // workgroup-uniform
        lds_barrier();  // (every wave of the quartet is done with the stages / the exchange; quartets 1.. have left)
        const int trow = wm * 32 + kq * 4, tcol = wn * 32 + r16;
        if (a.out_kind == TCE_OUT_INT8) {
            int8_t *tile = reinterpret_cast<int8_t *>(lds_deep);  // [64][64]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tile[(trow + i * 16 + r) * 64 + tcol + j * 16] = epilogue_i8(a, acc[i][j][r], bterm[j]);
            lds_barrier();
            flush_tile<64>(a, Cb, tile, 64, m_tile, n_tile, tid, 256);
        } else {
            float *tile = reinterpret_cast<float *>(lds_deep);  // [64][64]: 16 KiB, one quartet's stages
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tile[(trow + i * 16 + r) * 64 + tcol + j * 16] = epilogue_f32(a, acc[i][j][r], bterm[j]);
            lds_barrier();
            flush_tile<64>(a, Cb, tile, 64, m_tile, n_tile, tid, 256);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nn = n_base + j * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = m_base + i * 16 + kq * 4 + r;
                if (mm < a.M && nn < a.N) epilogue_store(a, Cb, mm, nn, acc[i][j][r], bterm[j]);
            }
        }
}

// Prefill-sized problems (OPT-1.3B / 6.7B linears at 512+ rows: 512 x 4096 x 4096 ran at 0.45 POP/s on the 64 x 64 kernel above, whose waves each fetch their
// own operand rows from memory).  128 x TN tile per workgroup (TN = 128 or 64), 4 waves as 2 x 2, wave tile 64 x TN/2 = 4 x (TN/32) MFMA tiles of 16 x 16 x 64;
// both operand panels of a k-step (64 k: 128 + TN rows of 64 bytes) go through LDS ONCE per workgroup -- coalesced 16-byte loads (four lanes per 64-byte row
// piece), lane-linear ds_write_b128 into the XOR-swizzled slot order of the kernel above, fragment-shaped ds_read_b128 back -- in a ring of three stages:
// iteration t writes the registers loaded for step t + 1, requests step t + 2, passes ONE barrier (lds_barrier: the requests stay in flight) and contracts
// step t.  int32 accumulation and the shared epilogue: bit-exact like every other path (the order of an int32 sum is free).  K % 64 == 0.
// KS = 2: two wave quartets per tile, each with its own ring, take the two halves of K and add their int32 tiles through LDS at the end (exact in any order) --
// for launches whose tiles number fewer than the CUs' capacity: a quartet alone on its CU is latency-bound (0.75 us per k-step at 512 x 4096 x 4096).
template <int TN, int KS>
__global__ __launch_bounds__(256 * KS) void w8a8_mfma_big_kernel(const W8A8Args a) {
    constexpr int TM = 128, MI = 4, NJ = TN / 32, STAGES = 3;
    constexpr int A_SLOTS = TM * 4, B_SLOTS = TN * 4, STAGE_SLOTS = A_SLOTS + B_SLOTS;  // 16-byte slots
    constexpr int BL = TN / 64;  // B pieces per thread and k-step (A: 2)
    extern __shared__ __attribute__((aligned(16))) int4_t lds_all[];  // [KS quartets][STAGES][A: 128 rows x 4 | B: TN rows x 4]
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3, grp = wave8 >> 2;
    const int tid = threadIdx.x & 255;  // within the quartet
    int4_t *lds_big = lds_all + grp * (STAGES * STAGE_SLOTS);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kq = lane >> 4;
    const int batch = blockIdx.z;
    const int8_t *A = a.A + (size_t)batch * a.strideA;
    const int8_t *B = a.B + (size_t)batch * a.strideB;
    const size_t c_off = (size_t)batch * a.strideC;
    void *Cb = a.out_kind == TCE_OUT_INT8 ? static_cast<void *>(static_cast<int8_t *>(a.C) + c_off) : static_cast<void *>(static_cast<float *>(a.C) + c_off);
    const int m_tile = blockIdx.y * TM, n_tile = blockIdx.x * TN;
    // load view: thread -> (row = tid / 4 (+ 64 per piece), chunk = tid % 4); the slot it writes
    const int lrow = tid >> 2, lchunk = tid & 3;
    const int8_t *pa[2], *pb[BL];
    int wsa[2], wsb[BL];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = lrow + 64 * i;
        int m = m_tile + row;
        m = m < a.M ? m : a.M - 1;
        pa[i] = A + (size_t)m * a.lda + lchunk * 16;
        wsa[i] = row * 4 + (lchunk ^ ((row >> 2) & 3));
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int row = lrow + 64 * i;
        int n = n_tile + row;
        n = n < a.N ? n : a.N - 1;
        pb[i] = B + (size_t)n * a.ldb + lchunk * 16;
        wsb[i] = A_SLOTS + row * 4 + (lchunk ^ ((row >> 2) & 3));
    }
    // MFMA view: the fragment of 16-row block R0 is slot (R0 + r16) * 4 + (kq ^ (r16 / 4 % 4))  (R0 a multiple of 16)
    const int rslot = r16 * 4 + (kq ^ ((r16 >> 2) & 3));
    const int fa0 = (wm * 64) * 4 + rslot, fb0 = A_SLOTS + (wn * (TN / 2)) * 4 + rslot;
    int4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = int4_t{0, 0, 0, 0};
    const int Tall = a.K >> 6;
    const int Tmax = (Tall + KS - 1) / KS;                              // every wave passes Tmax barriers
    const int t_begin = KS > 1 ? grp * Tmax : 0;
    const int T = KS > 1 ? ((t_begin + Tmax < Tall ? t_begin + Tmax : Tall) - t_begin) : Tall;  // this quartet's k-steps (the last quartet's may be fewer, or none)
    // (Two register sets -- the loads of step t + 3 requested in iteration t and written in iteration t + 2 -- were measured: the unrolled loop took 142 + 128
    //  registers instead of 83 + 64, one workgroup per CU instead of three, and 2048 x 16384 x 4096 ran 367 us instead of 199.)
    int4_t ra[2], rb[BL];
    auto request = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const int4_t *>(pa[i] + (size_t)(t_begin + t) * 64);
#pragma unroll
        for (int i = 0; i < BL; ++i) rb[i] = *reinterpret_cast<const int4_t *>(pb[i] + (size_t)(t_begin + t) * 64);
    };
    auto stage_write = [&](int st) {
        int4_t *base = lds_big + st * STAGE_SLOTS;
#pragma unroll
        for (int i = 0; i < 2; ++i) base[wsa[i]] = ra[i];
#pragma unroll
        for (int i = 0; i < BL; ++i) base[wsb[i]] = rb[i];
    };
    if (KS == 1 || T > 0) {
        request(0);
        stage_write(0);
    }
    if (T > 1) request(1);
    for (int t = 0; t < (KS > 1 ? Tmax : T); ++t) {
        const int st = t % STAGES;
        if (t + 1 < T) stage_write((t + 1) % STAGES);  // (the registers requested one iteration ago)
        if (t + 2 < T) request(t + 2);
        lds_barrier();
        if constexpr (KS > 1) {
            if (t >= T) continue;
        }
        const int4_t *base = lds_big + st * STAGE_SLOTS;
        int4_t fa[MI], fb[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = base[fa0 + i * 64];
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = base[fb0 + j * 64];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (KS > 1) {  // the other quartets hand their int32 tiles to quartet 0 through LDS (the rings are done with)
        lds_barrier();
        int4_t *red = lds_all;  // [quartet - 1][MI * NJ][256 threads]
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) red[((grp - 1) * MI * NJ + i * NJ + j) * 256 + tid] = acc[i][j];
        }
        lds_barrier();
        if (grp > 0) return;
        for (int g2 = 0; g2 < KS - 1; ++g2)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int4_t o = red[(g2 * MI * NJ + i * NJ + j) * 256 + tid];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += o[r];
                }
    }
    // D[row = 4 * (lane >> 4) + r][col = lane & 15]; the additive terms of the lane's NJ columns once
    const int m_base = m_tile + wm * 64, n_base = n_tile + wn * (TN / 2);
    float bterm[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n_base + j * 16 + r16;
        bterm[j] = bias_term(a, n < a.N ? n : a.N - 1);
    }
    // int8 outputs of a full-width tile leave through LDS as 16-byte row pieces (as in w8a8_mfma_kernel: the accumulator layout gives a lane one column of four rows)
    if (staged_ok(a, Cb) && n_tile + TN <= a.N) {  // workgroup-uniform
        lds_barrier();  // (every wave of the quartet is done with the ring / the exchange; quartets 1.. have left)
        const int tcol = wn * (TN / 2) + r16;
        if (a.out_kind == TCE_OUT_INT8) {
            int8_t *tile = reinterpret_cast<int8_t *>(lds_all);  // [TM][TN]
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tile[(wm * 64 + i * 16 + kq * 4 + r) * TN + tcol + j * 16] = epilogue_i8(a, acc[i][j][r], bterm[j]);
            lds_barrier();
            flush_tile<TN>(a, Cb, tile, TM, m_tile, n_tile, tid, 256);
        } else {
            // fp32: the tile's two 64-row halves one after the other (64 x TN floats: 32 KiB of the ring's 48)
            float *tile = reinterpret_cast<float *>(lds_all);  // [64][TN]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h) lds_barrier();  // the first half has been read
                if (wm == h) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) tile[(i * 16 + kq * 4 + r) * TN + tcol + j * 16] = epilogue_f32(a, acc[i][j][r], bterm[j]);
                }
                lds_barrier();
                flush_tile<TN>(a, Cb, tile, 64, m_tile + 64 * h, n_tile, tid, 256);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n_base + j * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + i * 16 + kq * 4 + r;
                if (m < a.M && n < a.N) epilogue_store(a, Cb, m, n, acc[i][j][r], bterm[j]);
            }
        }
}

// Generic path: any K, and the *_batch variants where row i of A has its own B_i ([M][N][K], ref :79,:153).
// One wavefront per output element pair would be overkill at these sizes (decode-time BMMs: heads x tgt_len x 64):
// one thread per output, 16-byte loads when K % 16 == 0, exact int32 accumulate.
__global__ __launch_bounds__(256) void w8a8_generic_kernel(const W8A8Args a) {
    const int batch = blockIdx.z;
    const int8_t *A = a.A + (size_t)batch * a.strideA;
    const int8_t *B = a.B + (size_t)batch * a.strideB;
    const size_t c_off = (size_t)batch * a.strideC;
    void *Cb = a.out_kind == TCE_OUT_INT8 ? static_cast<void *>(static_cast<int8_t *>(a.C) + c_off)
                                          : static_cast<void *>(static_cast<float *>(a.C) + c_off);
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)a.M * a.N) return;
    const int m = (int)(idx / a.N), n = (int)(idx % a.N);
    const int8_t *pa = A + (size_t)m * a.lda;
    // b_per_row: row m of A has its own B_m; the B_m are strideB elements apart (dense [M][N][K] when the caller gave none)
    const int8_t *pb = a.b_per_row ? B + (size_t)m * a.strideB + (size_t)n * a.ldb : B + (size_t)n * a.ldb;
    int acc = 0;
    int k = 0;
    if (a.vec_ok) {
        for (; k < a.K; k += 16) {
            const int4_t va = *reinterpret_cast<const int4_t *>(pa + k);
            const int4_t vb = *reinterpret_cast<const int4_t *>(pb + k);
#pragma unroll
            for (int w = 0; w < 4; ++w) acc = __builtin_amdgcn_sdot4(va[w], vb[w], acc, false);
        }
    }
    for (; k < a.K; ++k) acc += (int)pa[k] * (int)pb[k];
    epilogue_store(a, Cb, m, n, acc, bias_term(a, n));
}

// Decode-sized problems.  PER_ROW = 0: M <= 8 activation rows against a shared B (the linears of a decode step: 1 x 768 x 3072 is
// 12 tiles of 64 x 64 for the MFMA kernel -- twelve workgroups walking 48 serial k-steps, 11 us -- and 2.4 MB that 192 workgroups stream
// in one round trip): ONE WAVE per output column, the lanes across K in 16-byte pieces (a row of B is read once, coalesced, and meets
// all M rows), the activation rows staged in LDS once per workgroup.  PER_ROW = 1: the *_batch members (row m of A has its own B_m)
// with long rows -- the probabilities x V^T product of a decode step, K = the context: one wave per output element, both operands
// coalesced (the one-thread-per-output kernel reads 512-byte rows strided across its lanes).  int32 sums, exact in any order; the
// epilogue is the shared one.
constexpr int kRowdotMaxM = 8;
template <int PER_ROW>
__global__ __launch_bounds__(256) void w8a8_rowdot_kernel(const W8A8Args a) {
    extern __shared__ __attribute__((aligned(16))) int4_t lds_dyn[];  // PER_ROW = 0: [M][K] int8
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int batch = blockIdx.z;
    const int8_t *A = a.A + (size_t)batch * a.strideA;
    const int8_t *B = a.B + (size_t)(PER_ROW ? 0 : batch) * a.strideB;
    const size_t c_off = (size_t)batch * a.strideC;
    void *Cb = a.out_kind == TCE_OUT_INT8 ? static_cast<void *>(static_cast<int8_t *>(a.C) + c_off)
                                          : static_cast<void *>(static_cast<float *>(a.C) + c_off);
    const int pieces = a.K >> 4;
    if constexpr (PER_ROW == 0) {
        // the activation rows into LDS, four pieces per thread in flight (left as one load -> one store per iteration, M = 8 x K = 3072 is six dependent
        // memory round trips in front of everything)
        const int total = a.M * pieces;
        for (int e0 = tid; e0 < total; e0 += 256 * 4) {
            int4_t v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = e0 + 256 * i, ec = e < total ? e : 0;
                const int r = ec / pieces, p = ec - r * pieces;
                v[i] = *reinterpret_cast<const int4_t *>(A + (size_t)r * a.lda + p * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (e0 + 256 * i < total) lds_dyn[e0 + 256 * i] = v[i];
        }
        __syncthreads();
        const int n = blockIdx.x * 4 + wave;  // wave-uniform
        if (n >= a.N) return;
        const int4_t *brow = reinterpret_cast<const int4_t *>(B + (size_t)n * a.ldb);
        const float u = bias_term(a, n);  // requested now, used after the contraction
        int acc[kRowdotMaxM];
#pragma unroll
        for (int mm = 0; mm < kRowdotMaxM; ++mm) acc[mm] = 0;
        // a lane's pieces of the row, eight at a time: all of their loads are in flight before the first dot product (K = 3072 is three
        // pieces per lane; one load per loop iteration made them three dependent memory round trips)
        for (int base = lane; base < pieces; base += 64 * 8) {
            int4_t w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int p = base + 64 * i;
                w[i] = p < pieces ? brow[p] : int4_t{0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int p = base + 64 * i;
                if (p < pieces) {
#pragma unroll
                    for (int mm = 0; mm < kRowdotMaxM; ++mm) {
                        if (mm < a.M) {
                            const int4_t x = lds_dyn[mm * pieces + p];
                            acc[mm] = __builtin_amdgcn_sdot4(w[i].x, x.x, acc[mm], false);
                            acc[mm] = __builtin_amdgcn_sdot4(w[i].y, x.y, acc[mm], false);
                            acc[mm] = __builtin_amdgcn_sdot4(w[i].z, x.z, acc[mm], false);
                            acc[mm] = __builtin_amdgcn_sdot4(w[i].w, x.w, acc[mm], false);
                        }
                    }
                }
            }
        }
        int mine = 0;  // after the butterflies every lane holds every total; lane mm keeps row mm's and stores it
#pragma unroll
        for (int mm = 0; mm < kRowdotMaxM; ++mm) {
            if (mm < a.M) {
                int v = acc[mm];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
                if (lane == mm) mine = v;
            }
        }
        if (lane < a.M) epilogue_store(a, Cb, lane, n, mine, u);
    } else {
        const long long idx = (long long)blockIdx.x * 4 + wave;
        if (idx >= (long long)a.M * a.N) return;
        const int m = (int)(idx / a.N), n = (int)(idx % a.N);
        const int4_t *arow = reinterpret_cast<const int4_t *>(A + (size_t)m * a.lda);
        const int4_t *brow = reinterpret_cast<const int4_t *>(B + (size_t)m * a.strideB + (size_t)n * a.ldb);
        const float u = bias_term(a, n);
        int acc = 0;
        for (int base = lane; base < pieces; base += 64 * 4) {
            int4_t w[4], x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = base + 64 * i;
                w[i] = p < pieces ? brow[p] : int4_t{0, 0, 0, 0};
                x[i] = p < pieces ? arow[p] : int4_t{0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc = __builtin_amdgcn_sdot4(w[i].x, x[i].x, acc, false);
                acc = __builtin_amdgcn_sdot4(w[i].y, x[i].y, acc, false);
                acc = __builtin_amdgcn_sdot4(w[i].z, x[i].z, acc, false);
                acc = __builtin_amdgcn_sdot4(w[i].w, x[i].w, acc, false);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) epilogue_store(a, Cb, m, n, acc, u);
    }
}

thread_local int g_w8a8_ks = 0;  // forced K split (tuning), 0 = automatic; 3: the decode-sized kernels above are not used (A/B)

}  // namespace

void set_w8a8_ksplit(int ks) { g_w8a8_ks = (ks >= 1 && ks <= 4) ? ks : 0; }
thread_local int g_w8a8_xs = 0;  // the cut across workgroups: 0 the rule, 1 off, 2 / 3 / 4 / 6 / 8 forced where the rule's bounds allow it (A/B)
void set_w8a8_xsplit(int xs) { g_w8a8_xs = (xs >= 1 && xs <= 8) ? xs : 0; }
thread_local int g_w8a8_deep = 0;  // the 64 x 64 tile with 8 k-steps in flight: 0 the rule, 1 / 2 / 4 forced with that many quartets, 9 off (A/B)
thread_local int g_w8a8_big = 0;  // the 128-row tiles: 0 the rule, 1 / 2 forced with 128 / 64 columns (one quartet), 3 / 4 the same with two quartets, 9 off (A/B)
thread_local int g_w8a8_rows32 = 0;  // the 32 x 64 tiles: 0 the rule, 1 forced wherever the 64 x 64 kernel would run, 2 off (A/B)
void set_w8a8_rows32(int r) { g_w8a8_rows32 = (r >= 0 && r <= 2) ? r : 0; }
thread_local int g_w8a8_kslice = 0;  // the whole tile in every wave (w8a8_kslice_kernel): 0 the rule, 1 off, else forced: 304 / 404 / 904 = the 32 x 48 / 32 x 64 / 64 x 64 tile
void set_w8a8_kslice(int c) { g_w8a8_kslice = (c == 1 || c == 304 || c == 404 || c == 904) ? c : 0; }
void set_w8a8_big(int b) { g_w8a8_big = b; }
void set_w8a8_deep(int d) { g_w8a8_deep = d; }

size_t w8a8_scratch_bytes() { return 4096 + (size_t)1024 * 16384; }  // [1024 tile counters][1024 partial tiles of 64 x 64 int32]

// describe != nullptr: nothing is launched and no HIP call is made -- the form the rules below pick is written there as text (tce_w8a8_describe_dispatch; `scratch` then only
// says whether the caller would hand one over).  ONE decision path with the launch: every branch below names its form through TCE_W8A8_FORM in front of its launch.
#define TCE_W8A8_FORM(...)                                        \
    if (describe) {                                               \
        std::snprintf(describe, (size_t)describe_len, __VA_ARGS__); \
        return TCE_OK;                                            \
    }
int launch_w8a8(const tce_w8a8_desc &d, hipStream_t stream, hipError_t *hip_err, void *scratch, char *describe, int describe_len) {
    W8A8Args a{};
    a.A = static_cast<const int8_t *>(d.A);
    a.B = static_cast<const int8_t *>(d.B);
    a.bias = d.bias;
    a.C = d.C;
    a.strideA = d.batch > 1 ? d.strideA : 0;
    a.strideB = d.batch > 1 ? d.strideB : 0;
    a.strideC = d.batch > 1 ? d.strideC : 0;
    a.M = d.M;
    a.N = d.N;
    a.K = d.K;
    a.lda = d.lda ? d.lda : d.K;
    a.ldb = d.ldb ? d.ldb : d.K;
    a.ldc = d.ldc ? d.ldc : d.N;
    a.accumulate = d.accumulate;
    if (a.lda < d.K || a.ldb < d.K || a.ldc < d.N) return TCE_ERR_BAD_ARG;
    if (d.b_per_row) {  // the per-row B_m: strideB apart when given, dense [M][N][K] otherwise (batch is 1: tce_w8a8_matmul rejects anything else)
        if (d.batch > 1) return TCE_ERR_UNSUPPORTED_KIND;
        a.strideB = d.strideB ? d.strideB : (long long)d.N * a.ldb;
    }
    a.alpha = d.alpha;
    a.beta = d.beta;
    a.q_min = d.q_min;
    a.q_max = d.q_max;
    a.bias_kind = d.bias_kind;
    a.out_kind = d.out_kind;
    a.b_per_row = d.b_per_row;
    const bool aligned = (d.K % 16 == 0) && (reinterpret_cast<uintptr_t>(d.A) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(d.B) % 16 == 0) && a.lda % 16 == 0 && a.ldb % 16 == 0 && a.strideB % 16 == 0 &&
                         (d.batch == 1 || (d.strideA % 16 == 0 && d.strideB % 16 == 0));
    a.vec_ok = aligned ? 1 : 0;
    const bool rowdot_ok = aligned && g_w8a8_ks != 3;
    // wave-per-column for M <= 4, and for 5 .. 8 rows when K is long (8 x 768 x 768: 5.2 us against the MFMA kernel's 4.6; 8 x 768 x 3072: 6.5 against 9.8;
    // 1 x 768 x 3072: 3.4 against 9.4 -- scratch measurements of round 3); from 3 rows on only up to 4096 columns (round 4, scripts/probes/w8a8_rowdot_ab.py: a wave per
    // column walks all M rows -- 8 x 16384 x 4096: 30.5 us against the MFMA tiles' 15.6; 5 x 8192 x 2048: 10.4 against 9.2)
    if (rowdot_ok && !d.b_per_row && d.M <= kRowdotMaxM && (d.M <= 2 || (d.N <= 4096 && (d.M <= 4 || d.K >= 2048))) && d.K >= 64 && (size_t)d.M * d.K <= 64 * 1024) {
        TCE_W8A8_FORM("w8a8 wave-per-column rows=%d", d.M)
        hipLaunchKernelGGL(w8a8_rowdot_kernel<0>, dim3((d.N + 3) / 4, 1, d.batch), dim3(256), (size_t)d.M * d.K, stream, a);
    } else if (rowdot_ok && d.b_per_row && d.batch == 1 && d.K >= 256) {
        const long long outs = (long long)d.M * d.N;
        TCE_W8A8_FORM("w8a8 wave-per-output (a B per row of A)")
        hipLaunchKernelGGL(w8a8_rowdot_kernel<1>, dim3((unsigned)((outs + 3) / 4), 1, d.batch), dim3(256), 0, stream, a);
    } else if (!d.b_per_row && aligned && d.K % 64 == 0 && d.K >= 256 && g_w8a8_big != 9 &&
               ((g_w8a8_big >= 1 && g_w8a8_big <= 4) || (long)((d.M + 127) / 128) * ((d.N + 63) / 64) * d.batch >= 512 ||
                ((long)((d.M + 127) / 128) * ((d.N + 63) / 64) * d.batch >= 256 && d.K >= 2048 && d.K < 4096) ||
                (long)((d.M + 127) / 128) * ((d.N + 63) / 64) * d.batch >= 384)) {  // (round 6: from 384 tiles of 128 x 64 on -- 1024 x 3072 x 768 12.1 -> 9.7 us, 640 x 5120 x 1280 19.6 -> 13.7, 1024 x 3584 x 1024 15.4 -> 11.0, 768 x 4096 x 4096 36.5 -> 32.2; never behind the 64 x 64 kernels on nine launches: profiles/r6/w8a8_kslice_ab.jsonl)  // (round 4: from 64 k-steps on the 64 x 64 deep-pipeline kernel is level or ahead at one tile per CU -- 512 x 4096 x 16384: 83 -> 78 us -- and far ahead where M leaves most of a 128-row tile empty: 16 x 16384 x 4096 23.2 -> 16.6)
        // prefill-sized (scripts/w8a8_gemm_sizes.py, profiles/r3/w8a8_gemm_sizes.jsonl; never slower than the 64 x 64 kernel on the 18 shapes measured):
        // 128 x 128 tiles from two per CU on (512 x 16384 x 4096: 149 -> 57 us; 2048 x 4096 x 4096: 131 -> 56), 128 x 64 tiles from two per CU on
        // (512 x 8192 x 2048: 32 -> 24), from one per CU on with TWO quartets per tile when K is long (512 x 4096 x 4096: 38 -> 29; 512 x 4096 x 16384: 124 -> 85),
        // else the 64 x 64 kernel (512 x 2048 x 2048 stays at 10.7 us, 512 x 2048 x 8192 at 36).
        const long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch;
        const long t64 = (long)((d.M + 127) / 128) * ((d.N + 63) / 64) * d.batch;
        // (round 4: one 128 x 128 tile per CU with two quartets on half of K each beats the 128 x 64 tiles when K is long -- 2048 x 2048 x 8192: 71.8 -> 64.4 us)
        const bool long_k_one_per_cu = g_w8a8_big == 0 && t128 >= 256 && t128 < 512 && d.K >= 8192;
        const bool wide = g_w8a8_big == 1 || g_w8a8_big == 3 || (g_w8a8_big == 0 && t128 >= 512) || long_k_one_per_cu;
        const bool split = g_w8a8_big == 3 || g_w8a8_big == 4 || (g_w8a8_big == 0 && !wide && t64 < 512 && d.K >= 2048) || long_k_one_per_cu;  // (chains under 32 k-steps: one quartet -- 1024 x 3072 x 768 9.7 us against 10.9 with two)
        const size_t ring = (size_t)3 * (128 + (wide ? 128 : 64)) * 4 * 16;
        const size_t lds = split ? ((2 * ring > (size_t)(wide ? 16 : 8) * 256 * 16) ? 2 * ring : (size_t)(wide ? 16 : 8) * 256 * 16) : ring;
        const dim3 grid((d.N + (wide ? 127 : 63)) / (wide ? 128 : 64), (d.M + 127) / 128, d.batch);
        auto launch = [&](auto kfn, int threads) {
            if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kfn, grid, dim3(threads), lds, stream, a);
        };
        TCE_W8A8_FORM("w8a8 tile=128x%d quartets=%d", wide ? 128 : 64, split ? 2 : 1)
        if (wide && split) launch(w8a8_mfma_big_kernel<128, 2>, 512);
        else if (wide) launch(w8a8_mfma_big_kernel<128, 1>, 256);
        else if (split) launch(w8a8_mfma_big_kernel<64, 2>, 512);
        else launch(w8a8_mfma_big_kernel<64, 1>, 256);
    } else if (!d.b_per_row && aligned && d.K >= 64) {
        dim3 grid((d.N + 63) / 64, (d.M + 63) / 64, d.batch);
        // wave quartets per tile: while the tiles do not fill the chip and every quartet keeps >= 2 k-steps
        const long tiles = (long)grid.x * grid.y * grid.z;
        int ks = g_w8a8_ks == 3 ? 0 : g_w8a8_ks;
        if (ks == 0) {
            ks = 1;
            if (tiles <= 256 && d.K / 64 >= 4) ks = 2;  // four quartets measured no better than two (profiles/r1/w8a8_ksplit_sweep.jsonl); from 384 tiles on ONE is ahead (512 x 3072 x 768: 8.45 -> 8.0 us, profiles/r4/w8a8_form_sweep_before_refit.jsonl)
        }
        const size_t lds = (size_t)ks * 4 * 4 * 64 * 16;  // transpose slots; the reduction ((ks - 1) * 16 KiB) reuses them
        // long k chains on few tiles: the cooperative panels with 8 k-steps in flight (w8a8_mfma_deep_kernel)
        // Rule (round 4, scripts/w8a8_form_sweep.py over 36 launches, profiles/r4/w8a8_form_sweep_*.jsonl): from 48 k-steps on it is level with or ahead of the kernel above at every
        // tile count this branch sees -- 2048 x 768 x 3072: 25.6 -> 19.9 us, 108 x 768 x 3072: 15.3 -> 13.6, 512 x 2048 x 8192: 38.5 -> 30.3, 16 x 4096 x 16384: 58 -> 39 -- and behind it
        // at 32 and 12 steps (512 x 2048 x 2048: 10.2 -> 11.1; 512 x 768 x 768: 6.3 -> 8.0).  Two quartets up to 256 tiles (up to 512 from 128 steps on: 512 x 4096 x 8192 47.4 -> 44.8), one beyond; four were never ahead.
        // (48 .. 63 steps on 64 .. 256 tiles stay with the kernel above: a tie on weights from HBM -- 512 x 768 x 3072 13.1 / 13.2 us --, 6 % behind on weights that sit in L2: 11.5 / 10.85)
        const bool deep = g_w8a8_deep != 9 && d.K % 64 == 0 && (g_w8a8_deep > 0 || d.K / 64 >= 64 || (d.K / 64 >= 48 && (tiles < 64 || tiles > 256)));
        // round 5: few tiles with a long chain and a scratch area from the caller (tce_w8a8_desc_v2.scratch): the k-steps cut into runs on several workgroups, as many as
        // keep >= 6 steps per run and <= 384 workgroups (512 x 768 x 3072: 96 tiles x 4 runs of 12 steps; 108 x 768 x 3072: 24 tiles x 8 runs of 6)
        int xs = 1;
        // MEASURED (profiles/r5/w8a8_xsplit_ab.jsonl): the exchange is three dependent memory round trips (acknowledged stores, the counter, the coherent re-reads: ~5 us) --
        // 512 x 768 x 3072 (96 tiles, an ~8 us chain) 11.2 us uncut, 11.1 / 13.6 / 12.0 / 10.8 / 11.0 cut in 2 / 3 / 4 / 6 / 8: nothing gained; 108 x 768 x 3072 (24 tiles)
        // 11.4 -> 9.5 cut in 3 or 4, 11.0 cut in 8.  The rule cuts launches of at most 32 tiles, in four; forced cuts (tce_w4a16_set_debug_mode(182 .. 188)) go up to 128 tiles.
        if (scratch && (reinterpret_cast<uintptr_t>(scratch) & 255) == 0 && d.batch == 1 && g_w8a8_deep == 0 && g_w8a8_ks == 0 && g_w8a8_xs != 1 && tiles <= 128 && d.K / 64 >= 24) {
            if (g_w8a8_xs == 0) {
                if (tiles <= 32 && d.K / 64 / 4 >= 6) xs = 4;
            } else if (tiles * g_w8a8_xs <= 1024 && d.K / 64 / g_w8a8_xs >= 3) {
                xs = g_w8a8_xs;
            }
        }
        // round 6: the whole tile in every wave, the k-steps dealt to the waves (w8a8_kslice_kernel), four waves per workgroup.  Chains of >= 12 k-steps; the smallest of
        // 32 x 48 / 32 x 64 / 64 x 64 whose workgroups are at most one per CU (what bounds these launches is what ONE CU pulls through its L1: (rows + columns) x K per
        // workgroup), or 32 x 48 at two per CU.  MEASURED (profiles/r6/w8a8_kslice_ab.jsonl, weights rotating through HBM, 28 launches from 16 x 768 x 768 to
        // 512 x 2048 x 8192 against every form): 512 x 768 x 3072 10.24 -> 7.40 us, 512 x 768 x 768 5.07 -> 4.26, 108 x 768 x 3072 13.66 -> 6.70, 16 x 768 x 3072 10.90 -> 6.44,
        // 16 x 4096 x 16384 41.4 -> 27.6, 512 x 1024 x 4096 14.69 -> 9.80 (32 x 64), 512 x 2048 x 8192 30.5 -> 24.1 (64 x 64); never behind the other kernels where the rule
        // takes it, 2 % behind the best form over the set (eight and sixteen waves per workgroup: within 5 % of four everywhere, not kept).
        int kslice = g_w8a8_kslice > 1 ? g_w8a8_kslice : 0;
        if (g_w8a8_kslice == 0 && d.K / 64 >= 12 && g_w8a8_ks == 0 && g_w8a8_deep == 0 && g_w8a8_xs == 0 && g_w8a8_rows32 == 0) {  // (a forced form of another family: that family)
            const long r32 = (d.M + 31) / 32, w48 = r32 * ((d.N + 47) / 48) * d.batch, w64 = r32 * ((d.N + 63) / 64) * d.batch;
            if (w48 <= 256 || (w48 >= 448 && w48 <= 512)) kslice = 304;
            else if (w64 <= 256) kslice = 404;
            else if (tiles <= 256) kslice = 904;
            else if (tiles <= 512 && d.K / 64 >= 64) kslice = 904;  // (two per CU on long chains, against the deep-pipeline tile: 512 x 4096 x 4096 29.5 -> 28.2 us, x 16384 79.5 -> 74.7)
        }
        if (kslice) {
            const int tm16 = kslice >= 500 ? 4 : 2, tn16 = (kslice % 500) / 100, w = kslice % 100;
            const dim3 gk((d.N + 16 * tn16 - 1) / (16 * tn16), (d.M + 16 * tm16 - 1) / (16 * tm16), d.batch);
            const size_t slots = (size_t)w * (tm16 + tn16) * 64 * 16, sums = (size_t)w * tm16 * tn16 * 64 * 16;
            const size_t ldk = slots > sums ? slots : sums;  // (<= 64 KiB for every form kept)
            if (kslice != 304 && kslice != 404 && kslice != 904) return TCE_ERR_BAD_ARG;
            TCE_W8A8_FORM("w8a8 k-slice tile=%dx%d waves=%d workgroups=%ld", 16 * tm16, 16 * tn16, w, (long)gk.x * gk.y * gk.z)
            switch (kslice) {
                case 304: hipLaunchKernelGGL((w8a8_kslice_kernel<2, 3, 4>), gk, dim3(64 * w), ldk, stream, a); break;
                case 404: hipLaunchKernelGGL((w8a8_kslice_kernel<2, 4, 4>), gk, dim3(64 * w), ldk, stream, a); break;
                case 904: hipLaunchKernelGGL((w8a8_kslice_kernel<4, 4, 4>), gk, dim3(64 * w), ldk, stream, a); break;
                default: return TCE_ERR_BAD_ARG;
            }
        } else
        if (xs > 1) {
            a.xs = xs;
            a.xcnt = static_cast<unsigned *>(scratch);
            a.xpart = reinterpret_cast<int4_t *>(static_cast<unsigned char *>(scratch) + 4096);
            const dim3 gx(grid.x, grid.y, xs);
            TCE_W8A8_FORM("w8a8 tile=64x64 quartets=%d kcut=%d", d.K / 64 / xs >= 4 ? 2 : 1, xs)
            if (d.K / 64 / xs >= 4) hipLaunchKernelGGL((w8a8_mfma_kernel<2, true>), gx, dim3(512), (size_t)2 * 4 * 4 * 64 * 16, stream, a);
            else hipLaunchKernelGGL((w8a8_mfma_kernel<1, true>), gx, dim3(256), (size_t)4 * 4 * 64 * 16, stream, a);
        } else
        if (deep) {
            const int dks = g_w8a8_deep == 1 || g_w8a8_deep == 2 || g_w8a8_deep == 4 ? g_w8a8_deep : (tiles <= 256 || (tiles <= 512 && d.K / 64 >= 128) ? 2 : 1);
            const size_t dl = (size_t)dks * 2 * 512 * 16;  // (>= the reduction's (dks - 1) * 16 KiB)
            TCE_W8A8_FORM("w8a8 tile=64x64 deep-pipeline quartets=%d", dks)
            if (dks == 4) hipLaunchKernelGGL((w8a8_mfma_deep_kernel<4, 8>), grid, dim3(1024), dl, stream, a);
            else if (dks == 2) hipLaunchKernelGGL((w8a8_mfma_deep_kernel<2, 8>), grid, dim3(512), dl, stream, a);
            else hipLaunchKernelGGL((w8a8_mfma_deep_kernel<1, 8>), grid, dim3(256), dl, stream, a);
        } else
        // round 6: few 64 x 64 tiles with a long chain -- 512 x 768 x 3072 is 96 workgroups on 256 CUs -- run on 32 x 64 tiles instead (twice the
        // workgroups, the same chain per quartet; each CU pulls half the rows through its L1).  The rule: at most 128 tiles of 64 x 64, at least two 32-row tiles, >= 12 k-steps.
        // MEASURED (profiles/r6/w8a8_rows32_ab.jsonl, weights from HBM): 512 x 768 x 3072 13.19 -> 10.16 us (bench harness, weights in L2: 10.85 -> 8.16), 512 x 768 x 768
        // 6.26 -> 5.04 (hence >= 12 k-steps, not 24); 108 x 768 x 3072 a tie (13.6: the cut across workgroups above serves it); two quartets = four.
        if ((g_w8a8_rows32 == 1 || (g_w8a8_rows32 == 0 && tiles <= 128 && d.K / 64 >= 12)) && d.M > 32) {
            const dim3 g32(grid.x, (d.M + 31) / 32, d.batch);
            const long tiles32 = (long)g32.x * g32.y * g32.z;
            int ks32 = g_w8a8_ks == 3 ? 0 : g_w8a8_ks;
            if (ks32 == 0) ks32 = (tiles32 <= 256 && d.K / 64 >= 4) ? 2 : 1;
            const size_t lds32 = (size_t)ks32 * 4 * 4 * 64 * 16;
            TCE_W8A8_FORM("w8a8 tile=32x64 quartets=%d", ks32)
            if (ks32 == 4) hipLaunchKernelGGL((w8a8_mfma_kernel<4, false, 1>), g32, dim3(1024), lds32, stream, a);
            else if (ks32 == 2) hipLaunchKernelGGL((w8a8_mfma_kernel<2, false, 1>), g32, dim3(512), lds32, stream, a);
            else hipLaunchKernelGGL((w8a8_mfma_kernel<1, false, 1>), g32, dim3(256), lds32, stream, a);
        } else {
            TCE_W8A8_FORM("w8a8 tile=64x64 quartets=%d", ks)
            if (ks == 4) hipLaunchKernelGGL(w8a8_mfma_kernel<4>, grid, dim3(1024), lds, stream, a);
            else if (ks == 2) hipLaunchKernelGGL(w8a8_mfma_kernel<2>, grid, dim3(512), lds, stream, a);
            else hipLaunchKernelGGL(w8a8_mfma_kernel<1>, grid, dim3(256), lds, stream, a);
        }
    } else {
        const long long total = (long long)d.M * d.N;
        dim3 grid((unsigned)((total + 255) / 256), 1, d.batch);
        TCE_W8A8_FORM("w8a8 generic (one output per thread)")
        hipLaunchKernelGGL(w8a8_generic_kernel, grid, dim3(256), 0, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}
#undef TCE_W8A8_FORM

}  // namespace tce
