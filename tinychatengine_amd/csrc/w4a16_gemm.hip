// w4a16_gemm.hip -- W4A16 dequant-GEMM on MFMA for gfx950: the first, register-staged kernel.  The dispatcher now uses it only as
// the fallback of w4a16_gemm_dma.hip (which is 7-30 % faster and explains why); its tiles stay selectable
// (tce_w4a16_set_gemm_config) as the baseline of the sweeps.
//
// The reference has no GEMM kernel on this layout: gemv_forward_cuda re-runs its GEMV once per input row
// (grid.z = M, kernels/cuda/gemv_cuda.cu:229-231) and gemm_forward_cuda* are declared but never defined
// (kernels/matmul.h:140-145).  The math is the same dequant + dot product as the GEMV:
//     C[m][n] = fp16( sum_g fp32(s[n][g]) * sum_{k in g} A[m][k] * (q[n][k] - z[n][g]) ),   fp32 accumulate on MFMA.
//
// CDNA4 design:
//   * v_mfma_f32_16x16x32_f16; the A operand is the activation tile, the B operand the dequantized weights, so a
//     lane's accumulator registers share one output channel n = lane & 15 and the group scale is a per-lane scalar.
//   * weights never touch LDS: in the q4_6 layout one 32-bit word is exactly the 8 consecutive k of one n that a
//     lane feeds to one MFMA, so each lane streams a 16-byte chunk (4 MFMA k-steps) straight from HBM/L2 into
//     registers and converts it with the magic-number trick (tce_common.hpp).  Any permutation of k inside the
//     contraction is legal as long as A uses the same one: the pair order (0,4,1,5,2,6,3,7) is applied to the
//     activation tile while it is staged into LDS, never to the weights.
//   * the activation tile (BM x 128 halves) is staged through registers into an LDS image that is lane-linear
//     per MFMA fragment (ds_read_b128 conflict-free) with an XOR swizzle (row ^ k-step ^ 4*k-quarter) that
//     spreads a wave's coalesced staging writes over all 16 bank groups (row ^ k-step alone used 4 of them).
//   * 4 waves along N, each MT x NT tiles of 16x16; block -> tile mapping keeps every XCD on its own set of
//     weight columns so the re-reads of a weight panel by the M/BM row-blocks hit that XCD's L2.
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

namespace {

struct GemmArgs {
    const half_t *A;
    const uint4_t *qweight;
    const half_t *scales;
    const unsigned *zeros;
    half_t *C;
    int M, N, K, lda, ldc, scales_stride, zeros_stride, log2g;
    int n_blocks, m_blocks;
    int add_to_c;  // TCE_W4_ADD_TO_C
    int xm, m_per, n_per;  // XCD grid: xm x (8 / xm) XCDs over (row blocks x column blocks), contiguous slices of m_per x n_per blocks
};

thread_local int g_gemm_xm = 1;

template <int MT, int NT>
__global__ __launch_bounds__(256, (MT * NT >= 16 ? 1 : 2)) void w4a16_gemm_kernel(const GemmArgs g) {
    constexpr int BM = MT * 16;
    constexpr int BN = 4 * NT * 16;
    constexpr int BK = 128;
    __shared__ __attribute__((aligned(16))) uint4_t lds_a[2][MT * 4 * 64];  // double buffer x [mt][s][64 lanes] 16-byte pieces

    // ---- XCD-aware tile mapping: workgroup b runs on XCD b % 8 (observed, used for speed only) ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    const int m_blk = (xcd % g.xm) * g.m_per + slot % g.m_per;
    const int n_blk = (xcd / g.xm) * g.n_per + slot / g.m_per;
    if (n_blk >= g.n_blocks || m_blk >= g.m_blocks) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15;
    const int q = lane >> 4;
    const int m_base = m_blk * BM;
    const int n_base = n_blk * BN + wave * (NT * 16);
    const int nchunks = g.K >> 5;
    const int nkb = g.K / BK;

    // ---- per-lane weight rows ----
    int nrow[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n_base + j * 16 + n16;
        nrow[j] = n < g.N ? n : g.N - 1;
    }
    struct BRegs {
        uint4_t w[NT];
        half_t s[NT];
        unsigned z[NT];
    };
    auto load_b = [&](BRegs &b, int kb) {
        const int chunk = kb * 4 + q;
        const int grp = (chunk << 5) >> g.log2g;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            b.w[j] = g.qweight[(size_t)nrow[j] * nchunks + chunk];
            b.s[j] = g.scales[(size_t)nrow[j] * g.scales_stride + grp];
            b.z[j] = g.zeros[(size_t)nrow[j] * g.zeros_stride + (grp >> 3)];
        }
    };
    // ---- activation staging: MT pieces of 16 bytes per thread ----
    uint4_t areg[MT];
    auto load_a = [&](int kb) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int e = i * 256 + tid;
            const int row = e >> 4, pc = e & 15;
            int m = m_base + row;
            m = m < g.M ? m : g.M - 1;
            areg[i] = *reinterpret_cast<const uint4_t *>(g.A + (size_t)m * g.lda + kb * BK + pc * 8);
        }
    };
    auto write_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int e = i * 256 + tid;
            const int row = e >> 4, pc = e & 15;
            const int qq = pc >> 2, s = pc & 3;  // k offset 8*pc = 32*qq + 8*s
            lds_a[buf][((row >> 4) * 4 + s) * 64 + qq * 16 + ((row & 15) ^ s ^ (qq << 2))] = pair_permute(areg[i]);
        }
    };

    const NibbleMasks nmask = make_nibble_masks();
    float4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    // One k-block (128 k) is exactly one quantization group (the kernel is only dispatched for G == 128), so the
    // MFMAs of a block contract EXACT integers (q - z) with the fp16 activations in fp32, and the fp16 group scale is
    // applied once per block in fp32: acc += s * acc_block -- the same precision class as the GEMV kernel and the
    // oracle (rounding the scaled weight to fp16 first costs ~2^-12 relative per weight, which breaks 1e-3 on outputs
    // that are small by cancellation).
    auto compute = [&](const BRegs &b, int kb, int buf) {
        const int grp = ((kb * 4 + q) << 5) >> g.log2g;
        const int zsh = (grp & 7) * 4;
        ZeroPair zp[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) zp[j] = make_zero_pair((b.z[j] >> zsh) & 0xFu);
        float4_t blk[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) blk[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            half8_t bf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                half2_t d[4];
                dequant_word(b.w[j][s], zp[j], nmask, d);
                bf[j] = half8_t{d[0].x, d[0].y, d[1].x, d[1].y, d[2].x, d[2].y, d[3].x, d[3].y};
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint4_t araw = lds_a[buf][(i * 4 + s) * 64 + q * 16 + (n16 ^ s ^ (q << 2))];  // lane's row m16 == lane & 15
                const half8_t af = __builtin_bit_cast(half8_t, araw);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    blk[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], blk[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float sc = (float)b.s[j];  // lane's output channel n = lane & 15 is the same for its 4 registers
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(sc, blk[i][j][r], acc[i][j][r]);
        }
    };

    // Double-buffered LDS, one barrier per k-block: while block kb is contracted out of lds_a[kb & 1], the registers
    // prefetch block kb+1 (activations and weights); the activations are written to the other buffer after the MFMAs.
    BRegs b0, b1;
    load_a(0);
    load_b(b0, 0);
    write_a(0);
    __syncthreads();
    const int last = nkb - 1;
    for (int kb = 0; kb < nkb; kb += 2) {
        {
            const int nx = kb + 1 < nkb ? kb + 1 : last;  // clamped (never predicated) prefetch
            load_a(nx);
            load_b(b1, nx);
            // pin the prefetch in front of the MFMAs: left alone, the scheduler sinks these loads to the end of the
            // iteration, right in front of the next iteration's wait for them (one exposed L2/HBM latency per k-block)
            __builtin_amdgcn_sched_barrier(0);
            compute(b0, kb, 0);
            write_a(1);
            __syncthreads();
        }
        if (kb + 1 < nkb) {
            const int nx = kb + 2 < nkb ? kb + 2 : last;
            load_a(nx);
            load_b(b0, nx);
            __builtin_amdgcn_sched_barrier(0);
            compute(b1, kb + 1, 1);
            write_a(0);
            __syncthreads();
        }
    }

    // ---- epilogue: D[i = 4*(lane>>4) + r][j = lane & 15] of each 16x16 tile ----
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n_base + j * 16 + n16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + i * 16 + q * 4 + r;
                if (m < g.M && n < g.N) {
                    half_t *c = g.C + (size_t)m * g.ldc + n;
                    *c = g.add_to_c ? (half_t)(*c + (half_t)acc[i][j][r]) : (half_t)acc[i][j][r];
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Variant (C-ABI tile ids 100 + m_tiles): the PACKED weights go through LDS too, 4 KiB per 64 rows and k-block.  Not to
// share them but because of how an MFMA operand fragment is read: it puts output column n = lane % 16 and k-slice
// lane / 16 in a lane, so a fragment loaded straight from a K-contiguous matrix makes the 16 lanes of every
// quarter-wave read 16 different rows (64 cache-line look-ups per wave load instead of 8).  Here a thread loads 16 bytes
// such that 4 consecutive lanes read one row's 64 bytes, the tile is written to LDS with an XOR swizzle
// (slot = row*4 + (chunk ^ (row/4 % 4)): fragment reads touch all 16 bank groups), each lane reads its fragment word
// with one ds_read_b128, and the scales and packed zeros of the workgroup's rows are staged into LDS once.
// Measured (profiles/r1/gemm_pipeline_attempts.md): faster than the direct form for 32-row tiles (426 vs 380 TF at
// M=512, 4096^2), slower for the 64-row tiles the dispatcher prefers (449 vs 512-536 TF) -- the look-up count is not
// what bounds this loop either.  Kept as a selectable variant; the dispatcher uses the direct form.
// ---------------------------------------------------------------------------------------------------------------------
template <int MT, int NT>
__global__ __launch_bounds__(256, (MT * NT >= 16 ? 1 : 2)) void w4a16_gemm_lds_kernel(const GemmArgs g) {
    constexpr int BM = MT * 16;
    constexpr int BN = 4 * NT * 16;
    constexpr int BK = 128;
    constexpr int ABUF = MT * 4 * 64;  // 16-byte pieces of one activation tile image
    constexpr int WBUF = BN * 4;       // 16-byte pieces of one packed weight tile (64 bytes per row)
    extern __shared__ __attribute__((aligned(16))) uint4_t lds_dyn[];
    uint4_t *lds_a = lds_dyn;                  // [2][ABUF]
    uint4_t *lds_w = lds_dyn + 2 * ABUF;       // [2][WBUF]
    const int nkb = g.K / BK;
    const int zw = g.zeros_stride;
    unsigned *lds_z = reinterpret_cast<unsigned *>(lds_w + 2 * WBUF);  // [BN][zw]
    half_t *lds_s = reinterpret_cast<half_t *>(lds_z + BN * zw);       // [BN][nkb]

    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    const int m_blk = (xcd % g.xm) * g.m_per + slot % g.m_per;
    const int n_blk = (xcd / g.xm) * g.n_per + slot / g.m_per;
    if (n_blk >= g.n_blocks || m_blk >= g.m_blocks) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15;
    const int q = lane >> 4;
    const int m_base = m_blk * BM;
    const int nb0 = n_blk * BN;  // first weight row of the workgroup
    const int nchunks = g.K >> 5;

    // ---- scales and zeros of the workgroup's rows, all k-blocks, once ----
    for (int idx = tid; idx < BN * nkb; idx += 256) {
        const int row = idx / nkb, gi = idx - row * nkb;
        int n = nb0 + row;
        n = n < g.N ? n : g.N - 1;
        lds_s[idx] = g.scales[(size_t)n * g.scales_stride + gi];
    }
    for (int idx = tid; idx < BN * zw; idx += 256) {
        const int row = idx / zw, wi = idx - row * zw;
        int n = nb0 + row;
        n = n < g.N ? n : g.N - 1;
        lds_z[idx] = g.zeros[(size_t)n * g.zeros_stride + wi];
    }

    // ---- staging registers: MT activation pieces and NT weight pieces of 16 bytes per thread, two sets ----
    struct Tiles {
        uint4_t a[MT], w[NT];
    };
    auto load_tiles = [&](Tiles &t, int kb) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int p = i * 256 + tid;  // piece: row p/4 of the workgroup, 16-byte chunk p%4 of the k-block
            int n = nb0 + (p >> 2);
            n = n < g.N ? n : g.N - 1;
            t.w[i] = g.qweight[(size_t)n * nchunks + kb * 4 + (p & 3)];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int e = i * 256 + tid;
            const int row = e >> 4, pc = e & 15;
            int m = m_base + row;
            m = m < g.M ? m : g.M - 1;
            t.a[i] = *reinterpret_cast<const uint4_t *>(g.A + (size_t)m * g.lda + kb * BK + pc * 8);
        }
    };
    auto write_tiles = [&](const Tiles &t, int buf) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int p = i * 256 + tid;
            const int row = p >> 2, c = p & 3;
            lds_w[buf * WBUF + row * 4 + (c ^ ((row >> 2) & 3))] = t.w[i];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int e = i * 256 + tid;
            const int row = e >> 4, pc = e & 15;
            const int qq = pc >> 2, s = pc & 3;  // k offset 8*pc = 32*qq + 8*s
            lds_a[buf * ABUF + ((row >> 4) * 4 + s) * 64 + qq * 16 + ((row & 15) ^ s ^ (qq << 2))] = pair_permute(t.a[i]);
        }
    };

    const NibbleMasks nmask = make_nibble_masks();
    float4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int kb, int buf) {
        uint4_t bw[NT];
        ZeroPair zp[NT];
        float sc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int row = (wave * NT + j) * 16 + n16;  // the lane's weight row within the workgroup
            bw[j] = lds_w[buf * WBUF + row * 4 + (q ^ ((row >> 2) & 3))];
            zp[j] = make_zero_pair((lds_z[row * zw + (kb >> 3)] >> ((kb & 7) * 4)) & 0xFu);
            sc[j] = (float)lds_s[row * nkb + kb];
        }
        float4_t blk[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) blk[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            half8_t bf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                half2_t d[4];
                dequant_word(bw[j][s], zp[j], nmask, d);
                bf[j] = half8_t{d[0].x, d[0].y, d[1].x, d[1].y, d[2].x, d[2].y, d[3].x, d[3].y};
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint4_t araw = lds_a[buf * ABUF + (i * 4 + s) * 64 + q * 16 + (n16 ^ s ^ (q << 2))];
                const half8_t af = __builtin_bit_cast(half8_t, araw);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    blk[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], blk[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(sc[j], blk[i][j][r], acc[i][j][r]);
    };

    // Double-buffered LDS, one barrier per k-block, and TWO iterations of cover for the global loads: an iteration is
    // only MT*NT*4 MFMAs, and with one iteration of cover a lone workgroup ran at 0.7 us per k-block (M = 64: 22.7 us
    // for 32 k-blocks on an idle chip).  The tiles of block kb+2 are requested at the top of iteration kb into the
    // register set block kb vacated and written to LDS at the end of iteration kb+1.  Everything a step waits for was
    // requested one step earlier and is consumed at one place (the LDS writes), so the wait is a plain counted vmcnt and
    // exactly one set is in flight at the loop header on both paths into it.
    const int last = nkb - 1;
    Tiles t0, t1;
    load_tiles(t0, 0);
    load_tiles(t1, nkb > 1 ? 1 : 0);
    write_tiles(t0, 0);
    __syncthreads();
    for (int kb = 0; kb < nkb; kb += 2) {
        {
            load_tiles(t0, kb + 2 < nkb ? kb + 2 : last);  // clamped, never predicated
            __builtin_amdgcn_sched_barrier(0);
            compute(kb, 0);
            __builtin_amdgcn_sched_barrier(0);
            write_tiles(t1, 1);  // block kb+1
            __syncthreads();
        }
        if (kb + 1 < nkb) {
            load_tiles(t1, kb + 3 < nkb ? kb + 3 : last);
            __builtin_amdgcn_sched_barrier(0);
            compute(kb + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            write_tiles(t0, 0);  // block kb+2
            __syncthreads();
        }
    }

    const int n_base = nb0 + wave * (NT * 16);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n_base + j * 16 + n16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m_base + i * 16 + q * 4 + r;
                if (m < g.M && n < g.N) {
                    half_t *c = g.C + (size_t)m * g.ldc + n;
                    *c = g.add_to_c ? (half_t)(*c + (half_t)acc[i][j][r]) : (half_t)acc[i][j][r];
                }
            }
        }
}

void set_xcd_grid(GemmArgs &g) {
    g.xm = g_gemm_xm;
    g.m_per = (g.m_blocks + g.xm - 1) / g.xm;
    const int xn = 8 / g.xm;
    g.n_per = (g.n_blocks + xn - 1) / xn;
}

template <int MT, int NT>
hipError_t launch_gemm_lds(const GemmArgs &g0, hipStream_t stream) {
    GemmArgs g = g0;
    constexpr int BM = MT * 16, BN = 4 * NT * 16;
    g.n_blocks = (g.N + BN - 1) / BN;
    g.m_blocks = (g.M + BM - 1) / BM;
    const int nkb = g.K / 128;
    size_t lds = (size_t)(2 * MT * 4 * 64 + 2 * BN * 4) * 16 + (size_t)BN * g.zeros_stride * 4 + (size_t)BN * nkb * 2;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kfn = w4a16_gemm_lds_kernel<MT, NT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    set_xcd_grid(g);
    hipLaunchKernelGGL(kfn, dim3(8 * g.m_per * g.n_per), dim3(256), lds, stream, g);
    return hipGetLastError();
}

template <int MT, int NT>
hipError_t launch_gemm(const GemmArgs &g0, hipStream_t stream) {
    GemmArgs g = g0;
    constexpr int BM = MT * 16, BN = 4 * NT * 16;
    g.n_blocks = (g.N + BN - 1) / BN;
    g.m_blocks = (g.M + BM - 1) / BM;
    set_xcd_grid(g);
    hipLaunchKernelGGL((w4a16_gemm_kernel<MT, NT>), dim3(8 * g.m_per * g.n_per), dim3(256), 0, stream, g);
    return hipGetLastError();
}

}  // namespace

void set_gemm_xcd_rows(int xm) { g_gemm_xm = (xm == 2 || xm == 4 || xm == 8) ? xm : 1; }

bool gemm_variant_exists(int mt, int nt) {
#define TCE_V(M_, N_) \
    if ((mt == M_ || mt == 100 + M_) && nt == N_) return true;  // 100 + m_tiles: packed weights staged through LDS
    TCE_GEMM_VARIANTS(TCE_V)
#undef TCE_V
    return false;
}

int launch_w4a16_gemm(const tce_w4a16_desc &d, int forced_mt, int forced_nt, hipStream_t stream, hipError_t *hip_err) {
    if (d.K % 128 != 0 || d.group_size != 128) return TCE_ERR_UNSUPPORTED_SHAPE;  // caller falls back to the GEMV kernel
    GemmArgs g{};
    const int zw = zeros_width(d.K, d.group_size);
    g.A = static_cast<const half_t *>(d.A);
    g.qweight = static_cast<const uint4_t *>(d.qweight);
    g.scales = static_cast<const half_t *>(d.scales);
    g.zeros = static_cast<const unsigned *>(d.zeros);
    g.C = static_cast<half_t *>(d.C);
    g.M = d.M;
    g.N = d.N;
    g.K = d.K;
    g.lda = d.lda ? d.lda : d.K;
    g.ldc = d.ldc ? d.ldc : d.N;
    g.add_to_c = (d.flags & TCE_W4_ADD_TO_C) ? 1 : 0;
    g.scales_stride = d.scales_stride ? d.scales_stride : zw * 8;
    g.zeros_stride = d.zeros_stride ? d.zeros_stride : zw;
    g.log2g = d.group_size == 128 ? 7 : (d.group_size == 64 ? 6 : 5);
    int mt = forced_mt, nt = forced_nt;
    if (mt == 0) {
        // tile choice (measured on MI355X, profiles/): small M -> small row tiles; otherwise 128x64 tiles when that
        // still yields >= 2 workgroups per CU, else 64x64
        if (d.M <= 32) { mt = 2; nt = 2; }
        else if (d.M <= 64) { mt = 4; nt = 1; }
        else {
            // 64x128 tiles (each wave 4x2 MFMA tiles: the dequantized fragment feeds 8 MFMAs) once they still give
            // >= 2 workgroups per CU, else 64x64 (measured, profiles/r1/gemm_m512_sweep.jsonl)
            const long blocks_42 = (long)((d.M + 63) / 64) * ((d.N + 127) / 128);
            if (blocks_42 >= 512) { mt = 4; nt = 2; }
            else { mt = 4; nt = 1; }
        }
    }
    hipError_t e = hipSuccess;
    bool found = false;
#define TCE_V(M_, N_)                                                                     \
    if (!found && (mt == M_ || mt == 100 + M_) && nt == N_) {                              \
        found = true;                                                                     \
        e = mt >= 100 ? launch_gemm_lds<M_, N_>(g, stream) : launch_gemm<M_, N_>(g, stream); \
    }
    TCE_GEMM_VARIANTS(TCE_V)
#undef TCE_V
    if (!found) return TCE_ERR_BAD_ARG;
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
