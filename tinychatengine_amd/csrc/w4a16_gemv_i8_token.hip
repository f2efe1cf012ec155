// w4a16_gemv_i8_token.hip -- a decode token's dependent GEMV launches as ONE persistent kernel on the int8-contraction body (round 6).
//
// What it replaces: the stream-ordered plan issues 4 launches per decoder block (the reference 5: llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu:73-115 through
// llm/src/ops/cuda/linear.cu:36); each dependent launch pays a kernel boundary (~1.7 us), the first-byte latency of its weights and a fill / drain during which HBM idles.
// Rounds 2 / 3 measured two ways around that (w4a16_gemv_stream.hip's token kernel, w4a16_gemv_ovl.hip) on the fp16-unpack body, whose arithmetic took as long as its
// stream: both lost.  Round 4's body (w4a16_gemv_i8.hip) contracts exact int8 digit planes on the matrix pipe: the un-hidden arithmetic of a launch is well under a
// microsecond, so a wave that asks for its weights BEFORE it waits for the activations hides the whole hand-off behind the stream.  This file is that experiment.
//
//   * GRID        one workgroup of 16 waves per CU, all resident (the host checks the occupancy); a stage (= one launch of the list) deals its 16-row tiles to the
//                 workgroups round-robin (tile t -> workgroup t % P); a tile's K/1024 k-chunks are units; wave w of a workgroup takes its units w, w + 16, ...
//   * ARITHMETIC  w4a16_gemv_i8_kernel<1, 1, 1, 8, true>'s, operation for operation: chunk c's activations are converted by wave c of every workgroup (one block exponent
//                 per 1024-k chunk, four balanced digit planes in LDS), a unit is 16 v_mfma_i32_16x16x64_i8 + one fp32 fma per (row, group), the planes meet through DPP
//                 adds, a tile's chunks are added in chunk order by the wave that finishes the tile's last unit.  Same bits as the stream-ordered plan.
//   * HAND-OFF    every output is also written as a 32-bit word (token tag << 16 | fp16 bits) into a shadow vector the plan owns (four words per 16-byte write-through
//                 store); the stage that consumes it polls THE DATA with coherent loads -- no counter, no flag, no acknowledgement (the protocol of round 2's token
//                 kernel: w4a16_gemv_stream.hip, 70 000 replays across the tag wrap in tests/test_gpu_chain.py).
//   * ORDER       a wave that has finished its units of stage s reads stage s + 1's record, REQUESTS THE WEIGHTS of its first unit there (they depend on nothing), and only
//                 then -- if it converts a chunk -- polls for the activations.  Vector-memory results return in order, so the poll's answer arrives behind the weights; that
//                 costs nothing the unit would not have waited for anyway (it needs both).
//   * TWO BARRIERS per stage, both s_barrier without a vector-memory drain: A (nobody reads the previous stage's planes any more) in front of the conversion, B behind it.
//
// Scope (anything else: the plan is built stream-ordered, or the rest of the list follows the kernel as ordinary launches): M = 1, groups of 128, every linear with
// TCE_W4_ZERO_POINT_IS_8 and a packed copy, K <= 15360, at most kTokMaxUnits units and kTokMaxTiles tiles per workgroup and stage; epilogues: plain, TCE_W4_SILU_MUL_PAIRS,
// TCE_W4_ADD_TO_C.  No fused RMSNorm prologue (the headline list has none).
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"
#include "w4a16_mfma_layout.hpp"

#include <new>
#include <vector>

namespace tce {

namespace {

constexpr int kTokWaves = 16;
constexpr int kTokMaxChunks = 15;   // planes of 15 chunks (60 KiB) beside the partial rows
constexpr int kTokMaxUnits = 512;   // per workgroup and stage: 32 KiB of partial rows
constexpr int kTokMaxTiles = 128;   // per workgroup and stage

struct TokSeg {
    const void *words;      // u32 [NT16][U][64][4]
    const half_t *dscales;  // fp16 [NT16][U][16]
    half_t *C;
    unsigned *C_tag;        // (tag << 16 | bits) words, null: nobody inside the plan reads this output
    int N, epilogue, tile_begin, bytes_w, bytes_s, pad;
};
static_assert(sizeof(TokSeg) == 56, "record layout");
struct TokStage {
    const half_t *A;
    const unsigned *A_tag;  // null: the activations come from outside the plan (read at once)
    int K, U, nch, nseg, ntiles, pad;
    TokSeg seg[TCE_MAX_GROUP];
};
constexpr int kRecWords = sizeof(TokStage) / 4;
static_assert(sizeof(TokStage) % 8 == 0 && kRecWords <= 128, "a record is fetched by one wave, 8 bytes per lane");

struct TokArgs {
    const TokStage *stages;  // device memory
    int n_stages;
    unsigned *status;        // set to 1 if a wait timed out
    const unsigned *epoch;   // the token's tag (1..65535)
    unsigned long long *dbg; // stamps [workgroup][stage][8] (100 MHz), null normally
};

// LDS
constexpr int kPlanesBytes = kTokMaxChunks * 4096;                  // [chunk][8 units][2 halves][4 planes][4 kq][4 words]
constexpr int kRecOff = kPlanesBytes;                               // TokStage[3]
constexpr int kShOff = kRecOff + 3 * (int)sizeof(TokStage);         // int sh[16], int bad[16]
constexpr int kCntOff = kShOff + 128;                               // unsigned tile_cnt[kTokMaxTiles]
constexpr int kRedOff = kCntOff + kTokMaxTiles * 4;                 // float red[kTokMaxUnits][16]
constexpr int kLdsBytes = kRedOff + kTokMaxUnits * 64;

template <int DPP_CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned tok_dpp_max_u32(unsigned v) {
    const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, DPP_CTRL, ROW_MASK, 0xF, false);
    return v > t ? v : t;
}
template <int DPP_CTRL>
__device__ __forceinline__ float tok_dpp_add_f32(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), DPP_CTRL, 0xF, 0xF, false);
    return v + __builtin_bit_cast(float, t);
}
template <int DPP_CTRL>
__device__ __forceinline__ unsigned tok_dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, DPP_CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T *sgpr_ptr(T *p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}

// Two other forms were built on this body, measured and removed (profiles/r6/i8_token_kernel.md; both bit-identical, the code is in commit 2cfe49f):
//   * units dealt to the non-converting waves first, the converting waves polling IN FRONT of their own requests (a poll's answer otherwise queues behind the wave's 8 KiB
//     of weights: results return in order) -- 1.09 ms per Llama-3-8B token against 0.976: the answer still queues behind the chip's 32 MiB of requests in the memory system,
//     the converting waves' own units start a memory latency late, and waves that sweep flat out take bandwidth from the producers;
//   * no barriers: every wave polls and converts the chunk of ITS unit (the stream-ordered kernel's structure inside one launch) -- 1.139 ms: sixteen waves per CU convert
//     and poll where four did.
template <bool STAMPS>
__global__ __launch_bounds__(64 * kTokWaves) void w4a16_gemv_i8_token_kernel(const TokArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, j16 = lane & 15;
    const int p_l = j16 & 3, uu_l = j16 >> 2;
    const int P = gridDim.x, b = blockIdx.x;
    const int n = args.n_stages;
    unsigned *const planes_all = reinterpret_cast<unsigned *>(smem);
    TokStage *const rec = reinterpret_cast<TokStage *>(smem + kRecOff);
    int *const shx = reinterpret_cast<int *>(smem + kShOff);
    unsigned *const tile_cnt = reinterpret_cast<unsigned *>(smem + kCntOff);
    float *const red = reinterpret_cast<float *>(smem + kRedOff);
    const unsigned tag = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(args.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) << 16;
    unsigned long long *const stamps = STAMPS && args.dbg ? args.dbg + (size_t)b * n * 8 : nullptr;
    auto stamp = [&](int s, int i) {
        if constexpr (STAMPS) {
            if (stamps && tid == 0) stamps[s * 8 + i] = wall_clock64();
        }
    };

    if (tid < kRecWords) reinterpret_cast<unsigned *>(&rec[0])[tid] = reinterpret_cast<const unsigned *>(args.stages)[tid];
    __syncthreads();

    // a unit's registers: 8 KiB of weights per wave, the lane's scales of both passes
    uint4_t wq[8];
    uint2_t sc[2];
    const int4_t zero4 = int4_t{0, 0, 0, 0};

    for (int s = 0; s < n; ++s) {
        const TokStage *R = &rec[s % 3];
        const int K = sgpr(R->K), U = sgpr(R->U), nch = sgpr(R->nch), nseg = sgpr(R->nseg), ntiles = sgpr(R->ntiles);
        const int ntw = b < ntiles ? (ntiles - 1 - b) / P + 1 : 0;  // this workgroup's tiles: b, b + P, ...
        const int nunits = ntw * nch;
        stamp(s, 0);

        // the next stage's record, requested in front of everything else: EVERY wave fetches it (264 bytes, 8 per lane, lanes past the end repeat the last pair) and
        // every wave writes it into LDS between the barriers -- the same values sixteen times.  Unconditional on purpose: a load under `if (wave == 15)` whose use sits
        // under a second `if (wave == 15)` leaves hipcc's wait-count pass with a path on which the load is still in flight at the loop head, and it answers with
        // s_waitcnt vmcnt(0) there -- in front of the next stage's weight requests, behind this stage's write-through stores.
        const int rec_pair = (lane * 2 < kRecWords ? lane * 2 : kRecWords - 2);
        const uint2_t recv = *reinterpret_cast<const uint2_t *>(reinterpret_cast<const unsigned *>(args.stages + (s + 1 < n ? s + 1 : n - 1)) + rec_pair);

        // ---- a unit's requests ----
        auto seg_of = [&](int t) {
            int si = 0;
            for (int q = 1; q < nseg; ++q)
                if (t >= sgpr(R->seg[q].tile_begin)) si = q;
            return si;
        };
        auto request = [&](int u) {  // u: index among this workgroup's units (wave-uniform)
            const int i = u / nch, c = u - i * nch;
            const int t = b + i * P;
            const TokSeg *S = &R->seg[seg_of(t)];
            const int tl = t - sgpr(S->tile_begin);
            const int u0 = c * 8;
            const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(sgpr_ptr(S->words)), 0, sgpr(S->bytes_w), 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(sgpr_ptr(S->words)), 0, 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(sgpr_ptr(S->dscales)), 0, sgpr(S->bytes_s), 0x00020000);
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int ga = tl * U + u0 + ps * 4 + uu_l;  // groups past K (a ragged last chunk) read the next tile's values or zeros: their products are zero
                sc[ps] = __builtin_bit_cast(uint2_t, __builtin_amdgcn_raw_buffer_load_b64(rs_s, (ga * 16 + 4 * kq) * 2, 0, 0));
            }
#pragma unroll
            for (int tt = 0; tt < 8; ++tt)
                wq[tt] = __builtin_amdgcn_raw_buffer_load_b128(u0 + tt < U ? rs_w : rs_none, lane * 16, (tl * U + u0 + tt) * 1024, /*nt*/ 2);
        };
        int u_cur = w;
        const int u_first = u_cur;
        if (u_cur < nunits) request(u_cur);
        __builtin_amdgcn_sched_barrier(0);

        // ---- the activations of chunk w (waves 0 .. nch - 1) ----
        uint4_t xv[2];
        xv[0] = xv[1] = uint4_t{0u, 0u, 0u, 0u};
        if (w < nch) {
            const unsigned *A_tag = sgpr_ptr(R->A_tag);
            if (A_tag == nullptr) {
                const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(sgpr_ptr(R->A)), 0, K * 2, 0x00020000);
#pragma unroll
                for (int c = 0; c < 2; ++c) xv[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, w * 2048 + (lane + 64 * c) * 16, 0, 0);
            } else {
                const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(A_tag), 0, K * 4, 0x00020000);
                bool need[2];
                uint4_t lo[2], hi[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    need[c] = w * 1024 + (lane + 64 * c) * 8 < K;  // pieces past K (a ragged last chunk): zeros, nothing to wait for
                    lo[c] = hi[c] = uint4_t{tag, tag, tag, tag};
                }
                int tries = 0;
                for (;;) {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        if (need[c]) {
                            lo[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, w * 4096 + (lane + 64 * c) * 32, 0, /*sc0|sc1*/ 17);
                            hi[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, w * 4096 + (lane + 64 * c) * 32 + 16, 0, /*sc0|sc1*/ 17);
                        }
                    bool again = false;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        if (need[c]) {
                            const unsigned bad = ((lo[c].x ^ tag) | (lo[c].y ^ tag) | (lo[c].z ^ tag) | (lo[c].w ^ tag) | (hi[c].x ^ tag) | (hi[c].y ^ tag) | (hi[c].z ^ tag) |
                                                  (hi[c].w ^ tag)) >> 16;
                            need[c] = bad != 0u;
                            again |= need[c];
                        }
                    if (!__builtin_amdgcn_ballot_w64(again)) break;
                    __builtin_amdgcn_s_sleep(2);
                    if ((++tries & 255) == 0) {  // has anybody given up?  ~0.3 s: give up (the plan's status word says so)
                        if (__hip_atomic_load(args.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                        if (tries > (1 << 18)) {
                            __hip_atomic_store(args.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    xv[c] = uint4_t{(lo[c].x & 0xFFFFu) | (lo[c].y << 16), (lo[c].z & 0xFFFFu) | (lo[c].w << 16), (hi[c].x & 0xFFFFu) | (hi[c].y << 16),
                                    (hi[c].z & 0xFFFFu) | (hi[c].w << 16)};
                // (a piece past K keeps the tag words it was initialised with: its halves are the tag's low 16 bits = 0)
            }
            // the activations are IN REGISTERS when this block ends.  Without this use hipcc's wait-count pass sees a path on which the loads above are still in flight at
            // the next stage's head (this block taken, the conversion block below not) and puts s_waitcnt vmcnt(0) THERE -- in front of the next stage's weight requests,
            // behind this stage's write-through stores (1-2 us to their acknowledgement): measured as workgroups entering a stage 2-3 us after the previous one's end.
            asm volatile("" : "+v"(xv[0]), "+v"(xv[1]));
        }
        stamp(s, 1);
        lds_barrier();  // A: every wave of the workgroup is through the previous stage (its planes, its partial rows)
        stamp(s, 2);
        if (tid < kTokMaxTiles) tile_cnt[tid] = 0u;
        if (w < nch) {
            // w4a16_gemv_i8_kernel's conversion (MB = 1, UW = 8): one block exponent per chunk, balanced base-256 digits, byte-transposed into the B operand image
            unsigned mx = 0;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned v = xv[c][q] & 0x7FFF7FFFu;
                    const unsigned v2 = v << 16;
                    const unsigned m2 = v > v2 ? v : v2;
                    mx = mx > m2 ? mx : m2;
                }
            mx = tok_dpp_max_u32<0x111>(mx);
            mx = tok_dpp_max_u32<0x112>(mx);
            mx = tok_dpp_max_u32<0x114>(mx);
            mx = tok_dpp_max_u32<0x118>(mx);
            mx = tok_dpp_max_u32<0x142, 0xA>(mx);
            mx = tok_dpp_max_u32<0x143, 0xC>(mx);
            const int E = (int)((unsigned)__builtin_amdgcn_readlane((int)mx, 63) >> 26);
            const int sh = 44 - E;
            if (lane == 0) {
                shx[w] = sh;
                shx[16 + w] = E == 31 ? 1 : 0;
            }
            const float scale = __builtin_bit_cast(float, (unsigned)(127 + sh) << 23);
            unsigned *planes = planes_all + w * 1024;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                unsigned d[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const half_t lo = __builtin_bit_cast(half_t, (unsigned short)(xv[c][q] & 0xFFFFu));
                    const half_t hi = __builtin_bit_cast(half_t, (unsigned short)(xv[c][q] >> 16));
                    d[2 * q] = ((unsigned)(int)__builtin_fmaf((float)lo, scale, 0.0f) + 0x00808080u) ^ 0x00808080u;
                    d[2 * q + 1] = ((unsigned)(int)__builtin_fmaf((float)hi, scale, 0.0f) + 0x00808080u) ^ 0x00808080u;
                }
                const int cc = lane + 64 * c;
                const int tu = cc >> 4, sw = (cc >> 2) & 3, q4 = cc & 3;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned a0 = d[2 * h], a1 = d[2 * h + 4], a2 = d[2 * h + 1], a3 = d[2 * h + 5];
                    const unsigned t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u);
                    const unsigned t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
                    const unsigned t2 = __builtin_amdgcn_perm(a3, a2, 0x05010400u);
                    const unsigned t3 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
                    const int base = (((tu * 2 + h) * 4) * 4 + q4) * 4 + sw;  // + plane * 16
                    planes[base + 0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
                    planes[base + 16] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
                    planes[base + 32] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
                    planes[base + 48] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
                }
            }
        }
        lds_barrier();  // B: the planes of every chunk, the exponents, the zeroed tile counters
        stamp(s, 3);
        // the next stage's record into LDS: every wave writes the same values and reads them behind its own write, so no barrier orders it; three buffers because a wave
        // of this workgroup may still be reading stage s - 1's record (its units) while another is here.  HERE because the write waits for the fetch and with it (results
        // return in order) for everything this wave has requested since: behind barrier B that is the weights the wave is about to use anyway; in front of the conversion
        // it held the conversion up by the weights' latency, at the stage's end by the acknowledgement of the stage's write-through stores.
        *reinterpret_cast<uint2_t *>(reinterpret_cast<unsigned *>(&rec[(s + 1) % 3]) + rec_pair) = recv;

        // ---- this wave's units ----
        while (u_cur < nunits) {
            const int i = u_cur / nch, c = u_cur - i * nch;
            const unsigned *planes = planes_all + c * 1024;
            int4_t B[4][2];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) B[uu][0] = B[uu][1] = zero4;
            auto read_b = [&](int ps) {
#pragma unroll
                for (int uu = 0; uu < 4; ++uu)
                    if (uu_l == uu) {  // exec-masked reads: the operand of unit uu of the pass in the lanes of ITS columns, zeros in the others
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const unsigned *src = planes + (((((ps * 4 + uu) * 2 + h) * 4 + p_l) * 4 + kq) * 4);
                            B[uu][h] = __builtin_bit_cast(int4_t, *reinterpret_cast<const uint4_t *>(src));
                        }
                    }
            };
            read_b(0);
            if constexpr (STAMPS) {  // (diagnostics: the unit's weights have landed / its arithmetic is done / its partial row is counted in)
                if (u_cur == u_first) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    stamp(s, 5);
                }
            }
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                int4_t dd0 = zero4, dd1 = zero4;
#pragma unroll
                for (int uu = 0; uu < 4; ++uu) {
                    const int t = ps * 4 + uu;
                    int4_t alo, ahi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned x = wq[t][q];
                        ahi[q] = (int)((x & 0xF0F0F0F0u) ^ 0x80808080u);
                        alo[q] = (int)(((x << 4) & 0xF0F0F0F0u) ^ 0x80808080u);
                    }
                    dd0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(alo, B[uu][0], dd0, 0, 0, 0);
                    dd1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ahi, B[uu][1], dd1, 0, 0, 0);
                }
                const half_t s0 = __builtin_bit_cast(half_t, (unsigned short)(sc[ps][0] & 0xFFFFu));
                const half_t s1 = __builtin_bit_cast(half_t, (unsigned short)(sc[ps][0] >> 16));
                const half_t s2 = __builtin_bit_cast(half_t, (unsigned short)(sc[ps][1] & 0xFFFFu));
                const half_t s3 = __builtin_bit_cast(half_t, (unsigned short)(sc[ps][1] >> 16));
                const int4_t tot = dd0 + dd1;
                acc[0] = __builtin_fmaf((float)tot[0], (float)s0, acc[0]);
                acc[1] = __builtin_fmaf((float)tot[1], (float)s1, acc[1]);
                acc[2] = __builtin_fmaf((float)tot[2], (float)s2, acc[2]);
                acc[3] = __builtin_fmaf((float)tot[3], (float)s3, acc[3]);
                if (ps == 0) read_b(1);
            }
            const int sh_c = sgpr(shx[c]);
            const bool bad_c = sgpr(shx[16 + c]) != 0;
            const float cj = bad_c ? __builtin_nanf("") : __builtin_bit_cast(float, (unsigned)(127 + 8 * p_l - sh_c - 4) << 23);
            float *slot = red + (size_t)u_cur * 16;
            if constexpr (STAMPS) {
                if (u_cur == u_first) {
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
                    stamp(s, 6);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = acc[q] * cj;
                v = tok_dpp_add_f32<0xB1>(v);
                v = tok_dpp_add_f32<0x4E>(v);
                v = tok_dpp_add_f32<0x128>(v);
                v = tok_dpp_add_f32<0x124>(v);
                if (j16 == 0) slot[kq * 4 + q] = v;
            }
            // the tile's chunks meet: the wave that brings the last one adds them in chunk order and stores
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(tile_cnt + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");  // (the partial rows are read behind the count, never ahead of it)
            const bool last = (unsigned)__builtin_amdgcn_readfirstlane((int)old) == (unsigned)(nch - 1);
            if constexpr (STAMPS) {
                if (u_cur == u_first) stamp(s, 7);
            }
            const int t = b + i * P;
            if (last) {
                if constexpr (STAMPS) {
                    if (stamps && lane == 0) stamps[s * 8 + 4] = wall_clock64();  // (the workgroup's latest tile's store: whichever wave brings it)
                }
                const TokSeg *S = &R->seg[seg_of(t)];
                const int tl = t - sgpr(S->tile_begin);
                const int N = sgpr(S->N), epi = sgpr(S->epilogue);
                half_t *C = sgpr_ptr(S->C);
                unsigned *C_tag = sgpr_ptr(S->C_tag);
                const float *parts = red + (size_t)(i * nch) * 16 + (lane & 15);
                float v = 0.f;
                for (int k2 = 0; k2 < nch; ++k2) v += parts[k2 * 16];
                const int row = tl * 16 + (lane & 15);
                half_t y = (half_t)v;
                const half_t y_other = __builtin_bit_cast(half_t, (unsigned short)tok_dpp_u32<0xB1>((unsigned)__builtin_bit_cast(unsigned short, y)));
                int idx = row;           // where the value goes
                bool live = lane < 16 && row < N;
                if (epi & TCE_W4_SILU_MUL_PAIRS) {
                    y = silu_mul_half(y, y_other);
                    idx = row >> 1;
                    live = live && (lane & 1) == 0;
                } else if (epi & TCE_W4_ADD_TO_C) {
                    // the residual's old value may have been written by another CU earlier in this kernel: a coherent read (the data flow orders it: the plan builder
                    // refuses lists in which this stage does not come after that writer through the tagged words)
                    unsigned short oldc = 0;
                    if (live) oldc = __hip_atomic_load(reinterpret_cast<unsigned short *>(C) + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    y = __builtin_bit_cast(half_t, oldc) + y;
                }
                const unsigned short bits = __builtin_bit_cast(unsigned short, y);
                if (live) {
                    if (epi & TCE_W4_ADD_TO_C) __hip_atomic_store(reinterpret_cast<unsigned short *>(C) + idx, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else C[idx] = y;
                }
                if (C_tag) {
                    // four tagged words per store (the words are single-copy atomic each: every one carries its own tag)
                    const unsigned word = tag | bits;
                    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(C_tag, 0, 0x7FFFFFF0, 0x00020000);
                    if (epi & TCE_W4_SILU_MUL_PAIRS) {
                        // eight outputs of the tile, in the even lanes: lanes 0 and 8 collect four each
                        const unsigned w1 = tok_dpp_u32<0x102>(word), w2 = tok_dpp_u32<0x104>(word), w3 = tok_dpp_u32<0x106>(word);  // row_shl 2 / 4 / 6
                        if (lane < 16 && (lane & 7) == 0 && row + 7 < N) {
                            __builtin_amdgcn_raw_buffer_store_b128(uint4_t{word, w1, w2, w3}, rs_c, idx * 4, 0, /*sc0|sc1: write-through*/ 17);
                        } else if (lane < 16 && row < N && (lane & 1) == 0 && !(((row & ~7) + 7) < N)) {
                            __hip_atomic_store(C_tag + idx, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    } else {
                        const unsigned w1 = tok_dpp_u32<0x101>(word), w2 = tok_dpp_u32<0x102>(word), w3 = tok_dpp_u32<0x103>(word);  // row_shl 1 / 2 / 3
                        if (lane < 16 && (lane & 3) == 0 && row + 3 < N) {
                            __builtin_amdgcn_raw_buffer_store_b128(uint4_t{word, w1, w2, w3}, rs_c, idx * 4, 0, /*sc0|sc1: write-through*/ 17);
                        } else if (lane < 16 && row < N && !(((row & ~3) + 3) < N)) {
                            __hip_atomic_store(C_tag + idx, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
            u_cur += kTokWaves;
            if (u_cur < nunits) request(u_cur);
        }
    }
}

__global__ void i8_token_epoch_kernel(unsigned *epoch) { *epoch = *epoch % 65535u + 1u; }

int tok_num_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}

thread_local unsigned long long *g_tok_stamps = nullptr;
thread_local int g_tok_mode = 0;  // 0: tagged plans take this kernel where the list allows; 1: never (round 2's token kernel instead)
thread_local int g_tok_max_units = kTokMaxUnits;  // a stage with more units per workgroup ends the prefix the kernel takes (A/B: lm_head inside / behind the kernel)

}  // namespace

struct I8TokenPlan {
    TokStage *stages = nullptr;  // device
    unsigned *sync = nullptr;    // device: [0] status, [1] the token's tag
    unsigned *shadow = nullptr;  // device: the outputs as tagged words
    int n = 0, blocks = 0;
    unsigned long long *stamps = nullptr;
};

void set_i8_token_stamps(void *buf) { g_tok_stamps = static_cast<unsigned long long *>(buf); }
void set_i8_token_mode(int mode) { g_tok_mode = mode == 1 ? 1 : 0; }
void set_i8_token_max_units(int u) { g_tok_max_units = u >= 1 && u <= kTokMaxUnits ? u : kTokMaxUnits; }

void i8_token_plan_destroy(I8TokenPlan *tp) {
    if (!tp) return;
    if (tp->stages) (void)hipFree(tp->stages);
    if (tp->sync) (void)hipFree(tp->sync);
    if (tp->shadow) (void)hipFree(tp->shadow);
    delete tp;
}

int i8_token_plan_stages(const I8TokenPlan *tp) { return tp ? tp->n : 0; }
int i8_token_plan_blocks(const I8TokenPlan *tp) { return tp ? tp->blocks : 0; }

// Takes the longest PREFIX of the launch list this kernel can run (n_taken launches; the caller issues the rest behind the kernel as ordinary launches, ordered by the
// stream).  TCE_ERR_UNSUPPORTED_SHAPE: not even two launches -- build the plan another way.
int i8_token_plan_create(const tce_w4a16_desc *descs, const int32_t *groups, int n_launches, I8TokenPlan **out, int *n_taken, hipError_t *hip_err) {
    if (g_tok_mode == 1) return TCE_ERR_UNSUPPORTED_SHAPE;
    const int cus = tok_num_cus();
    if (cus == 0) return TCE_ERR_HIP;
#ifdef TCE_LAB
    const void *kfn = g_tok_stamps ? reinterpret_cast<const void *>(w4a16_gemv_i8_token_kernel<true>) : reinterpret_cast<const void *>(w4a16_gemv_i8_token_kernel<false>);
#else
    const void *kfn = reinterpret_cast<const void *>(w4a16_gemv_i8_token_kernel<false>);  // (the instantiation with wall-clock stamps: the lab build, build.py --lab)
#endif
    hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    int per_cu = 0;
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64 * kTokWaves, kLdsBytes);
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    if (per_cu < 1) return TCE_ERR_UNSUPPORTED_SHAPE;
    const int P = cus;

    std::vector<TokStage> host;
    std::vector<int> first_desc;
    for (int l = 0, off = 0; l < n_launches; off += groups[l], ++l) {
        const tce_w4a16_desc &d0 = descs[off];
        bool ok = d0.M == 1 && d0.group_size == 128 && d0.K % 128 == 0 && d0.K >= 128 && (d0.K + 1023) / 1024 <= kTokMaxChunks && !d0.rmsnorm_gamma &&
                  (d0.lda == 0 || d0.lda == d0.K) && groups[l] <= TCE_MAX_GROUP;
        TokStage st{};
        int tiles = 0;
        for (int i = 0; i < groups[l] && ok; ++i) {
            const tce_w4a16_desc &d = descs[off + i];
            ok = d.M == 1 && d.K == d0.K && d.group_size == 128 && d.A == d0.A && d.prepacked && !(reinterpret_cast<uintptr_t>(d.prepacked) & 255) && (d.flags & TCE_W4_ZERO_POINT_IS_8) &&
                 !(d.flags & TCE_W4_FORCE_GEMM) && !d.rmsnorm_gamma && (long long)pk::nt16(d.N) * (d.K / 2) * 16 < (1LL << 31) && !(d.flags & ~(TCE_W4_ZERO_POINT_IS_8 | TCE_W4_SILU_MUL_PAIRS | TCE_W4_ADD_TO_C));
            if ((d.flags & TCE_W4_SILU_MUL_PAIRS) && (d.N % 2 != 0 || (d.flags & TCE_W4_ADD_TO_C))) ok = false;
            if (!ok) break;
            TokSeg &sg = st.seg[i];
            const unsigned char *base = static_cast<const unsigned char *>(d.prepacked);
            sg.words = base;
            sg.dscales = reinterpret_cast<const half_t *>(base + pk::dscales_offset(d.N, d.K, d.group_size));
            sg.C = static_cast<half_t *>(d.C);
            sg.C_tag = nullptr;
            sg.N = d.N;
            sg.epilogue = d.flags & (TCE_W4_SILU_MUL_PAIRS | TCE_W4_ADD_TO_C);
            sg.tile_begin = tiles;
            sg.bytes_w = (int)pk::words_bytes(d.N, d.K);
            sg.bytes_s = (int)pk::dscales_bytes(d.N, d.K, d.group_size);
            tiles += pk::nt16(d.N);
        }
        if (ok) {
            st.A = static_cast<const half_t *>(d0.A);
            st.A_tag = nullptr;
            st.K = d0.K;
            st.U = d0.K / 128;
            st.nch = (st.U + 7) / 8;
            st.nseg = groups[l];
            st.ntiles = tiles;
            for (int i = groups[l]; i < TCE_MAX_GROUP; ++i) st.seg[i] = st.seg[0];
            const int tiles_wg = (tiles + P - 1) / P;
            ok = tiles_wg <= kTokMaxTiles && tiles_wg * st.nch <= g_tok_max_units;
        }
        if (!ok) break;
        host.push_back(st);
        first_desc.push_back(off);
    }
    const int n = (int)host.size();
    if (n < 2) return TCE_ERR_UNSUPPORTED_SHAPE;

    I8TokenPlan *tp = new (std::nothrow) I8TokenPlan();
    if (!tp) return TCE_ERR_BAD_ARG;
    tp->n = n;
    tp->blocks = P;
#ifdef TCE_LAB
    tp->stamps = g_tok_stamps;
#endif
    auto n_out_of = [&](const TokStage &L, int i) { return (L.seg[i].epilogue & TCE_W4_SILU_MUL_PAIRS) ? L.seg[i].N / 2 : L.seg[i].N; };
    // the shadow vectors: one per linear of every stage (buffers reused from layer to layer must not alias)
    size_t words = 0;
    std::vector<size_t> base((size_t)n * TCE_MAX_GROUP, 0);
    for (int l = 0; l < n; ++l)
        for (int i = 0; i < host[l].nseg; ++i) {
            base[(size_t)l * TCE_MAX_GROUP + i] = words;
            words += ((size_t)n_out_of(host[l], i) + 63) & ~(size_t)63;
        }
    e = hipMalloc(reinterpret_cast<void **>(&tp->shadow), words * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(tp->shadow, 0, words * sizeof(unsigned));
    // who produces what: the latest earlier writer whose output holds the whole activation vector (w4a16_gemv_stream.hip's token_plan_create, same rules)
    std::vector<int> producer(n, -1);
    std::vector<std::vector<char>> consumed((size_t)n, std::vector<char>(TCE_MAX_GROUP, 0));
    bool refuse = false;
    for (int l = 0; l < n && e == hipSuccess && !refuse; ++l) {
        TokStage &L = host[l];
        const char *a0 = reinterpret_cast<const char *>(L.A), *a1 = a0 + (size_t)L.K * 2;
        for (int q = l - 1; q >= 0 && producer[l] < 0 && !refuse; --q)
            for (int i = 0; i < host[q].nseg; ++i) {
                const char *c0 = reinterpret_cast<const char *>(host[q].seg[i].C), *c1 = c0 + (size_t)n_out_of(host[q], i) * 2;
                if (a0 >= c0 && a1 <= c1) {
                    if ((a0 - c0) % 16 != 0) {
                        refuse = true;
                        break;
                    }
                    L.A_tag = tp->shadow + base[(size_t)q * TCE_MAX_GROUP + i] + (a0 - c0) / 2;
                    producer[l] = q;
                    consumed[q][i] = 1;
                    break;
                }
                if (a0 < c1 && a1 > c0) {
                    refuse = true;
                    break;
                }
            }
    }
    // every output gets its shadow (a later list built on this plan's buffers could consume any of them; writing the words costs one store per four outputs)
    for (int l = 0; l < n; ++l)
        for (int i = 0; i < host[l].nseg; ++i) host[l].seg[i].C_tag = tp->shadow + base[(size_t)l * TCE_MAX_GROUP + i];
    // anti- and output dependences on the plain buffers must follow the data flow (see token_plan_create)
    if (e == hipSuccess && !refuse) {
        const size_t nw64 = ((size_t)n + 63) / 64;
        std::vector<uint64_t> anc((size_t)n * nw64, 0);
        for (int l = 0; l < n; ++l)
            if (producer[l] >= 0) {
                const int q = producer[l];
                for (size_t k = 0; k < nw64; ++k) anc[l * nw64 + k] = anc[q * nw64 + k];
                anc[l * nw64 + (size_t)q / 64] |= 1ull << (q % 64);
            }
        for (int l = 1; l < n && !refuse; ++l)
            for (int i = 0; i < host[l].nseg && !refuse; ++i) {
                const char *w0 = reinterpret_cast<const char *>(host[l].seg[i].C), *w1 = w0 + (size_t)n_out_of(host[l], i) * 2;
                for (int q = 0; q < l; ++q) {
                    bool touches = false;
                    if (producer[q] < 0) {
                        const char *r0 = reinterpret_cast<const char *>(host[q].A), *r1 = r0 + (size_t)host[q].K * 2;
                        touches = r0 < w1 && r1 > w0;
                    }
                    for (int k = 0; k < host[q].nseg && !touches; ++k) {
                        const char *c0 = reinterpret_cast<const char *>(host[q].seg[k].C), *c1 = c0 + (size_t)n_out_of(host[q], k) * 2;
                        touches = c0 < w1 && c1 > w0;
                    }
                    if (touches && !((anc[l * nw64 + (size_t)q / 64] >> (q % 64)) & 1ull)) {
                        refuse = true;
                        break;
                    }
                }
            }
    }
    if (refuse) {
        i8_token_plan_destroy(tp);
        return TCE_ERR_UNSUPPORTED_SHAPE;
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&tp->stages), sizeof(TokStage) * n);
    if (e == hipSuccess) e = hipMemcpy(tp->stages, host.data(), sizeof(TokStage) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&tp->sync), sizeof(unsigned) * 2);
    if (e == hipSuccess) {
        const unsigned init[2] = {0u, 1u};
        e = hipMemcpy(tp->sync, init, sizeof(init), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        i8_token_plan_destroy(tp);
        return TCE_ERR_HIP;
    }
    *out = tp;
    *n_taken = n;
    return TCE_OK;
}

int i8_token_plan_enqueue(I8TokenPlan *tp, hipStream_t stream, hipError_t *hip_err) {
    TokArgs a;
    a.stages = tp->stages;
    a.n_stages = tp->n;
    a.status = tp->sync;
    a.epoch = tp->sync + 1;
    a.dbg = tp->stamps;
    const dim3 grid(tp->blocks), block(64 * kTokWaves);
#ifdef TCE_LAB
    if (tp->stamps) hipLaunchKernelGGL(w4a16_gemv_i8_token_kernel<true>, grid, block, kLdsBytes, stream, a);
    else
#endif
    hipLaunchKernelGGL(w4a16_gemv_i8_token_kernel<false>, grid, block, kLdsBytes, stream, a);
    hipLaunchKernelGGL(i8_token_epoch_kernel, dim3(1), dim3(1), 0, stream, tp->sync + 1);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

int i8_token_plan_status(I8TokenPlan *tp, unsigned *status, hipError_t *hip_err) {
    const hipError_t e = hipMemcpy(status, tp->sync, sizeof(unsigned), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
