// w4a16_gemv_stream.hip -- persistent W4A16 dequant-GEMV for gfx950 (MI355X), M = 1.
//
// Same math and data layout as w4a16_gemv.hip (reference kernels/cuda/gemv_cuda.cu:140-194 behind
// MatmulOperator::gemv_forward_cuda; read that file's header for the unpack / MFMA-diagonal scheme), different work
// distribution:
//
//   * ONE GENERATION   the grid is at most (CUs x workgroups-per-CU) workgroups, all resident at once: there is no
//                      second, nearly empty generation of workgroups (the 2-3 us tail of the workgroup-per-row-block
//                      kernel).  Workgroup b owns row groups b, b+G, b+2G, ... (a row group = ROWS consecutive output
//                      rows of one linear of the group; G = workgroups in the launch).
//   * WAVES PULL WORK  the waves of a workgroup take the workgroup's row groups from a counter in LDS, one at a time.
//                      A static split starves the young waves: the SIMD arbiter favours the oldest wave, so with equal
//                      shares wave slots 0-3 of a 16-wave workgroup finished at 34 us and slots 12-15 at 51 us on the
//                      128k-row lm_head (profiles/r1/timeline_stream.jsonl) -- the launch ended with one wave per SIMD
//                      and nothing to overlap its HBM latency with.
//   * ONE X IMAGE      the activation vector is staged into LDS once per workgroup (not once per 32 rows), together
//                      with the per-lane sums sum_k x_k of every 32-weight chunk that the zero-point correction needs
//                      (computed once on the matrix pipe instead of 8 MFMAs per step).
//   * RING             a wave keeps DEPTH steps (ROWS x 1 KiB of weights + scales/zeros each) in flight across row-group
//                      boundaries; the issue side and the compute side each walk their own (row group, step) cursor with
//                      "sticky" per-linear state in SGPRs, so nothing but the loaded registers travels through the ring.
//   * TOKEN KERNEL     (w4a16_gemv_token_kernel, TCE_PLAN_TAGGED / TCE_PLAN_CHAINED plans) ONE launch of the same workgroups
//                      walks a whole list of GEMV launches -- a decode token's linears -- and there is NO barrier between
//                      them.  Every output is ALSO written as one 32-bit word (token tag << 16 | fp16 bits) with a
//                      device-scope store into a per-launch shadow vector the plan owns; the launch that consumes it
//                      first issues the first DEPTH steps of its WEIGHTS (they depend on nothing) and then polls THE
//                      DATA with coherent 16-byte loads until all tags of a piece match -- the poll is the activation read,
//                      and no store acknowledgement, arrival counter or second round trip is on the path.  All workgroups
//                      are resident at once (the host checks the occupancy), so the wait cannot deadlock; a wait that lasts
//                      ~0.3 s gives up and flags the plan.  Round 1's form (arrival counter per launch, one polling lane,
//                      then the coherent read) cost 5.0 us per bare hand-off against 1.8 us for this one and 2.0 us for a
//                      kernel boundary (scripts/probes/handoff_probe.hip, profiles/r2/handoff_probe.jsonl) and is gone.
//                      Status: correct (bit-identical to the stream-ordered plan), NOT faster for a Llama-7B token -- 1.40
//                      vs 0.995 ms: DESIGN.md 3.1b has the per-launch timeline (profiles/r2/token_kernel_timeline.jsonl).
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

#include <new>
#include <vector>

namespace tce {

namespace {

struct StreamArgs {
    StreamLaunch launch;
    unsigned long long *dbg;  // MODE 2 only
};

struct TokenArgs {
    const StreamLaunch *launches;  // device memory
    unsigned long long *dbg;       // diagnostics: [workgroup][launch][8] wall-clock stamps (100 MHz), null normally
    int n_launches;
    unsigned *status;  // set to 1 if a wait timed out
    const unsigned *epoch;  // the token's tag (1..65535), advanced by a one-thread kernel behind this one
};

// What a launch of a token kernel needs beside its own record.
struct Barrier {
    unsigned *status;
    unsigned tag;  // this token's tag, already shifted into the upper half
    unsigned long long *stamps;  // diagnostics (null normally): thread 0 writes {entered, activations in registers, image staged, own wave done, workgroup done}
};

// One GEMV launch, executed by all waves of the (persistent) grid.  MODE: 0 = the GEMV.  Diagnostics
// (tce_w4a16_set_debug_mode): 1 = stream the weights only, 2 = GEMV + per-wave timestamps, 4 = unpack/MFMA only.
// CHAIN: 0 = a launch of its own, 2 = a launch inside the token kernel (tagged hand-off).  (1 was round 1's arrival-counter
// barrier: 5.0 us per bare hand-off against 1.8 us tagged -- scripts/probes/handoff_probe.hip -- and removed.)
template <int ROWS, int DEPTH, bool Z8, int MODE, int CHAIN = 0>
__device__ __forceinline__ void gemv_launch_body(const StreamLaunch &L, const Barrier bar, unsigned char *smem, unsigned long long *ts_x_ready) {
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = nthreads >> 6;
    const int G = gridDim.x;

    const int K = L.K;
    const int nseg = L.nseg;
    const int nchunks = K >> 5;           // 16-byte chunks per weight row
    const int T = (nchunks + 63) >> 6;    // steps per row group
    const int gshift = L.log2g - 5;       // chunk -> quantization group
    const int rowbytes = nchunks * 16;
    const int n_rg = L.n_rg;
    const half_t *A = L.A;
    // LDS: x image [T][4][64] x 16 B, chunk sums [T][64] floats, the workgroup's row-group counter
    int *rg_counter = reinterpret_cast<int *>(smem + (size_t)T * (4096 + 256));  // [64]; word 0 is the counter
    auto stamp = [&](int i) {
        if constexpr (CHAIN != 0) {
            if (bar.stamps && tid == 0) bar.stamps[i] = wall_clock64();
        }
    };
    stamp(0);
    if (tid == 0) *rg_counter = 0;
    __syncthreads();
    // The ticket is lane 0's atomic on word 0; the value is read (readfirstlane) one row group later.  All 64 lanes
    // execute the instruction, lanes 1..63 adding 0 to words of their own (64 banks, one LDS pass): a `lane == 0` branch
    // around it makes hipcc treat every cursor variable behind the join as divergent (descriptors in VGPRs, waterfall
    // loops around each buffer load: +40 VGPRs).
    auto grab = [&]() -> int {
        return __hip_atomic_fetch_add(rg_counter + lane, lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    // ---------------------------------------------------------------------------------------------------------------
    // issue side
    // ---------------------------------------------------------------------------------------------------------------
    struct Step {
        uint4_t w[ROWS];
        unsigned short s[ROWS];
        unsigned z[ROWS];
        int rg;  // wave-uniform: the row group of the step, -1 = padding past the end of the work
    };
    int i_rg = 0, i_t = 0;             // cursor: next step to issue
    bool i_done = false;               // the workgroup's row groups are used up
    int v_next = grab();               // the next row group's ticket (in lane 0), in flight
    int i_si = 0, i_begin = 0;         // sticky: the linear that contains i_rg
    int i_end = nseg > 1 ? L.seg[1].block_begin : n_rg;
    int i_N = L.seg[0].N, i_sstr = L.seg[0].scales_stride * 2, i_zstr = L.seg[0].zeros_stride * 4;
    // num_records is a constant just below 2 GiB: real extents were checked on the host, and a padding step uses an
    // offset above it, which the buffer unit answers with zeros and no memory request
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4_t *>(L.seg[0].qweight), 0, 0x7FFFFFF0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(L.seg[0].scales), 0, 0x7FFFFFF0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(L.seg[0].zeros), 0, 0x7FFFFFF0, 0x00020000);
    int so_w[ROWS], so_s[ROWS], so_z[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) so_w[i] = so_s[i] = so_z[i] = 0;

    auto issue = [&](Step &st) {
        if (!i_done && i_t == 0) {  // a new row group: take the ticket, ask for the next one, row offsets into SGPRs
            i_rg = blockIdx.x + __builtin_amdgcn_readfirstlane(v_next) * G;
            if (i_rg >= n_rg) i_done = true;
            else v_next = grab();
        }
        const bool live = !i_done;  // wave-uniform; past the end: padding (no memory traffic)
        st.rg = live ? i_rg : -1;
        if (live && i_t == 0) {
            if (i_rg >= i_end) {  // rare: the row group belongs to a later linear of the group
                do {
                    ++i_si;
                    i_begin = i_end;
                    i_end = i_si + 1 < nseg ? L.seg[i_si + 1].block_begin : n_rg;
                } while (i_rg >= i_end);
                const GemvSeg &sg = L.seg[i_si];
                i_N = sg.N;
                i_sstr = sg.scales_stride * 2;
                i_zstr = sg.zeros_stride * 4;
                rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4_t *>(sg.qweight), 0, 0x7FFFFFF0, 0x00020000);
                rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(sg.scales), 0, 0x7FFFFFF0, 0x00020000);
                rs_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(sg.zeros), 0, 0x7FFFFFF0, 0x00020000);
            }
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                int r = (i_rg - i_begin) * ROWS + i;
                r = r < i_N ? r : i_N - 1;  // clamped; the store is masked
                so_w[i] = r * rowbytes;
                so_s[i] = r * i_sstr;
                so_z[i] = r * i_zstr;
            }
        }
        const int c = i_t * 64 + lane;
        const int cc = c < nchunks ? c : nchunks - 1;  // tail lanes re-read the last chunk; their x image is zero
        const int g = cc >> gshift;
        const int oob = live ? 0 : (int)0x7FFFFFF0;
        const int vo_w = cc * 16 + oob, vo_s = g * 2 + oob, vo_z = (g >> 3) * 4 + oob;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            if constexpr (MODE == 4) {  // no weight traffic: the unpack/MFMA work runs on synthetic registers
                const unsigned v = (unsigned)(vo_w + so_w[i]) * 2654435761u;
                st.w[i] = uint4_t{v, v ^ 0x9E3779B9u, v * 3u, v * 5u};
                st.s[i] = (unsigned short)0x2000;
                st.z[i] = 0x88888888u;
                asm volatile("" : "+v"(st.w[i]));
                continue;
            }
            st.w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo_w, so_w[i], /*nt*/ 2);
            st.s[i] = __builtin_amdgcn_raw_buffer_load_b16(rs_s, vo_s, so_s[i], 0);
            // Z8: the caller vouches (TCE_W4_ZERO_POINT_IS_8) that every zero point is 8 -- what the reference quantizer
            // always writes (quantize_methods.py:436-440) -- so the packed zeros are not streamed at all
            if constexpr (Z8) st.z[i] = 0x88888888u;
            else st.z[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_z, vo_z, so_z[i], 0);
        }
        if (live) {
            if (++i_t == T) i_t = 0;
        }
    };

    // ---------------------------------------------------------------------------------------------------------------
    // prologue.  Behind a device-wide barrier: weights first (they do not depend on the predecessor), then the wait,
    // then the activations.  Otherwise activations first: with empty memory queues the x image is ready in ~1 us;
    // queued behind the first weight steps it took 3-5 us.
    // ---------------------------------------------------------------------------------------------------------------
    Step st[DEPTH];
    const bool tagged_in = CHAIN == 2 && L.A_tag != nullptr;                               // grid-uniform: x is polled, not read
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(A), 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(tagged_in ? L.A_tag : reinterpret_cast<const unsigned *>(A)), 0, 0x7FFFFFF0, 0x00020000);
    uint4_t *xs = reinterpret_cast<uint4_t *>(smem);                  // [T][4][64] pieces of 16 bytes (pair-permuted, lane-linear)
    float *xsl = reinterpret_cast<float *>(smem + (size_t)T * 4096);  // [T][64]: sum of the 32 activations of (step, lane)
    const int total_pieces = T * 256;
    // x piece p -> 16 bytes of the vector.  sc0 sc1: device-coherent read -- behind a barrier the vector was written by
    // other CUs (other XCDs, other L2s) moments ago, and this CU / this L2 may still hold the previous token's lines.
    auto piece_halves = [&](int p) -> int {  // first activation of image piece p, -1 = padding
        const int c = (p >> 8) * 64 + (p & 63);
        const int j = (p >> 6) & 3;
        return p < total_pieces && c < nchunks ? c * 32 + j * 8 : -1;
    };
    auto x_piece = [&](int p) -> uint4_t {
        const int k0 = piece_halves(p);
        uint4_t q = uint4_t{0u, 0u, 0u, 0u};
        if (k0 >= 0) q = __builtin_amdgcn_raw_buffer_load_b128(rs_a, k0 * 2, 0, /*sc0|sc1*/ 17);
        return q;
    };
    // Tagged form: NP pieces of this thread at once -- all requests go out, then the tags are checked, and only the pieces that
    // were not there yet are asked for again (one round trip when the producers are done, which is the common case at the
    // slower workgroups; the 32-bit words are single-copy atomic, a piece is complete when its eight tags match).
    auto poll_pieces = [&](auto np_tag, const int *k0, uint4_t *out) {
        constexpr int NP = decltype(np_tag)::value;
        const unsigned tag = bar.tag;
        bool need[NP];
        uint4_t lo[NP], hi[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            need[i] = k0[i] >= 0;
            lo[i] = hi[i] = uint4_t{tag, tag, tag, tag};
        }
        int tries = 0;
        for (;;) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (need[i]) {
                    lo[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, k0[i] * 4, 0, /*sc0|sc1*/ 17);
                    hi[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, k0[i] * 4 + 16, 0, /*sc0|sc1*/ 17);
                }
            bool again = false;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (need[i]) {
                    const unsigned bad = ((lo[i].x ^ tag) | (lo[i].y ^ tag) | (lo[i].z ^ tag) | (lo[i].w ^ tag) | (hi[i].x ^ tag) | (hi[i].y ^ tag) |
                                          (hi[i].z ^ tag) | (hi[i].w ^ tag)) >> 16;
                    need[i] = bad != 0u;
                    again |= need[i];
                }
            if (!again) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++tries & 255) == 0) {  // ~0.3 ms: has anybody given up?  ~0.3 s: give up (the plan's status word says so)
                if (__hip_atomic_load(bar.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (tries > (1 << 18)) {
                    __hip_atomic_store(bar.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i)
            out[i] = k0[i] >= 0 ? uint4_t{(lo[i].x & 0xFFFFu) | (lo[i].y << 16), (lo[i].z & 0xFFFFu) | (lo[i].w << 16), (hi[i].x & 0xFFFFu) | (hi[i].y << 16),
                                          (hi[i].z & 0xFFFFu) | (hi[i].w << 16)}
                                : uint4_t{0u, 0u, 0u, 0u};
    };
    constexpr int XR = 4;  // pieces per thread held in registers across the weight prologue (covers K <= 8192 x waves/16)
    uint4_t xr[XR];
    if (tagged_in) {  // weights first (they depend on nothing), then the wait
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(st[d]);
        __builtin_amdgcn_sched_barrier(0);
        int k0[XR];
#pragma unroll
        for (int i = 0; i < XR; ++i) k0[i] = piece_halves(tid + i * nthreads);
        poll_pieces(std::integral_constant<int, XR>{}, k0, xr);
    } else {
        // x loads first, the weight prologue right behind them, and only then the first use of x: the x data returns
        // first (in-order), and the weights are already in flight while the image is written and summed
#pragma unroll
        for (int i = 0; i < XR; ++i) xr[i] = x_piece(tid + i * nthreads);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(st[d]);
        __builtin_amdgcn_sched_barrier(0);
    }
    stamp(1);
    if (L.gamma) {
        // fused RMSNorm prologue (generalT5LayerNorm, LlamaRMSNorm.cu:68-93): rs by the workgroup in the shape-independent
        // order of rmsnorm_rs_block, then half(clamp((x * rs) * gamma)) goes into the image instead of x.
        // Token kernel: the polled pieces are parked RAW in their own image slots first (a slot is written and later rewritten
        // by the same thread), and both passes read them from there instead of from memory.
        if (tagged_in) {
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int p = tid + i * nthreads;
                if (p < total_pieces) xs[p] = xr[i];
            }
            for (int p = tid + XR * nthreads; p < total_pieces; p += nthreads) {  // long K only
                const int k1[1] = {piece_halves(p)};
                uint4_t one[1];
                poll_pieces(std::integral_constant<int, 1>{}, k1, one);
                xs[p] = one[0];
            }
            __syncthreads();
        }
        auto raw_piece = [&](int lin) -> half8_t {  // 8 activations k = 8 lin ..: image slot of (chunk lin / 4, quarter lin % 4)
            if (tagged_in) {
                const int c = lin >> 2, j = lin & 3;
                return __builtin_bit_cast(half8_t, xs[(c >> 6) * 256 + j * 64 + (c & 63)]);
            }
            return __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_a, lin * 16, 0, /*sc0|sc1*/ 17));
        };
        const float rs = rmsnorm_rs_block(raw_piece, K, L.eps, wave, NW, lane, reinterpret_cast<float *>(smem + (size_t)T * (4096 + 256) + 256));
        for (int p = tid; p < total_pieces; p += nthreads) {
            const int c = (p >> 8) * 64 + (p & 63), j = (p >> 6) & 3;
            uint4_t o = uint4_t{0u, 0u, 0u, 0u};
            if (c < nchunks) {
                const int k0 = c * 32 + j * 8;
                const half8_t v = __builtin_bit_cast(half8_t, tagged_in ? xs[p] : x_piece(p));
                const float4_t g0 = *reinterpret_cast<const float4_t *>(L.gamma + k0), g1 = *reinterpret_cast<const float4_t *>(L.gamma + k0 + 4);
                half8_t y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = rmsnorm_out(v[e], rs, g0[e]);
                    y[4 + e] = rmsnorm_out(v[4 + e], rs, g1[e]);
                }
                o = pair_permute(__builtin_bit_cast(uint4_t, y));
            }
            xs[p] = o;
        }
    } else {
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int p = tid + i * nthreads;
            if (p < total_pieces) xs[p] = pair_permute(xr[i]);
        }
        for (int p = tid + XR * nthreads; p < total_pieces; p += nthreads) {  // long K only
            if (tagged_in) {
                const int k1[1] = {piece_halves(p)};
                uint4_t one[1];
                poll_pieces(std::integral_constant<int, 1>{}, k1, one);
                xs[p] = pair_permute(one[0]);
            } else {
                xs[p] = pair_permute(x_piece(p));
            }
        }
    }
    __syncthreads();
    {
        // D[i][j] = sum_k A_i[k] B_j[k] with A = ones: every accumulator register of a lane holds that lane's own sum
        const half4_t ones = half4_t{(half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};
        for (int t = wave; t < T; t += NW) {
            float4_t xs4 = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4_t xp = xs[(t * 4 + j) * 64 + lane];
                xs4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, __builtin_bit_cast(half4_t, uint2_t{xp.x, xp.y}), xs4, 0, 0, 0);
                xs4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, __builtin_bit_cast(half4_t, uint2_t{xp.z, xp.w}), xs4, 0, 0, 0);
            }
            xsl[t * 64 + lane] = xs4[0];
        }
    }
    __syncthreads();
    if constexpr (MODE == 2) *ts_x_ready = wall_clock64();
    stamp(2);

    // ---------------------------------------------------------------------------------------------------------------
    // compute side
    // ---------------------------------------------------------------------------------------------------------------
    float acc[ROWS][4];  // the 4 accumulator registers of the 4x4x4 MFMA; the lane's own dot product is [lane & 3]
    float corr[ROWS];    // sum over chunks of s * (1024 + 16 z) * sum_k x_k
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        corr[i] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    }
    unsigned mask_hi;  // nibble mask in a VGPR (tce_common.hpp: one scalar operand per VOP3 on gfx9)
    asm volatile("v_mov_b32 %0, 0x00F000F0" : "=v"(mask_hi));
    const unsigned magic = 0x64006400u;  // (1024.0h, 1024.0h)
    float diag[4];                       // one-hot pick of the lane's diagonal accumulator and the final /16 in one factor
#pragma unroll
    for (int q = 0; q < 4; ++q) diag[q] = (lane & 3) == q ? 0.0625f : 0.0f;

    int c_t = 0;  // cursor: step within the row group being computed
    int c_si = 0, c_begin = 0;
    int c_end = nseg > 1 ? L.seg[1].block_begin : n_rg;
    half_t *c_C = L.seg[0].C;
    unsigned *c_T = CHAIN == 2 ? L.C_tag[0] : nullptr;  // the same outputs as (tag << 16 | bits) words for the launches behind this one
    int c_N = L.seg[0].N;
    int c_epi = L.seg[0].epilogue;

    auto compute = [&](const Step &st) {
        const int t = c_t;
        if constexpr (MODE == 1) {  // consume the loads, skip the math
#pragma unroll
            for (int i = 0; i < ROWS; ++i) acc[i][0] += (float)((st.w[i].x ^ st.w[i].y ^ st.w[i].z ^ st.w[i].w ^ st.s[i] ^ st.z[i]) & 0xFFu);
        } else {
            half4_t xb[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4_t xp = xs[(t * 4 + j) * 64 + lane];
                xb[2 * j] = __builtin_bit_cast(half4_t, uint2_t{xp.x, xp.y});      // (x0,x4,x1,x5) of word j
                xb[2 * j + 1] = __builtin_bit_cast(half4_t, uint2_t{xp.z, xp.w});  // (x2,x6,x3,x7)
            }
            const float xsum = xsl[t * 64 + lane];
            int zsh = 0;
            if constexpr (!Z8) {
                const int c = t * 64 + lane;
                const int cc = c < nchunks ? c : nchunks - 1;
                zsh = ((cc >> gshift) & 7) * 4;
            }
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                // keep each row's unpack -> MFMA -> scale sequence together: without the fence the scheduler hoists all
                // rows' unpacks ahead of the first MFMA and the register allocation explodes
                __builtin_amdgcn_sched_barrier(0);
                float4_t blk = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned w = st.w[i][j];
                    const unsigned t0 = ((w << 4) & mask_hi) | magic;  // (1024+16 q0, 1024+16 q4)
                    const unsigned t1 = (w & mask_hi) | magic;         // (q1, q5)
                    const unsigned t2 = ((w >> 4) & mask_hi) | magic;  // (q2, q6)
                    const unsigned t3 = ((w >> 8) & mask_hi) | magic;  // (q3, q7)
                    blk = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(half4_t, uint2_t{t0, t1}), xb[2 * j], blk, 0, 0, 0);
                    blk = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(half4_t, uint2_t{t2, t3}), xb[2 * j + 1], blk, 0, 0, 0);
                }
                // lanes 4b..4b+3 share a quantization group (32 weights per lane, groups of >= 32), so scaling all four
                // accumulator registers by this lane's scale is consistent; the diagonal is picked at the row group's end
                const float s = (float)__builtin_bit_cast(half_t, st.s[i]);
                const float cz = Z8 ? 1152.0f : __builtin_fmaf((float)((st.z[i] >> zsh) & 0xFu), 16.0f, 1024.0f);  // 1024 + 16 z
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][r] = __builtin_fmaf(s, blk[r], acc[i][r]);
                corr[i] = __builtin_fmaf(s * cz, xsum, corr[i]);
            }
        }
        if (t != T - 1) {
            c_t = t + 1;
            return;
        }
        // ---- end of a row group: reduce over the 64 lanes, store, reset (wave-uniform branch) ----
        const int c_rg = st.rg;  // a wave's row groups only move forward, so the per-linear state is sticky here too
        if (c_rg >= c_end) {
            do {
                ++c_si;
                c_begin = c_end;
                c_end = c_si + 1 < nseg ? L.seg[c_si + 1].block_begin : n_rg;
            } while (c_rg >= c_end);
            c_C = L.seg[c_si].C;
            if constexpr (CHAIN == 2) c_T = L.C_tag[c_si];
            c_N = L.seg[c_si].N;
            c_epi = L.seg[c_si].epilogue;
        }
        const int row0 = (c_rg - c_begin) * ROWS;
        half_t outv[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            float v = corr[i] * -0.0625f;
            v = __builtin_fmaf(acc[i][0], diag[0], v);
            v = __builtin_fmaf(acc[i][1], diag[1], v);
            v = __builtin_fmaf(acc[i][2], diag[2], v);
            v = __builtin_fmaf(acc[i][3], diag[3], v);
            if constexpr (MODE == 1) v = acc[i][0];
            v = wave_sum_dpp_lane63(v);  // total in lane 63
            outv[i] = (half_t)v;
            corr[i] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        }
        // device-scope (write-through) stores: visible to the next launch's coherent reads without a cache flush
        auto put = [&](int idx, half_t h) {
            __hip_atomic_store(reinterpret_cast<unsigned short *>(c_C + idx), __builtin_bit_cast(unsigned short, h), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (CHAIN == 2) {
                if (c_T)  // the hand-off itself: value and "it is there" in ONE word; nothing waits for the store to be acknowledged
                    __hip_atomic_store(c_T + idx, bar.tag | (unsigned)__builtin_bit_cast(unsigned short, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        if (lane == 63) {
            if (c_epi & TCE_W4_SILU_MUL_PAIRS) {  // rows (2n, 2n+1) = (gate n, up n); the host picked an even ROWS
                if constexpr (ROWS % 2 == 0) {
#pragma unroll
                    for (int i = 0; i < ROWS; i += 2)
                        if (row0 + i + 1 < c_N) put((row0 + i) >> 1, silu_mul_half(outv[i], outv[i + 1]));
                }
            } else if (c_epi & TCE_W4_ADD_TO_C) {
#pragma unroll
                for (int i = 0; i < ROWS; ++i)
                    if (row0 + i < c_N) {
                        // device-scope read: inside a token kernel the residual may have been written by another CU
                        const unsigned short old = __hip_atomic_load(reinterpret_cast<unsigned short *>(c_C + row0 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        put(row0 + i, __builtin_bit_cast(half_t, old) + outv[i]);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < ROWS; ++i)
                    if (row0 + i < c_N) put(row0 + i, outv[i]);
            }
        }
        c_t = 0;
    };

    // ---- the ring: consume slot d, refill it; once the work is used up the refills are padding, and the loop ends
    // after the first round that found nothing to consume ----
    bool any;
    do {
        any = false;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (st[d].rg >= 0) {  // wave-uniform
                compute(st[d]);
                any = true;
            }
            issue(st[d]);
        }
    } while (any);
    stamp(3);

    if constexpr (CHAIN == 2) __syncthreads();  // everyone is done with the x image and the ticket counter; the stores are on their way, nobody waits for them
    stamp(4);
}

template <int ROWS, int DEPTH, bool Z8, int MODE = 0>
__global__ __launch_bounds__(1024) void w4a16_gemv_stream_kernel(const StreamArgs args) {
    unsigned long long ts0 = 0, ts1 = 0, cy0 = 0;
    if constexpr (MODE == 2) {
        ts0 = wall_clock64();
        cy0 = clock64();
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    gemv_launch_body<ROWS, DEPTH, Z8, MODE>(args.launch, Barrier{nullptr, 0u, nullptr}, smem, &ts1);
    if constexpr (MODE == 2) {
        if ((threadIdx.x & 63) == 63 && args.dbg) {
            unsigned long long *d = args.dbg + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
            d[0] = ts0; d[1] = ts1; d[2] = wall_clock64(); d[3] = clock64() - cy0;  // shader cycles, for the clock rate
        }
    }
}

// A whole list of launches (a decode token's linears) in one kernel; see the header.
// Workgroup size: 16 waves where the body fits 128 registers; the forms that did not (every general-zero-point form and four rows per wave: 3-34 registers
// spilled to scratch under __launch_bounds__(1024), round-3 verdict) run 8-wave workgroups, two per CU, with the 256 registers that leaves them.
constexpr int token_max_waves(int rows, int depth, bool z8) { return (!z8 || rows == 4) ? 8 : 16; }
template <int ROWS, int DEPTH, bool Z8, int CHAIN>
__global__ __launch_bounds__(64 * token_max_waves(ROWS, DEPTH, Z8)) void w4a16_gemv_token_kernel(const TokenArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = args.n_launches;
    unsigned tag = 0;
    if constexpr (CHAIN == 2) tag = __builtin_amdgcn_readfirstlane(__hip_atomic_load(args.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) << 16;
    for (int j = 0; j < n; ++j) {
        Barrier bar;
        bar.status = args.status;
        bar.tag = tag;
        bar.stamps = args.dbg ? args.dbg + ((size_t)blockIdx.x * n + j) * 8 : nullptr;
        gemv_launch_body<ROWS, DEPTH, Z8, 0, CHAIN>(args.launches[j], bar, smem, nullptr);
    }
}

// the token's tag for the NEXT replay: 1, 2, ..., 65535, 1, ... (0 is what freshly allocated shadow vectors hold)
__global__ void token_epoch_kernel(unsigned *epoch) { *epoch = *epoch % 65535u + 1u; }

int g_num_cus = 0;
thread_local int g_stream_rows = 0, g_stream_nw = 0, g_stream_depth = 0, g_stream_bpc = 0;  // forced geometry (0 = automatic)
thread_local int g_stream_mode = 0;
unsigned long long *g_stream_dbg = nullptr;

int num_cus() {
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_num_cus;
}

size_t lds_bytes(int K) { return (size_t)(((K >> 5) + 63) / 64) * (4096 + 256) + 256 + 4096; }  // x image, chunk sums, ticket words, RMSNorm partials

template <typename KFn>
hipError_t set_lds(KFn kfn, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <int ROWS, int DEPTH, bool Z8, int MODE = 0>
hipError_t launch_stream(const StreamArgs &a, int blocks, int nw, hipStream_t stream) {
    const size_t lds = lds_bytes(a.launch.K);
    auto kfn = w4a16_gemv_stream_kernel<ROWS, DEPTH, Z8, MODE>;
    hipError_t e = set_lds(kfn, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(blocks, 1, 1), dim3(64 * nw, 1, 1), lds, stream, a);
    return hipGetLastError();
}

// Fills one launch record; returns false if a linear does not fit the 2 GiB buffer-descriptor window.
bool fill_launch(const tce_w4a16_desc *descs, int count, int rows, StreamLaunch *out, bool *z8) {
    const tce_w4a16_desc &d0 = descs[0];
    StreamLaunch &a = *out;
    a = StreamLaunch{};
    a.A = static_cast<const half_t *>(d0.A);
    a.K = d0.K;
    a.log2g = d0.group_size == 128 ? 7 : (d0.group_size == 64 ? 6 : 5);
    a.nseg = count;
    a.gamma = static_cast<const float *>(d0.rmsnorm_gamma);
    a.eps = d0.rmsnorm_eps;
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
        s.epilogue = d.flags & (TCE_W4_SILU_MUL_PAIRS | TCE_W4_ADD_TO_C);
        s.ldc = d.ldc ? d.ldc : d.N;
        s.scales_stride = d.scales_stride ? d.scales_stride : zw * 8;
        s.zeros_stride = d.zeros_stride ? d.zeros_stride : zw;
        if ((long long)d.N * (d.K / 2) >= 0x7FFFFFF0LL || (long long)d.N * s.scales_stride * 2 >= 0x7FFFFFF0LL ||
            (long long)d.N * s.zeros_stride * 4 >= 0x7FFFFFF0LL)
            return false;
        s.block_begin = n_rg;
        n_rg += (d.N + rows - 1) / rows;
        if (!(d.flags & TCE_W4_ZERO_POINT_IS_8)) *z8 = false;
    }
    for (int i = count; i < TCE_MAX_GROUP; ++i) a.seg[i] = a.seg[0];
    a.n_rg = n_rg;
    return true;
}

// rows per row group: 2 (one x read and one set of addresses per two rows) once every wave still gets several row
// groups to pull, else 1 (measured: profiles/r1/stream_sweep.jsonl)
int pick_rows(const tce_w4a16_desc *descs, const int32_t *groups, int n_launches, long waves) {
    double rows = 0.0, bytes = 0.0;
    for (int l = 0, off = 0; l < n_launches; off += groups[l], ++l) {
        long n = 0;
        for (int i = 0; i < groups[l]; ++i) n += descs[off + i].N;
        rows += (double)n * n * descs[off].K;  // byte-weighted mean of the launches' row counts
        bytes += (double)n * descs[off].K;
    }
    return rows / bytes >= 4.0 * (double)waves ? 2 : 1;
}

}  // namespace

void set_gemv_stream_debug(int mode, void *buf) {
    g_stream_mode = mode;
    g_stream_dbg = static_cast<unsigned long long *>(buf);
}

void set_gemv_stream_config(int rows, int nw, int depth) {
    g_stream_bpc = rows >= 10 ? rows / 10 : 0;  // tuning: rows = 10 * workgroups-per-CU + rows-per-row-group
    g_stream_rows = rows % 10;
    g_stream_nw = nw;
    g_stream_depth = depth;
}

bool gemv_stream_supports(const tce_w4a16_desc *descs, int count) {
    if (descs[0].M != 1) return false;
    if (lds_bytes(descs[0].K) > 84 * 1024) return false;
    StreamLaunch tmp;
    bool z8 = true;
    return fill_launch(descs, count, 1, &tmp, &z8);
}

// Returns TCE_ERR_UNSUPPORTED_SHAPE when the persistent form does not apply (the caller then uses the workgroup-per-
// row-block kernel of w4a16_gemv.hip).
int launch_w4a16_gemv_stream(const tce_w4a16_desc *descs, int count, hipStream_t stream, hipError_t *hip_err, const float *gamma, float eps) {
    if (!gemv_stream_supports(descs, count)) return TCE_ERR_UNSUPPORTED_SHAPE;
    const int cus = num_cus();
    if (cus == 0) return TCE_ERR_HIP;

    // ---- geometry: one workgroup of 16 waves per CU, two rows per row group for large launches ----
    int rows = g_stream_rows, nw = g_stream_nw ? g_stream_nw : 16, depth = g_stream_depth;
    int bpc = g_stream_bpc ? g_stream_bpc : 1;
    if (bpc * nw > 32) bpc = 32 / nw > 0 ? 32 / nw : 1;  // 8 waves per SIMD at most
    const int32_t one_group = count;
    if (rows == 0) rows = pick_rows(descs, &one_group, 1, (long)cus * bpc * nw);
    for (int i = 0; i < count; ++i)
        if ((descs[i].flags & TCE_W4_SILU_MUL_PAIRS) && rows == 1) rows = 2;  // a (gate, up) row pair sits in one row group
    if (rows != 1 && rows != 2 && rows != 4) return TCE_ERR_BAD_ARG;
    if (depth == 0) depth = rows == 1 ? 3 : 2;  // deeper rings measured slower: the memory system is oversubscribed as it is
    if (rows == 4 && depth > 2) depth = 2;
    if (nw < 1 || nw > 16 || depth < 2 || depth > 3) return TCE_ERR_BAD_ARG;

    StreamArgs a{};
    bool z8 = true;
    if (!fill_launch(descs, count, rows, &a.launch, &z8)) return TCE_ERR_UNSUPPORTED_SHAPE;
    a.dbg = g_stream_dbg;
    if (gamma) {
        a.launch.gamma = gamma;
        a.launch.eps = eps;
    }
    int blocks = cus * bpc;
    if ((long)blocks * nw > a.launch.n_rg) blocks = (a.launch.n_rg + nw - 1) / nw;  // fewer row groups than waves

    hipError_t e = hipSuccess;
    bool found = true;
#ifdef TCE_LAB  // (diagnostic instantiations -- stream only / timestamps / arithmetic only --: the lab build, build.py --lab)
    if (g_stream_mode != 0) {
        const int md = g_stream_mode;
        if (rows == 2 && depth == 2) e = md == 1 ? launch_stream<2, 2, false, 1>(a, blocks, nw, stream) : (md == 4 ? launch_stream<2, 2, false, 4>(a, blocks, nw, stream) : launch_stream<2, 2, false, 2>(a, blocks, nw, stream));
        else if (rows == 4 && depth == 2) e = md == 1 ? launch_stream<4, 2, false, 1>(a, blocks, nw, stream) : (md == 4 ? launch_stream<4, 2, false, 4>(a, blocks, nw, stream) : launch_stream<4, 2, false, 2>(a, blocks, nw, stream));
        else return TCE_ERR_BAD_ARG;
    } else
#endif
    {
#define TCE_S(R_, D_) \
    if (rows == R_ && depth == D_) e = z8 ? launch_stream<R_, D_, true>(a, blocks, nw, stream) : launch_stream<R_, D_, false>(a, blocks, nw, stream); else
        TCE_S(1, 2) TCE_S(1, 3) TCE_S(2, 2) TCE_S(2, 3) TCE_S(4, 2)
        found = false;
#undef TCE_S
    }
    if (!found) return TCE_ERR_BAD_ARG;
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Token plans: the launch list lives in device memory; one kernel walks it.
// ---------------------------------------------------------------------------------------------------------------------
struct TokenPlan {
    StreamLaunch *launches = nullptr;  // device
    unsigned *sync = nullptr;          // device: [0] status, [1] the next token's tag
    unsigned *shadow = nullptr;        // device: every launch's outputs as (tag << 16 | fp16 bits) words
    int n = 0, rows = 0, depth = 0, nw = 0, blocks = 0;
    bool z8 = false;
    size_t lds = 0;
    // mode 1 (overlapped launches, w4a16_gemv_ovl.hip): one kernel per launch, issued on `chains` alternating branches
    int mode = 0, chains = 0;
    std::vector<StreamLaunch> host;
    std::vector<OvlGeom> geom;
    std::vector<hipStream_t> side;   // chains - 1 side streams
    std::vector<hipEvent_t> events;  // [0] fork, [1 ..] joins
};

void token_plan_destroy(TokenPlan *tp) {
    if (!tp) return;
    if (tp->launches) (void)hipFree(tp->launches);
    if (tp->sync) (void)hipFree(tp->sync);
    if (tp->shadow) (void)hipFree(tp->shadow);
    for (hipStream_t s : tp->side) (void)hipStreamDestroy(s);
    for (hipEvent_t e : tp->events) (void)hipEventDestroy(e);
    delete tp;
}

int token_plan_mode(const TokenPlan *tp) { return tp ? tp->mode : 0; }


namespace {
template <int ROWS, int DEPTH, bool Z8>
hipError_t token_setup(const TokenPlan &tp, int *max_blocks_per_cu) {
    const void *kfn = reinterpret_cast<const void *>(w4a16_gemv_token_kernel<ROWS, DEPTH, Z8, 2>);
    if (tp.lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tp.lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(max_blocks_per_cu, kfn, 64 * tp.nw, tp.lds);
}
template <int ROWS, int DEPTH, bool Z8>
hipError_t token_launch(const TokenPlan &tp, hipStream_t stream) {
    TokenArgs a;
    a.dbg = g_stream_mode == 2 ? g_stream_dbg : nullptr;  // captured when the plan is built
    a.launches = tp.launches;
    a.n_launches = tp.n;
    a.status = tp.sync;
    a.epoch = tp.sync + 1;
    hipLaunchKernelGGL((w4a16_gemv_token_kernel<ROWS, DEPTH, Z8, 2>), dim3(tp.blocks, 1, 1), dim3(64 * tp.nw, 1, 1), tp.lds, stream, a);
    hipLaunchKernelGGL(token_epoch_kernel, dim3(1), dim3(1), 0, stream, tp.sync + 1);
    return hipGetLastError();
}
#define TCE_TOKEN_DISPATCH(FN, ...)                                                                 \
    (tp.rows == 1 ? (tp.z8 ? FN<1, 3, true>(__VA_ARGS__) : FN<1, 3, false>(__VA_ARGS__))            \
     : tp.rows == 2 ? (tp.depth == 3 ? (tp.z8 ? FN<2, 3, true>(__VA_ARGS__) : FN<2, 3, false>(__VA_ARGS__)) \
                                     : (tp.z8 ? FN<2, 2, true>(__VA_ARGS__) : FN<2, 2, false>(__VA_ARGS__)))  \
                    : (tp.z8 ? FN<4, 2, true>(__VA_ARGS__) : FN<4, 2, false>(__VA_ARGS__)))
}  // namespace

// Builds the device-side launch list.  Returns TCE_ERR_UNSUPPORTED_SHAPE if a launch is not an M = 1 GEMV this kernel takes.
int token_plan_create(const tce_w4a16_desc *descs, const int32_t *groups, int n_launches, TokenPlan **out, hipError_t *hip_err, int mode) {
    const int cus = num_cus();
    if (cus == 0) return TCE_ERR_HIP;
    int maxK = 0;
    for (int l = 0, off = 0; l < n_launches; off += groups[l], ++l) {
        if (!gemv_stream_supports(descs + off, groups[l])) return TCE_ERR_UNSUPPORTED_SHAPE;
        if (descs[off].K > maxK) maxK = descs[off].K;
    }
    TokenPlan *tpp = new (std::nothrow) TokenPlan();
    if (!tpp) return TCE_ERR_BAD_ARG;
    TokenPlan &tp = *tpp;
    tp.n = n_launches;
    tp.mode = mode;
    if (mode == 1) {  // per-launch geometry: waves per workgroup from K, the grid from the occupancy (ovl_geometry)
        tp.chains = ovl_chains();
        tp.geom.resize(n_launches);
        for (int l = 0, off = 0; l < n_launches; off += groups[l], ++l) {
            const int rc = ovl_geometry(descs + off, groups[l], cus, tp.chains, &tp.geom[l], hip_err);
            if (rc != TCE_OK) {
                token_plan_destroy(tpp);
                return rc;
            }
        }
    }
    tp.nw = g_stream_nw ? g_stream_nw : 16;
    int bpc = g_stream_bpc ? g_stream_bpc : 1;
    if (bpc * tp.nw > 16) bpc = 16 / tp.nw > 0 ? 16 / tp.nw : 1;
    tp.rows = g_stream_rows ? g_stream_rows : pick_rows(descs, groups, n_launches, (long)cus * bpc * tp.nw);
    {
        int total = 0;
        for (int l = 0; l < n_launches; ++l) total += groups[l];
        for (int i = 0; i < total; ++i)
            if ((descs[i].flags & TCE_W4_SILU_MUL_PAIRS) && tp.rows == 1) tp.rows = 2;
    }
    tp.depth = g_stream_depth ? g_stream_depth : (tp.rows == 4 ? 2 : 3);
    if (tp.rows == 4) tp.depth = 2;
    if (tp.rows == 1) tp.depth = 3;
    tp.lds = lds_bytes(maxK);
    tp.z8 = true;
    std::vector<StreamLaunch> &host = tp.host;
    host.resize(n_launches);
    for (int l = 0, off = 0; l < n_launches; off += groups[l], ++l) {
        bool z8 = true;
        if (!fill_launch(descs + off, groups[l], mode == 1 ? tp.geom[l].rows_per_block : tp.rows, &host[l], mode == 1 ? &z8 : &tp.z8)) {
            token_plan_destroy(tpp);
            return TCE_ERR_UNSUPPORTED_SHAPE;
        }
    }
    hipError_t e = hipSuccess;
    if (mode == 0) {
        if (tp.nw > token_max_waves(tp.rows, tp.depth, tp.z8)) {  // (the kernel form is known only now: the zero points were just examined)
            tp.nw = token_max_waves(tp.rows, tp.depth, tp.z8);
            if (!g_stream_bpc) bpc = 16 / tp.nw;
        }
        // every workgroup must be resident at once: a launch spins until its producers have delivered
        int per_cu = 0;
        e = TCE_TOKEN_DISPATCH(token_setup, tp, &per_cu);
        if (e == hipSuccess && per_cu < 1) {
            token_plan_destroy(tpp);
            return TCE_ERR_UNSUPPORTED_SHAPE;
        }
        if (per_cu < bpc) bpc = per_cu;
        tp.blocks = cus * bpc;
    } else {
        tp.blocks = 0;
        for (const OvlGeom &g : tp.geom)
            if (g.grid > tp.blocks) tp.blocks = g.grid;
        if (tp.blocks > 4096) {  // (the diagnostics buffer is laid out for 4096 workgroups per launch)
            token_plan_destroy(tpp);
            return TCE_ERR_UNSUPPORTED_SHAPE;
        }
        tp.rows = 2;
        tp.depth = 2;
        tp.nw = tp.geom[0].wn;
    }
    if (e == hipSuccess) {
        // Who produces what: launch l's activation vector is looked up among the outputs of the launches in front of it, latest
        // first (buffers are reused from layer to layer; the latest writer is the one stream order would have made visible).
        // Found: the launch polls that producer's shadow words.  Not found: the vector comes from outside the plan and is simply read.
        size_t words = 0;
        std::vector<size_t> base(n_launches * TCE_MAX_GROUP, 0);
        for (int l = 0; l < n_launches; ++l)
            for (int i = 0; i < host[l].nseg; ++i) {
                base[l * TCE_MAX_GROUP + i] = words;
                const int n_out = (host[l].seg[i].epilogue & TCE_W4_SILU_MUL_PAIRS) ? host[l].seg[i].N / 2 : host[l].seg[i].N;
                words += ((size_t)n_out + 63) & ~(size_t)63;
            }
        e = hipMalloc(reinterpret_cast<void **>(&tp.shadow), words * sizeof(unsigned));
        if (e == hipSuccess) e = hipMemset(tp.shadow, 0, words * sizeof(unsigned));
        std::vector<int> producer(n_launches, -1);
        for (int l = 0; l < n_launches && e == hipSuccess; ++l) {
            StreamLaunch &L = host[l];
            for (int i = 0; i < TCE_MAX_GROUP; ++i) L.C_tag[i] = i < L.nseg ? tp.shadow + base[l * TCE_MAX_GROUP + i] : nullptr;
            L.A_tag = nullptr;
            const char *a0 = reinterpret_cast<const char *>(L.A), *a1 = a0 + (size_t)L.K * 2;
            for (int q = l - 1; q >= 0 && !L.A_tag; --q)
                for (int i = 0; i < host[q].nseg; ++i) {
                    const int n_out = (host[q].seg[i].epilogue & TCE_W4_SILU_MUL_PAIRS) ? host[q].seg[i].N / 2 : host[q].seg[i].N;
                    const char *c0 = reinterpret_cast<const char *>(host[q].seg[i].C), *c1 = c0 + (size_t)n_out * 2;
                    if (a0 >= c0 && a1 <= c1) {  // the whole vector lies inside this output (a prefix / slice of it is fine)
                        if ((a0 - c0) % 16 != 0) {  // pieces of 8 values must stay 32-byte aligned in the shadow
                            token_plan_destroy(tpp);
                            return TCE_ERR_UNSUPPORTED_SHAPE;
                        }
                        L.A_tag = host[q].C_tag[i] + (a0 - c0) / 2;
                        producer[l] = q;
                        break;
                    }
                    if (a0 < c1 && a1 > c0) {  // straddles an output: produced in pieces, not something a poll of one shadow can wait for
                        token_plan_destroy(tpp);
                        return TCE_ERR_UNSUPPORTED_SHAPE;
                    }
                }
        }
        // What the polls do NOT order: anti- and output dependences on the plain buffers.  A launch that writes memory which an
        // earlier launch reads un-tagged (an activation vector from outside the plan; the old value of a residual add) or also
        // writes must come after that launch THROUGH THE DATA FLOW -- a launch starts to compute when all of its producer's rows
        // are there, every one of which was computed after that producer's own input was complete, and so on up the chain.
        // A launch list whose stream order protects such a hazard by position only is not taken (built stream-ordered instead).
        const size_t nw64 = ((size_t)n_launches + 63) / 64;
        std::vector<uint64_t> anc((size_t)n_launches * nw64, 0);  // anc[l]: bit q = launch q is an ancestor of l in the data flow
        for (int l = 0; l < n_launches; ++l)
            if (producer[l] >= 0) {
                const int q = producer[l];
                for (size_t w = 0; w < nw64; ++w) anc[l * nw64 + w] = anc[q * nw64 + w];
                anc[l * nw64 + (size_t)q / 64] |= 1ull << (q % 64);
            }
        auto n_out_of = [&](const StreamLaunch &L, int i) { return (L.seg[i].epilogue & TCE_W4_SILU_MUL_PAIRS) ? L.seg[i].N / 2 : L.seg[i].N; };
        for (int l = 1; l < n_launches && e == hipSuccess; ++l)
            for (int i = 0; i < host[l].nseg; ++i) {
                const char *w0 = reinterpret_cast<const char *>(host[l].seg[i].C), *w1 = w0 + (size_t)n_out_of(host[l], i) * 2;
                for (int q = 0; q < l; ++q) {
                    bool touches = false;
                    if (producer[q] < 0) {  // q read its activations un-tagged
                        const char *r0 = reinterpret_cast<const char *>(host[q].A), *r1 = r0 + (size_t)host[q].K * 2;
                        touches = r0 < w1 && r1 > w0;
                    }
                    for (int k = 0; k < host[q].nseg && !touches; ++k) {
                        const char *c0 = reinterpret_cast<const char *>(host[q].seg[k].C), *c1 = c0 + (size_t)n_out_of(host[q], k) * 2;
                        touches = c0 < w1 && c1 > w0;
                    }
                    if (touches && !((anc[l * nw64 + (size_t)q / 64] >> (q % 64)) & 1ull)) {
                        token_plan_destroy(tpp);
                        return TCE_ERR_UNSUPPORTED_SHAPE;
                    }
                }
            }
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&tp.launches), sizeof(StreamLaunch) * n_launches);
    if (e == hipSuccess) e = hipMemcpy(tp.launches, host.data(), sizeof(StreamLaunch) * n_launches, hipMemcpyHostToDevice);
    if (e == hipSuccess && mode == 1) {
        for (int i = 1; i < tp.chains && e == hipSuccess; ++i) {
            hipStream_t s = nullptr;
            e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            if (e == hipSuccess) tp.side.push_back(s);
        }
        for (int i = 0; i < tp.chains && e == hipSuccess; ++i) {
            hipEvent_t ev = nullptr;
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) tp.events.push_back(ev);
        }
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&tp.sync), sizeof(unsigned) * 2);
    if (e == hipSuccess) {
        const unsigned init[2] = {0u, 1u};  // status clear; the first token's tag
        e = hipMemcpy(tp.sync, init, sizeof(init), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        token_plan_destroy(tpp);
        return TCE_ERR_HIP;
    }
    *out = tpp;
    return TCE_OK;
}

// Enqueues the token: one kernel runs the whole list, a one-thread kernel behind it advances the tag.
int token_plan_enqueue(TokenPlan *tpp, hipStream_t stream, hipError_t *hip_err) {
    const TokenPlan &tp = *tpp;
    hipError_t e = hipSuccess;
    if (tp.mode == 1) {
        // fork: the side branches start where `stream` is now; launch l goes to branch l % chains; join; then the tag advances.
        // (Inside a stream capture this records a graph of `chains` parallel kernel chains with no edge between consecutive launches.)
        e = hipEventRecord(tp.events[0], stream);
        for (size_t i = 0; i < tp.side.size() && e == hipSuccess; ++i) e = hipStreamWaitEvent(tp.side[i], tp.events[0], 0);
        for (int l = 0; l < tp.n && e == hipSuccess; ++l) {
            const int b = l % tp.chains;
            e = ovl_enqueue(tp.launches + l, tp.geom[l], tp.sync + 1, tp.sync, l, b == 0 ? stream : tp.side[b - 1]);
        }
        for (size_t i = 0; i < tp.side.size() && e == hipSuccess; ++i) {
            e = hipEventRecord(tp.events[1 + i], tp.side[i]);
            if (e == hipSuccess) e = hipStreamWaitEvent(stream, tp.events[1 + i], 0);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(token_epoch_kernel, dim3(1), dim3(1), 0, stream, tp.sync + 1);
            e = hipGetLastError();
        }
    } else {
        e = TCE_TOKEN_DISPATCH(token_launch, tp, stream);
    }
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

int token_plan_status(TokenPlan *tp, unsigned *status, hipError_t *hip_err) {
    const hipError_t e = hipMemcpy(status, tp->sync, sizeof(unsigned), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

void token_plan_geometry(const TokenPlan *tp, int *rows, int *depth, int *waves, int *blocks) {
    *rows = tp->rows;
    *depth = tp->depth;
    *waves = tp->nw;
    *blocks = tp->blocks;
}

}  // namespace tce
