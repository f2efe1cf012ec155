// w4a16_gemv_ovl.hip -- W4A16 decode GEMV launches that OVERLAP with the launch that produces their activations
// (TCE_PLAN_OVERLAPPED plans), gfx950 / MI355X, M = 1.
//
// Same math and data layout as w4a16_gemv.hip (reference kernels/cuda/gemv_cuda.cu:140-194 behind
// MatmulOperator::gemv_forward_cuda; that file's header explains the unpack / MFMA-diagonal scheme).  What differs is how a
// launch is ordered against the launch in front of it (call order: llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu:73-115)
// and where the weights wait for their turn.
//
// A stream-ordered decode token pays, per dependent launch, a kernel boundary (~1.7 us during which HBM idles) and then the
// first-byte latency of the new launch's weight stream (~1.5 us) -- DESIGN.md 3.1: 3.4 us of a 7.6 us average launch.  Here the
// token's launches are issued on S alternating graph branches with NO edge between consecutive launches; what orders launch
// j+1 behind launch j is the DATA: every output is also written as one 32-bit word (token tag << 16 | fp16 bits) with a
// write-through store into a shadow vector the plan owns (the scheme of the token kernel, w4a16_gemv_stream.hip), and a
// consumer workgroup
//     1. requests its activation words and looks at their tags,
//     2. starts its weight stream: every wave owns a RING of D step slots in LDS that LDS-DMA fills
//        (buffer_load_dwordx4 ... lds, non-temporal: 1 KiB per instruction and row, plus the row's group scales) -- the ring costs
//        no registers, so the kernel keeps 8 waves per SIMD's worth of VGPR budget and two launches fit on a CU side by side,
//     3. while tags are missing polls ONE address per wave (the last words of the vector) with s_sleep in between -- a
//        thousand waiting workgroups sweeping the whole vector would cost the producer its bandwidth -- and reads the missing
//        pieces again once that sentinel has arrived,
//     4. stages x in LDS ONCE per workgroup and walks its row blocks; the ring runs across row-block boundaries
//        (issue cursor D steps ahead of the compute cursor), padding steps past the end are answered by the buffer unit
//        without memory traffic, so the counted s_waitcnt vmcnt in front of every step is a constant.
// So when launch j's last output lands, launch j+1's workgroups are already resident with their first D steps of weights in LDS.
//
// No deadlock: a launch's grid is one workgroup per CU at most and at most S launches are in flight at any time (a branch is
// stream-ordered in itself); the host checks that S workgroups of consecutive launches fit a CU together (LDS, waves,
// registers), so waiting workgroups can never keep the workgroups they wait for off the chip.  Every wait is bounded (~0.3 s)
// and flags the plan instead of hanging the queue.
//
// Inside the loop NOTHING but the LDS-DMAs and the output stores may enter the vector-memory queue: the compiler does not
// see the asm DMAs, so any load it schedules there would be answered with a wait that drains the ring.  The launch record is
// therefore read through the constant address space (scalar loads).
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

#include <type_traits>

namespace tce {

namespace {

struct OvlArgs {
    const StreamLaunch *L;       // device memory; block_begin / n_rg count ROW BLOCKS of 2 * NW rows here
    const unsigned *epoch;       // the token's tag (1..65535)
    unsigned *status;            // set to 1 if a wait gave up
    unsigned long long *stamps;  // diagnostics: [workgroup][4] wall-clock stamps (100 MHz) {entered, x complete, x staged, done}; null normally
};

typedef const StreamLaunch __attribute__((address_space(4))) *ConstLaunch;  // read with s_load: uniform, and not a vector-memory operation

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int kRows = 2;                                // rows per wave and step
constexpr int kSlotBytes = kRows * 1024 + kRows * 256;  // one step of one wave: the rows' 1 KiB weight pieces, then their scale dwords

// NW waves per workgroup, D ring slots per wave, XB activation pieces per thread and batch.
// X4: what one 16-byte LDS-DMA instruction adds to the wave's vector-memory counter (experiment: 1 or 4).
template <int NW, int D, int XB, int X4>
__global__ __launch_bounds__(64 * NW, 8 / (NW / 4)) void w4a16_gemv_ovl_kernel(const OvlArgs args) {
    constexpr int kOps = kRows * X4 + kRows;  // vector-memory counter increments per step: the rows' weight DMAs and scale DMAs
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = 64 * NW;
    constexpr int ROWS = kRows;
    const ConstLaunch L = (ConstLaunch)(unsigned long long)args.L;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;

    const int K = L->K;
    const int nchunks = K >> 5;           // 16-byte chunks per weight row
    const int T = (nchunks + 63) >> 6;    // steps per row group
    const int gshift = L->log2g - 5;      // chunk -> quantization group
    const int rowbytes = nchunks * 16;
    const int n_rb = L->n_rg;
    const int nseg = L->nseg;
    const unsigned tag = __builtin_amdgcn_readfirstlane(__hip_atomic_load(args.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) << 16;
    auto stamp = [&](int i) {
        if (args.stamps && tid == 0) args.stamps[(size_t)blockIdx.x * 4 + i] = wall_clock64();
    };
    stamp(0);

    uint4_t *xs = reinterpret_cast<uint4_t *>(smem);  // [T][4][64] pieces of 16 bytes (pair-permuted, lane-linear), then the rings
    const int total_pieces = T * 256;
    // behind the image: 64 halves per wave for the wave's outputs (they leave together at the very end: a write-through store takes
    // microseconds to be acknowledged and the vector-memory counter is in-order, so a store in the middle of the stream stalls the ring)
    half_t *const obuf = reinterpret_cast<half_t *>(smem + (size_t)total_pieces * 16);  // [row block of this workgroup][2 * NW rows]
    const unsigned ring0 = (unsigned)total_pieces * 16u + (unsigned)(NW * 128) + (unsigned)wave * (unsigned)(D * kSlotBytes);  // this wave's ring (LDS byte address)
    const half_t *const A = L->A;
    const unsigned *const A_tag = L->A_tag;
    const bool tagged_in = A_tag != nullptr;  // grid-uniform: x is polled, not read
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(A), 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_t =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(tagged_in ? A_tag : reinterpret_cast<const unsigned *>(A)), 0, 0x7FFFFFF0, 0x00020000);
    auto piece_k0 = [&](int p, bool &valid) -> int {  // first activation of image piece p (clamped), valid = inside the vector
        const int c = (p >> 8) * 64 + (p & 63);
        const int j = (p >> 6) & 3;
        valid = p < total_pieces && c < nchunks;
        return valid ? c * 32 + j * 8 : 0;
    };

    // ---- 1. the activation words of this thread's first XB pieces ----
    int k0[XB];
    bool xok[XB], need[XB];
    uint4_t lo[XB], hi[XB];
    auto request = [&](int base) {
#pragma unroll
        for (int i = 0; i < XB; ++i) {
            k0[i] = piece_k0(base + tid + i * NT, xok[i]);
            if (tagged_in) {
                lo[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, k0[i] * 4, 0, /*sc0|sc1*/ 17);
                hi[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, k0[i] * 4 + 16, 0, /*sc0|sc1*/ 17);
            } else {
                lo[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, k0[i] * 2, 0, 0);
                hi[i] = lo[i];
            }
            need[i] = tagged_in && xok[i];
        }
    };
    request(0);

    // ---- 2. the weight stream: issue cursor ----
    // This workgroup's row blocks are rb = blockIdx.x + k * G; wave w owns the w-th row group (ROWS rows) of each.
    const int n_groups = n_rb > (int)blockIdx.x ? (n_rb - (int)blockIdx.x + G - 1) / G : 0;
    int i_k = 0, i_t = 0;  // next step to issue: group i_k, step i_t
    __amdgpu_buffer_rsrc_t rs_w = rs_a, rs_s = rs_a;
    int so_w0 = 0, so_w1 = 0, so_s0 = 0, so_s1 = 0;
    auto pick_seg = [&](int rb) -> int {
        int si = 0;
        if (1 < nseg && rb >= L->seg[1].block_begin) si = 1;
        if (2 < nseg && rb >= L->seg[2].block_begin) si = 2;
        if (3 < nseg && rb >= L->seg[3].block_begin) si = 3;
        return si;
    };
    auto issue_step = [&](int slot) {  // slot: compile-time constant at every call site
        const bool live = i_k < n_groups;  // wave-uniform; past the end: padding (offsets beyond num_records: zeros, no memory traffic)
        if (live && i_t == 0) {
            const int rb = (int)blockIdx.x + i_k * G;
            const int si = pick_seg(rb);
            const int N = L->seg[si].N;
            rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4_t *>(L->seg[si].qweight), 0, 0x7FFFFFF0, 0x00020000);
            rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(L->seg[si].scales), 0, 0x7FFFFFF0, 0x00020000);
            const int row0 = ((rb - L->seg[si].block_begin) * NW + wave) * ROWS;
            const int r0 = row0 < N ? row0 : N - 1, r1 = row0 + 1 < N ? row0 + 1 : N - 1;  // clamped; the stores are masked
            const int sstr = L->seg[si].scales_stride * 2;
            so_w0 = r0 * rowbytes;
            so_w1 = r1 * rowbytes;
            so_s0 = r0 * sstr;
            so_s1 = r1 * sstr;
        }
        const int c = i_t * 64 + lane;
        const int cc = c < nchunks ? c : nchunks - 1;  // tail lanes re-read the last chunk; their x image is zero
        const int oob = live ? 0 : (int)0x7FFFFFF0;
        const int vo_w = cc * 16 + oob, vo_s = (cc >> gshift) * 2 + oob;
        const unsigned l0 = ring0 + (unsigned)slot * kSlotBytes;
        unsigned keep;
        // One statement: M0 is written in the statement that uses it; s_nop 4 covers an SGPR operand fresh from a VALU write
        // (v_readfirstlane) being read by the buffer instructions; s_nop 0 between the M0 write and the LDS-DMA that reads it.
        asm volatile(
            "s_mov_b32 %[keep], m0\n\t"
            "s_nop 4\n\t"
            "s_mov_b32 m0, %[l0]\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %[vw], %[rw], %[sw0] offen nt lds\n\t"
            "s_add_u32 m0, %[l0], 1024\n\t"
            "s_nop 0\n\t"
            "buffer_load_dwordx4 %[vw], %[rw], %[sw1] offen nt lds\n\t"
            "s_add_u32 m0, %[l0], 2048\n\t"
            "s_nop 0\n\t"
            "buffer_load_ushort %[vs], %[rs], %[ss0] offen lds\n\t"
            "s_add_u32 m0, %[l0], 2304\n\t"
            "s_nop 0\n\t"
            "buffer_load_ushort %[vs], %[rs], %[ss1] offen lds\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep)
            : [l0] "s"(l0), [vw] "v"(vo_w), [vs] "v"(vo_s), [rw] "s"(rs_w), [rs] "s"(rs_s), [sw0] "s"(so_w0), [sw1] "s"(so_w1), [ss0] "s"(so_s0), [ss1] "s"(so_s1)
            : "memory", "scc");
        if (live) {
            if (++i_t == T) {
                i_t = 0;
                ++i_k;
            }
        }
    };

    // ---- 3. wait for the activations ----
    auto check_tags = [&]() -> bool {
        bool again = false;
#pragma unroll
        for (int i = 0; i < XB; ++i)
            if (need[i]) {
                const unsigned bad = ((lo[i].x ^ tag) | (lo[i].y ^ tag) | (lo[i].z ^ tag) | (lo[i].w ^ tag) | (hi[i].x ^ tag) | (hi[i].y ^ tag) |
                                      (hi[i].z ^ tag) | (hi[i].w ^ tag)) >> 16;
                need[i] = bad != 0u;
                again |= need[i];
            }
        return again;
    };
    bool gave_up = false;
    auto wait_pieces = [&](bool missing) {  // until every piece this thread asked for carries the token's tag
        int tries = 0;
        bool tight = false;
        while (missing && !gave_up) {
            // far from done: the wave looks at FOUR spots of the vector (its first, last and two inner 16-byte word groups, one per
            // lane quarter) with a pause in between; as soon as one of them carries the tag the producer's workgroups are finishing
            // and the missing pieces are polled directly, back to back
            while (!tight) {
                const int spot = ((lane & 3) * ((K - 4) / 3)) & ~3;
                const uint4_t s = __builtin_amdgcn_raw_buffer_load_b128(rs_t, spot * 4, 0, /*sc0|sc1*/ 17);
                const bool ok = (((s.x ^ tag) | (s.y ^ tag) | (s.z ^ tag) | (s.w ^ tag)) >> 16) == 0u;
                if (__builtin_amdgcn_ballot_w64(ok) != 0ull) {
                    tight = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                if ((++tries & 255) == 0) {  // ~0.3 ms: has anybody given up?  ~0.3 s: give up (the plan's status word says so)
                    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(args.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u) gave_up = true;
                    if (tries > (1 << 18)) {
                        __hip_atomic_store(args.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gave_up = true;
                    }
                    if (gave_up) break;
                }
            }
            if (gave_up) break;
#pragma unroll
            for (int i = 0; i < XB; ++i)
                if (need[i]) {
                    lo[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, k0[i] * 4, 0, /*sc0|sc1*/ 17);
                    hi[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, k0[i] * 4 + 16, 0, /*sc0|sc1*/ 17);
                }
            missing = __builtin_amdgcn_ballot_w64(check_tags()) != 0ull;
            if (++tries > (1 << 18)) {
                __hip_atomic_store(args.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = true;
            }
        }
    };
    auto write_pieces = [&](int base) {
#pragma unroll
        for (int i = 0; i < XB; ++i) {
            const int p = base + tid + i * NT;
            uint4_t v = lo[i];
            if (tagged_in)
                v = uint4_t{(lo[i].x & 0xFFFFu) | (lo[i].y << 16), (lo[i].z & 0xFFFFu) | (lo[i].w << 16), (hi[i].x & 0xFFFFu) | (hi[i].y << 16),
                            (hi[i].z & 0xFFFFu) | (hi[i].w << 16)};
            v = pair_permute(v);
            if (!xok[i]) v = uint4_t{0u, 0u, 0u, 0u};
            if (p < total_pieces) xs[p] = v;
        }
    };
    // The first look at the tags (the compiler's own wait: nothing else is in the vector-memory queue yet) comes BEFORE the ring is
    // started: a wait for later reads would sit behind the ring's DMAs, which is fine while we wait for the producer anyway.
    bool missing = tagged_in && __builtin_amdgcn_ballot_w64(check_tags()) != 0ull;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < D; ++d) issue_step(d);
    __builtin_amdgcn_sched_barrier(0);
    if (tagged_in) wait_pieces(missing);
    stamp(1);
    write_pieces(0);
    for (int base = XB * NT; base < total_pieces; base += XB * NT) {  // a K the compiled batch does not cover in one go (the host avoids it)
        request(base);
        if (tagged_in) wait_pieces(__builtin_amdgcn_ballot_w64(check_tags()) != 0ull);
        write_pieces(base);
    }
    __syncthreads();
    stamp(2);

    // ---- 4. the row groups of this wave ----
    unsigned mask_hi;  // nibble mask in a VGPR (tce_common.hpp: one scalar operand per VOP3 on gfx9)
    asm volatile("v_mov_b32 %0, 0x00F000F0" : "=v"(mask_hi));
    const unsigned magic = 0x64006400u;  // (1024.0h, 1024.0h)
    const half4_t ones = half4_t{(half_t)1.0f, (half_t)1.0f, (half_t)1.0f, (half_t)1.0f};
    float acc[ROWS][4];  // the 4 accumulator registers of the 4x4x4 MFMA; the lane's own dot product is [lane & 3]
    float corr[ROWS];    // sum over chunks of s * (1024 + 16 z) * sum_k x_k, z = 8
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        corr[i] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    }
    int c_k = 0, c_t = 0;  // compute cursor

    auto consume = [&](int slot) {
        const unsigned char *sl = smem + ring0 + (unsigned)slot * kSlotBytes;
        uint4_t w[ROWS];
        unsigned sbits[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            w[i] = *reinterpret_cast<const uint4_t *>(sl + i * 1024 + lane * 16);
            sbits[i] = *reinterpret_cast<const unsigned *>(sl + ROWS * 1024 + i * 256 + lane * 4);
        }
        const int t = c_t;
        half4_t xb[8];
        float4_t xs4 = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4_t xp = xs[(t * 4 + j) * 64 + lane];
            xb[2 * j] = __builtin_bit_cast(half4_t, uint2_t{xp.x, xp.y});      // (x0,x4,x1,x5) of word j
            xb[2 * j + 1] = __builtin_bit_cast(half4_t, uint2_t{xp.z, xp.w});  // (x2,x6,x3,x7)
            xs4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, xb[2 * j], xs4, 0, 0, 0);
            xs4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, xb[2 * j + 1], xs4, 0, 0, 0);
        }
        const float xsum = xs4[0];  // D[i][j] = sum_k B_j[k] for every i: all four registers hold this lane's sum
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            __builtin_amdgcn_sched_barrier(0);  // one row's unpack -> MFMA -> scale at a time (register pressure, see w4a16_gemv.hip)
            float4_t blk = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned wj = w[i][j];
                const unsigned t0 = ((wj << 4) & mask_hi) | magic;  // (1024+16 q0, 1024+16 q4)
                const unsigned t1 = (wj & mask_hi) | magic;         // (q1, q5)
                const unsigned t2 = ((wj >> 4) & mask_hi) | magic;  // (q2, q6)
                const unsigned t3 = ((wj >> 8) & mask_hi) | magic;  // (q3, q7)
                blk = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(half4_t, uint2_t{t0, t1}), xb[2 * j], blk, 0, 0, 0);
                blk = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(half4_t, uint2_t{t2, t3}), xb[2 * j + 1], blk, 0, 0, 0);
            }
            const float s = (float)__builtin_bit_cast(half_t, (unsigned short)(sbits[i] & 0xFFFFu));
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = __builtin_fmaf(s, blk[r], acc[i][r]);
            corr[i] = __builtin_fmaf(s * 1152.0f, xsum, corr[i]);  // 1024 + 16 * 8
        }
        // every LDS read of this slot has returned before the slot is handed back to the DMA engine
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t != T - 1) {
            c_t = t + 1;
            return;
        }
        // ---- end of a row group: reduce over the 64 lanes, park the results in LDS, reset (wave-uniform branch) ----
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            float v = corr[i] * -0.0625f;
#pragma unroll
            for (int q = 0; q < 4; ++q) v = __builtin_fmaf(acc[i][q], (lane & 3) == q ? 0.0625f : 0.0f, v);  // one-hot pick of the diagonal and the final /16
            v = wave_sum_dpp_lane63(v);  // total in lane 63
            if (lane == 63) obuf[(c_k * NW + wave) * ROWS + i] = (half_t)v;
            corr[i] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        }
        c_t = 0;
        ++c_k;
    };

    // The ring: before step s is consumed at most D - 1 younger steps may be outstanding (loads return in order); then its slot is
    // refilled with step s + D.  (Output stores enter the same queue: behind a row group's end the wait covers up to four
    // operations more than it has to, once per row group.)
    const int s_total = n_groups * T;
    for (int s = 0; s < s_total; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (s + d < s_total) {  // wave-uniform
                wait_vmcnt<(D - 1) * kOps>();
                consume(d);
            }
            issue_step(d);
        }
    }
    wait_vmcnt<0>();  // no DMA may still be on its way into this workgroup's LDS when the workgroup ends
    // ---- 5. the workgroup's outputs leave together: per row block 2 * NW consecutive rows, i.e. whole 16-byte pieces ----
    // write-through (device-scope) stores: visible to the consumers' coherent reads without a cache flush; the tagged word is the
    // hand-off itself -- value and "it is there" in ONE word, nothing waits for an acknowledgement.  (Row by row from lane 63 of
    // every wave these were 44 000 two- and four-byte fabric writes per gate + up launch; as pieces they are 2 700.)
    __syncthreads();
    if (wave == 0 && lane < n_groups * 2) {
        const int k = lane >> 1, h8 = (lane & 1) * NW;  // this lane: rows h8 .. h8 + NW - 1 of the workgroup's k-th row block (NW = 8 rows)
        static_assert(NW == 8, "a lane stores one 16-byte piece of outputs");
        const int rb = (int)blockIdx.x + k * G;
        const int si = pick_seg(rb);
        const int epi = L->seg[si].epilogue;
        half_t *const c_C = L->seg[si].C;
        unsigned *const c_T = L->C_tag[si];
        const int row0 = (rb - L->seg[si].block_begin) * (NW * ROWS) + h8;
        half8_t v = *reinterpret_cast<const half8_t *>(obuf + k * (NW * ROWS) + h8);
        auto st8 = [&](void *p, unsigned long long bits) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto tagged2 = [&](half_t a, half_t b2) -> unsigned long long {
            return (unsigned long long)(tag | (unsigned)__builtin_bit_cast(unsigned short, a)) | ((unsigned long long)(tag | (unsigned)__builtin_bit_cast(unsigned short, b2)) << 32);
        };
        if (epi & TCE_W4_SILU_MUL_PAIRS) {  // rows (2n, 2n+1) = (gate n, up n): four outputs from this lane's eight rows
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = silu_mul_half(v[2 * e], v[2 * e + 1]);
            const int idx = row0 >> 1;
            st8(c_C + idx, __builtin_bit_cast(unsigned long long, o));
            if (c_T) {
                st8(c_T + idx, tagged2(o[0], o[1]));
                st8(c_T + idx + 2, tagged2(o[2], o[3]));
            }
        } else {
            if (epi & TCE_W4_ADD_TO_C) {  // device-scope read: the residual may have been written by another CU moments ago
                unsigned long long old[2];
                old[0] = __hip_atomic_load(reinterpret_cast<unsigned long long *>(c_C + row0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                old[1] = __hip_atomic_load(reinterpret_cast<unsigned long long *>(c_C + row0 + 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const half4_t o0 = __builtin_bit_cast(half4_t, old[0]), o1 = __builtin_bit_cast(half4_t, old[1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = o0[e] + v[e];
                    v[4 + e] = o1[e] + v[4 + e];
                }
            }
            const uint4_t bits = __builtin_bit_cast(uint4_t, v);
            st8(c_C + row0, (unsigned long long)bits.x | ((unsigned long long)bits.y << 32));
            st8(c_C + row0 + 4, (unsigned long long)bits.z | ((unsigned long long)bits.w << 32));
            if (c_T) {
#pragma unroll
                for (int e = 0; e < 4; ++e) st8(c_T + row0 + 2 * e, tagged2(v[2 * e], v[2 * e + 1]));
            }
        }
    }
    stamp(3);
}

thread_local int g_ovl_chains = 0;  // tuning: graph branches the launches alternate over (0 = 2)
thread_local int g_ovl_depth = 0;   // tuning: ring slots per wave (0 = as many as fit, up to 4)
unsigned long long *g_ovl_stamps = nullptr;

thread_local int g_ovl_x4 = 1;
template <int NW, int D, int XB>
hipError_t ovl_setup(size_t lds, int *per_cu) {
    const void *kfn = g_ovl_x4 == 4 ? reinterpret_cast<const void *>(w4a16_gemv_ovl_kernel<NW, D, XB, 4>) : reinterpret_cast<const void *>(w4a16_gemv_ovl_kernel<NW, D, XB, 1>);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kfn, 64 * NW, lds);
}
template <int NW, int D, int XB>
hipError_t ovl_launch(const OvlArgs &a, int grid, size_t lds, hipStream_t stream) {
    if (g_ovl_x4 == 4) hipLaunchKernelGGL((w4a16_gemv_ovl_kernel<NW, D, XB, 4>), dim3(grid, 1, 1), dim3(64 * NW, 1, 1), lds, stream, a);
    else hipLaunchKernelGGL((w4a16_gemv_ovl_kernel<NW, D, XB, 1>), dim3(grid, 1, 1), dim3(64 * NW, 1, 1), lds, stream, a);
    return hipGetLastError();
}
#define TCE_OVL_D(FN, g, X_, ...) \
    ((g).depth == 2 ? FN<8, 2, X_>(__VA_ARGS__) : (g).depth == 3 ? FN<8, 3, X_>(__VA_ARGS__) : FN<8, 4, X_>(__VA_ARGS__))
#define TCE_OVL_DISPATCH(FN, g, ...) ((g).xb == 2 ? TCE_OVL_D(FN, g, 2, __VA_ARGS__) : TCE_OVL_D(FN, g, 4, __VA_ARGS__))

}  // namespace

void set_gemv_ovl_config(int depth, int chains) {
    g_ovl_x4 = depth >= 10 ? 4 : 1;  // experiment: depth + 10 = count a 16-byte DMA as four operations
    depth %= 10;
    g_ovl_depth = depth;
    g_ovl_chains = chains;
}
void set_gemv_ovl_stamps(void *buf) { g_ovl_stamps = static_cast<unsigned long long *>(buf); }

int ovl_chains() { return g_ovl_chains >= 1 && g_ovl_chains <= 4 ? g_ovl_chains : 2; }

// Geometry of one launch of an overlapped plan: eight waves per workgroup, one workgroup per CU at most, the ring as deep as the CU's
// LDS allows next to the other launches in flight.  Returns TCE_ERR_UNSUPPORTED_SHAPE when the launch is not one this kernel takes.
int ovl_geometry(const tce_w4a16_desc *descs, int count, int cus, int chains, OvlGeom *g, hipError_t *hip_err) {
    const tce_w4a16_desc &d0 = descs[0];
    if (d0.M != 1 || d0.rmsnorm_gamma || d0.group_size != 128) return TCE_ERR_UNSUPPORTED_SHAPE;
    const int nchunks = d0.K >> 5, T = (nchunks + 63) >> 6;
    if (T < 1) return TCE_ERR_UNSUPPORTED_SHAPE;
    g->wn = 8;
    const int pieces = T * 256, per_thread = (pieces + 64 * g->wn - 1) / (64 * g->wn);
    g->xb = per_thread <= 2 ? 2 : 4;
    g->rows_per_block = kRows * g->wn;
    g->z8 = true;
    long n_rb = 0;
    for (int i = 0; i < count; ++i) {
        // (the ring carries weights and scales only: zero point 8, what the reference quantizer writes; no residual read inside the loop)
        if (!(descs[i].flags & TCE_W4_ZERO_POINT_IS_8)) return TCE_ERR_UNSUPPORTED_SHAPE;
        // a row block's outputs leave as whole 16-byte pieces
        if (descs[i].N % g->rows_per_block != 0 || (reinterpret_cast<uintptr_t>(descs[i].C) & 15)) return TCE_ERR_UNSUPPORTED_SHAPE;
        n_rb += (descs[i].N + g->rows_per_block - 1) / g->rows_per_block;
    }
    // LDS: the x image, then the rings; `chains` workgroups (of consecutive launches) share a CU's 160 KiB
    const size_t x_bytes = (size_t)pieces * 16 + (size_t)g->wn * 128;  // + the waves' output buffers
    const size_t budget = (size_t)160 * 1024 / chains;
    int depth = g_ovl_depth >= 2 && g_ovl_depth <= 4 ? g_ovl_depth : 4;
    while (depth > 2 && x_bytes + (size_t)g->wn * depth * kSlotBytes > budget) --depth;
    g->depth = depth;
    g->lds = x_bytes + (size_t)g->wn * depth * kSlotBytes;
    if (g->lds > budget) return TCE_ERR_UNSUPPORTED_SHAPE;
    int per_cu = 0;
    const hipError_t e = TCE_OVL_DISPATCH(ovl_setup, *g, g->lds, &per_cu);
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    if (per_cu < chains) return TCE_ERR_UNSUPPORTED_SHAPE;  // a launch may never fill the chip (see the header)
    const long cap = (long)cus;  // one workgroup per CU
    const long rounds = (n_rb + cap - 1) / cap;
    if (rounds * kRows > 64) return TCE_ERR_UNSUPPORTED_SHAPE;  // a wave parks its outputs in 64 LDS slots
    g->grid = (int)((n_rb + rounds - 1) / rounds);  // every workgroup the same number of row blocks (+-1)
    g->per_cu = per_cu;
    return TCE_OK;
}

hipError_t ovl_enqueue(const StreamLaunch *L, const OvlGeom &g, const unsigned *epoch, unsigned *status, int launch_index, hipStream_t stream) {
    OvlArgs a;
    a.L = L;  // device memory
    a.epoch = epoch;
    a.status = status;
    a.stamps = g_ovl_stamps ? g_ovl_stamps + (size_t)launch_index * 4096 * 4 : nullptr;  // [launch][<= 4096 workgroups][4]
    return TCE_OVL_DISPATCH(ovl_launch, g, a, g.grid, g.lds, stream);
}

}  // namespace tce
