// attention_fast.hip -- one decode step of the reference's Llama attention block between the fused q/k/v linear and o_proj, as
// ONE launch, bandwidth-bound in the KV cache ("flash-decoding").
//
// What it replaces per token and layer (llm/src/nn_modules/cuda/Int4llamaAttention.cu:116-229):
//     shape_qkv_cuda (:41-64, 130-136)          the fused projection's row [3 * heads * hd] is read in place (q | k | v, head-major)
//     RotaryPosEmb_cuda_forward (:157-159)      applied to q and the new k while they are loaded -- the reference's binary16
//                                               arithmetic hfma(x, cos, hmul(rot, sin)) (llm/src/ops/cuda/RotaryPosEmb.cu:4-34): the
//                                               key that enters the cache is bit-identical to the reference's
//     the KV append (:161-181)                  the reference copies the WHOLE past into the layer's other cache buffer every token
//                                               (4 cudaMemcpyAsync per head, O(t)); here the cache is one fixed-capacity array
//                                               [heads][max_keys][hd] per K and V and the new row is written at index `pos`
//     qk_bmm -> batch_Add -> check_inf_half -> softmax -> transpose_1_2idx -> pv_bmm -> unshape (:184-217)
//                                               scores, online softmax and the weighted sum of V rows in fp32, the key range cut into
//                                               chunks over workgroups (heads x chunks fills the chip); V is read in the layout it was
//                                               appended in (no transposed copy); the output row [heads * hd] is o_proj's input
// The bit-exact form of the same block -- binary16 accumulation chains in the reference's order -- stays available as
// tce_attention_decode_f16 (attention_ops.hip); that one is what parity is claimed with.  THIS kernel computes the same function
// in fp32 and is checked against it with a stated tolerance (tests/test_gpu_attention.py): |out - ref| <= 2e-3 * max|ref| per head
// + one binary16 ulp, i.e. the difference between fp32 and binary16 accumulation, not a different algorithm.
//
// Work decomposition: workgroup = (head, chunk of keys), 4 waves; a wave takes 4 keys per step -- lane = (key slot = lane / 16,
// piece = lane % 16): one 16-byte load per lane covers 4 consecutive cache rows completely (1 KiB contiguous), the dot product is 4
// v_dot2_f32_f16 per lane plus a 4-step DPP sum over the 16 lanes of a row, every lane of the row then holds the score and
// rescales its own 8 output dimensions (online softmax state per (wave, key slot)).  The 16 states of a workgroup are merged
// through LDS, the chunks of a head through a small fp32 workspace by the last workgroup to arrive (write-through stores +
// device-scope loads: MI355X_MICROARCH.md, inter-workgroup visibility; no cache-wide fence).
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

namespace {

struct FastAttnArgs {
    const half_t *qkv;    // [q_heads + 2 * kv_heads][hd]: q heads, then k heads, then v heads (the fused projection's row)
    half_t *kc, *vc;      // [kv_heads][max_keys][hd]
    const half_t *cosv, *sinv;  // [positions][hd] or null (no RoPE: q / k used as they are)
    const half_t *mask;   // [keys] additive or null
    half_t *out;          // [q_heads][hd]
    float *part;          // [q_heads][chunks][2 + hd]
    unsigned *cnt;        // [kv_heads], zero between launches
    int heads, kv_heads, rep, hd, max_keys, pos, keys, chunk, chunks;  // heads = query heads, rep = heads / kv_heads; chunks = chunk slots of the grid
    const int *pos_dev;   // non-null: the position is read from this device word (a captured launch replayed token after token); pos / keys above
                          // then only bound it (the grid and the chunk length were cut for them)
    float alpha;
    int defer;             // round 5: with several chunk slots the launch ENDS at its partial states -- plain stores of (M, L, -, -, O[hd]) per (query head, chunk slot),
                           // stride kDeferStride floats -- and the consumer (the o_proj launch: w4a16_gemv_i8.hip's COMB prologue) combines them while it stages its
                           // activations.  No acknowledged stores, no counter, no last workgroup: the kernel boundary orders the two launches
    int probe_no_combine;  // timing experiment (tce_w4a16_set_debug_mode(2931)): the partial states are stored plainly and the launch ends -- `out` is NOT written
};

__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row, result in every lane of the row
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
    return v;
}

// acc + p * (the low / high binary16 half of `pair`, widened exactly): one v_fma_mix_f32, no separate conversion
__device__ __forceinline__ float fma_mix_lo(float p, unsigned pair, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(p), "v"(pair), "v"(acc));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(float p, unsigned pair, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(p), "v"(pair), "v"(acc));
    return d;
}

// RotaryPosEmb_cuda_forward on the 8 elements of `piece` (hd = 128: the partner half is piece ^ 8):
//   out[j] = hfma(x[j], cos[j], hmul(rot[j], sin[j])),  rot[j] = j < hd/2 ? -x[j + hd/2] : x[j - hd/2]
// v = the piece, p = the partner piece, c / s = the cos / sin pieces (all loaded by the caller, in one batch)
__device__ __forceinline__ half8_t rope_apply(const half8_t v, const half8_t p, const half8_t c, const half8_t s, int piece) {
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const half_t rot = piece < 8 ? (half_t)(-p[e]) : p[e];
        const half_t t = rot * s[e];                       // hmul
        o[e] = __builtin_fmaf16(v[e], c[e], t);            // hfma: v_fma_f16, one rounding (as tce_rope_half, attention_ops.hip)
    }
    return o;
}

constexpr int kHD = 128;
constexpr float kNegBig = -1.0e30f;
constexpr int kDeferStride = 4 + kHD;  // floats per deferred partial state: M, L, two unused, O[128] (16-byte aligned rows of O)

// MASK: the caller gave a mask row (a compile-time form: a branch inside the fetch block makes hipcc drain the load queue at the loop head)
// NW: waves per workgroup (4; 8 and 16 exist for the sweep that ruled them out, see pick_chunk).
// R: query heads per key / value head (grouped-query attention, llm/src/nn_modules/non_cuda/Int4llamaAttention.cc:166-185: query head i reads
//    key / value head i / R; Llama-3-8B: 32 / 8, llm/include/model.h:83).  A workgroup is (key / value head, chunk of keys): it streams the
//    chunk's cache rows ONCE and keeps R online-softmax states per lane, so the R query heads cost one pass over the cache, not R.
template <bool MASK, int NW, int R>
__global__ __launch_bounds__(64 * NW) void attn_decode_fast_kernel(const FastAttnArgs a) {
    constexpr int NT = 64 * NW, NS = 4 * NW;
    __shared__ __attribute__((aligned(16))) float st[NS][R][2 + kHD];  // the (wave, slot) states per query head: m, l, o[hd]
    __shared__ __attribute__((aligned(16))) half_t newrow[2][kHD];  // the token's own (rotated) key and value
    __shared__ unsigned last_flag;
    // grp: this workgroup's group of R consecutive query heads (R == rep: all the query heads of a key / value head, its cache rows streamed
    // once for all of them; R < rep: rep / R workgroups read the same cache rows -- from HBM once, the others from the memory-side cache)
    const int grp = blockIdx.x / a.chunks, c = blockIdx.x - grp * a.chunks;
    // the position: by value, or from a device word (wave-uniform scalar load) -- then chunks past the context have nothing to do and the
    // head's combine expects only the chunks that exist
    const int pos = a.pos_dev ? __builtin_amdgcn_readfirstlane(*a.pos_dev) : a.pos;
    const int keys = pos + 1;
    const int chunks = a.pos_dev ? (keys + a.chunk - 1) / a.chunk : a.chunks;  // active chunks (<= the grid's chunk slots)
    if (c >= chunks) return;
    const int head = (grp * R) / a.rep;  // the key / value head
    const bool appends = (grp * R) % a.rep == 0;  // one workgroup group per key / value head writes the token's row into the caches
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = lane >> 4, piece = lane & 15;
    const int key0 = c * a.chunk, key1 = key0 + a.chunk < keys ? key0 + a.chunk : keys;
    const half_t *cosr = a.cosv ? a.cosv + (size_t)pos * kHD : nullptr, *sinr = a.sinv ? a.sinv + (size_t)pos * kHD : nullptr;
    const size_t hoff = (size_t)head * kHD;
    // ---- this wave's keys: chunk / 4 consecutive ones, 4 per step.  Their addresses depend on nothing but the arguments, so the
    //      first block of cache rows is requested in the same batch as q / cos / sin: one memory round trip per launch instead of
    //      two -- and BEHIND them (round 4): loads return in order, the prologue's pieces are L2 hits, and with the cache rows in
    //      front of them the rotation waited for HBM (4.95 -> 4.5 us at 128 keys, 7.8 -> 7.5 at 512, 10.2 -> 9.8 at 2048) ----
    const int per_wave = a.chunk / NW;  // (the host made the chunk a multiple of 4 * NW)
    const int kw0 = key0 + wave * per_wave;
    const int kw1 = kw0 + per_wave < key1 ? kw0 + per_wave : key1;  // a block is 16 keys, a wave's run any multiple of 4: the rest weighs nothing
    const half_t *kbase = a.kc + (size_t)head * a.max_keys * kHD, *vbase = a.vc + (size_t)head * a.max_keys * kHD;
    // blocks of 4 steps (16 keys per wave): the 8 loads of the next block are in flight while this block's scores and
    // exponentials are computed (the online-softmax state is the only loop-carried dependence; without the explicit double
    // buffer every step paid a full memory round trip: 20 us at 2048 keys, profiles/r2/attention_decode_step.jsonl)
    constexpr int BLK = 4;
    half8_t kbuf[2][BLK], vbuf[2][BLK];
    half_t mbuf[2][BLK];  // the keys' mask values travel with their rows (a load per step inside consume() drained the queue)
    auto fetch = [&](half8_t (&kd)[BLK], half8_t (&vd)[BLK], half_t (&md)[BLK], int it0) {
#pragma unroll
        for (int u = 0; u < BLK; ++u) {
            const int key = kw0 + it0 + u * 4 + slot;
            const int kk = key < kw1 ? key : (keys - 1);  // clamped: rows past the range are read (harmlessly) and weigh nothing
            kd[u] = *reinterpret_cast<const half8_t *>(kbase + (size_t)kk * kHD + piece * 8);
            vd[u] = *reinterpret_cast<const half8_t *>(vbase + (size_t)kk * kHD + piece * 8);
            if constexpr (MASK) md[u] = a.mask[kk];
            else md[u] = (half_t)0;
        }
    };
    // ---- every piece the prologue needs, requested together: q, k (with their partner halves), v, cos, sin.  All four
    //      waves fetch the k / v pieces (L2 hits, 3 instructions) so that nobody waits for a second batch; wave 0 uses them ----
    const bool rope = cosr != nullptr;
    const half_t *xq = a.qkv + (size_t)grp * R * kHD, *xk = a.qkv + (size_t)a.heads * kHD + hoff, *xv = a.qkv + (size_t)(a.heads + a.kv_heads) * kHD + hoff;
    const half_t *cp = rope ? cosr : xq, *sp = rope ? sinr : xq;  // no rotation: harmless repeats of the q piece, not used
    auto ld8 = [](const half_t *ptr) { return *reinterpret_cast<const half8_t *>(ptr); };
    half8_t q_v[R], q_p[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        q_v[r] = ld8(xq + r * kHD + piece * 8);
        q_p[r] = ld8(xq + r * kHD + (piece ^ 8) * 8);
    }
    const half8_t cc = ld8(cp + piece * 8), ss = ld8(sp + piece * 8);
    const half8_t k_v = ld8(xk + piece * 8), k_p = ld8(xk + (piece ^ 8) * 8), v_v = ld8(xv + piece * 8);
    __builtin_amdgcn_sched_barrier(0);
    fetch(kbuf[0], vbuf[0], mbuf[0], 0);  // (the row at index pos may not be in the cache yet: consume() takes it from LDS)
    // (Round 4, tried: the SECOND block requested here too, so that a wave's first 32 keys cost one memory round trip instead of two.  Slower at every context, in
    // front of the prologue's loads or behind them -- 7.2 -> 7.7 us at 512 keys, 9.05 -> 9.6 at 1024; and again after consume() was made cheaper, as a ring of three
    // buffers with two blocks in flight: 4.35 -> 4.67 us at 128 keys, 7.2 -> 7.55 at 512, a tie on 256-key-per-wave runs.  Same-session A/Bs with
    // scripts/probes/attn_quick.py, profiles/r4/attention_block_softmax_ab.jsonl.)
    __builtin_amdgcn_sched_barrier(0);
    // ---- the R query heads (rotated), this lane's 8 dimensions of each ----
    half8_t qh[R];
#pragma unroll
    for (int r = 0; r < R; ++r) qh[r] = rope ? rope_apply(q_v[r], q_p[r], cc, ss, piece) : q_v[r];
    // ---- the new key / value of this head: into LDS for this workgroup's use, into the cache by the workgroup that owns index pos ----
    if (wave == 0) {
        const half8_t kh = rope ? rope_apply(k_v, k_p, cc, ss, piece) : k_v;
        const half8_t vh = v_v;
        if (slot == 0) {
            *reinterpret_cast<half8_t *>(&newrow[0][piece * 8]) = kh;
            *reinterpret_cast<half8_t *>(&newrow[1][piece * 8]) = vh;
            if (appends && pos >= key0 && pos < key1) {
                *reinterpret_cast<half8_t *>(a.kc + ((size_t)head * a.max_keys + pos) * kHD + piece * 8) = kh;
                *reinterpret_cast<half8_t *>(a.vc + ((size_t)head * a.max_keys + pos) * kHD + piece * 8) = vh;
            }
        }
    }
    __syncthreads();
    float m[R], l[R], acc[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        m[r] = kNegBig;
        l[r] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[r][e] = 0.f;
    }
    // One block = 4 steps = 16 keys of the wave.  The online-softmax state moves ONCE per block (round 4): the block's scores first, one new maximum, one rescale
    // of the accumulators, then the four weighted value rows as fused multiply-adds that take the binary16 value straight from its register (v_fma_mix_f32).  The
    // step-by-step form this replaces -- a rescale, two exponentials and 8 conversions per step -- was bound by exactly that arithmetic: 58 vector instructions per
    // step, one wave per SIMD, 0.62 us per block against a memory round trip of about the same length (profiles/r4/attention_block_softmax_ab.jsonl).
    auto consume = [&](const half8_t (&kd)[BLK], const half8_t (&vd)[BLK], const half_t (&md)[BLK], int it0) {
        half8_t kk[BLK], vv[BLK];
        float sco[R][BLK];
        bool valid[BLK];
#pragma unroll
        for (int u = 0; u < BLK; ++u) {
            kk[u] = kd[u];
            vv[u] = vd[u];
            valid[u] = true;
        }
        // the rare blocks -- the one that runs past the wave's range, the one that holds the token's own row -- are told apart by a wave-uniform test, so the
        // others carry no per-slot validity arithmetic at all
        const int b0 = kw0 + it0;
        if (b0 + BLK * 4 > kw1 || (pos >= b0 && pos < b0 + BLK * 4)) {
#pragma unroll
            for (int u = 0; u < BLK; ++u) {
                const int key = b0 + u * 4 + slot;
                valid[u] = key < kw1;
                if (key == pos) {  // the token's own row: not necessarily visible in the cache yet
                    kk[u] = *reinterpret_cast<const half8_t *>(&newrow[0][piece * 8]);
                    vv[u] = *reinterpret_cast<const half8_t *>(&newrow[1][piece * 8]);
                }
                // a slot past the range was loaded from a row that may hold anything (the row at `pos` before this launch wrote it, an
                // uninitialised cache): its weight is 0, and 0 * inf would still be NaN -- the value row is zeroed, not just weighted
                if (!valid[u]) vv[u] = half8_t{(half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0, (half_t)0};
            }
        }
#pragma unroll
        for (int u = 0; u < BLK; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e += 2) d = __builtin_amdgcn_fdot2(half2_t{qh[r][e], qh[r][e + 1]}, half2_t{kk[u][e], kk[u][e + 1]}, d, false);
                d = row16_sum(d);
                float sv = a.alpha * d;
                if constexpr (MASK) sv += (float)md[u];
                if (!(__builtin_fabsf(sv) <= 65504.0f)) sv = -65504.0f;  // check_inf_half (Int4llamaAttention.cu:105-115): inf / nan / beyond binary16 -> -65504
                if (!valid[u]) sv = kNegBig;
                sco[r][u] = sv;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float bm = __builtin_fmaxf(__builtin_fmaxf(sco[r][0], sco[r][1]), __builtin_fmaxf(sco[r][2], sco[r][3]));
            const float mn = __builtin_fmaxf(m[r], bm);
            const float sc = __expf(m[r] - mn);
            float p[BLK];
#pragma unroll
            for (int u = 0; u < BLK; ++u) p[u] = valid[u] ? __expf(sco[r][u] - mn) : 0.f;  // (a block of nothing but invalid slots on a fresh state: mn == kNegBig, the difference is 0)
            m[r] = mn;
            l[r] = l[r] * sc + ((p[0] + p[1]) + (p[2] + p[3]));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r][e] *= sc;
#pragma unroll
            for (int u = 0; u < BLK; ++u) {
                const uint4_t vw = __builtin_bit_cast(uint4_t, vv[u]);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    acc[r][e] = fma_mix_lo(p[u], vw[e >> 1], acc[r][e]);
                    acc[r][e + 1] = fma_mix_hi(p[u], vw[e >> 1], acc[r][e + 1]);
                }
            }
        }
    };
    // per_wave is a multiple of 4 (steps); blocks of 4 steps, the last one possibly past the range (clamped loads, zero weights)
    for (int it = 0; it < per_wave; it += 2 * BLK * 4) {
        fetch(kbuf[1], vbuf[1], mbuf[1], it + BLK * 4);
        consume(kbuf[0], vbuf[0], mbuf[0], it);
        if (it + BLK * 4 >= per_wave) break;
        fetch(kbuf[0], vbuf[0], mbuf[0], it + 2 * BLK * 4);
        consume(kbuf[1], vbuf[1], mbuf[1], it + BLK * 4);
    }
    // ---- merge the workgroup's 4 * NW states, per query head ----
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float *s_ = st[wave * 4 + slot][r];
        if (piece == 0) {
            s_[0] = m[r];
            s_[1] = l[r];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s_[2 + piece * 8 + e] = acc[r][e];
    }
    __syncthreads();
    float M[R], Lq[R], O[R];  // thread d < 128 owns output dimension d of every query head
    if (tid < kHD) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float Mx = kNegBig;
#pragma unroll
            for (int i = 0; i < NS; ++i) Mx = __builtin_fmaxf(Mx, st[i][r][0]);
            float Lx = 0.f, Ox = 0.f;
#pragma unroll 16
            for (int i = 0; i < NS; ++i) {
                const float w = __expf(st[i][r][0] - Mx);
                Lx += st[i][r][1] * w;
                Ox += st[i][r][2 + tid] * w;
            }
            M[r] = Mx;
            Lq[r] = Lx;
            O[r] = Ox;
        }
    }
    const size_t qoff = (size_t)grp * R * kHD;  // the first of this workgroup's query heads in `out`
    if (chunks == 1) {
        if (tid < kHD) {
#pragma unroll
            for (int r = 0; r < R; ++r) a.out[qoff + r * kHD + tid] = (half_t)(O[r] / Lq[r]);
        }
        return;
    }
    if (a.defer) {  // the combine happens in the next launch's prologue (same arithmetic, same order: tce_w4a16_forward_deferred_attention)
        if (tid < kHD) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float *mine = a.part + ((size_t)(grp * R + r) * a.chunks + c) * kDeferStride;
                mine[4 + tid] = O[r];
                if (tid == 0) {
                    mine[0] = M[r];
                    mine[1] = Lq[r];
                }
            }
        }
        return;
    }
    if (a.probe_no_combine) {  // (round 5 probe: what the launch costs without its combine -- the upper bound of moving the combine into the next launch's prologue)
        if (tid < kHD) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float *mine = a.part + ((size_t)(grp * R + r) * chunks + c) * (2 + kHD);
                mine[2 + tid] = O[r];
                if (tid == 0) {
                    mine[0] = M[r];
                    mine[1] = Lq[r];
                }
            }
        }
        return;
    }
    // ---- several chunks per head: partial (M, L, O) to the workspace, the last workgroup to arrive combines ----
    if (tid < kHD) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float *mine = a.part + ((size_t)(grp * R + r) * chunks + c) * (2 + kHD);
            __hip_atomic_store(mine + 2 + tid, O[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through (sc1) stores
            if (tid == 0) {
                __hip_atomic_store(mine, M[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + 1, Lq[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores are acknowledged before the workgroup arrives
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.cnt + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old == (unsigned)chunks - 1 ? 1u : 0u;
        if (last_flag) __hip_atomic_store(a.cnt + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    if (!last_flag) return;
    // The partials were stored write-through; they are read back with device-coherent (sc0 sc1) buffer loads -- plain loads as
    // far as the compiler is concerned, so all of a thread's loads are in flight together (a loop of relaxed atomic loads is a
    // chain of round trips: 1 us per chunk, measured).  Thread i < chunks fetches (M_i, L_i), every thread d < hd its O_i[d].
    constexpr int kMaxChunksUnrolled = 16;
    float *ml = &st[0][0][0];  // [R][chunks][2], reuses the state area
    const int stride = (2 + kHD) * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part + (size_t)grp * R * chunks * (2 + kHD), 0, (int)((size_t)R * chunks * stride), 0x00020000);
    for (int i = tid; i < R * chunks; i += NT) {
        ml[2 * i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i * stride, 0, /*sc0|sc1*/ 17));
        ml[2 * i + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i * stride + 4, 0, 17));
    }
    float oi[R][kMaxChunksUnrolled];
    if (tid < kHD) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < kMaxChunksUnrolled; ++i)
                oi[r][i] = i < chunks ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (r * chunks + i) * stride + (2 + tid) * 4, 0, 17)) : 0.f;
    }
    __syncthreads();
    if (tid < kHD) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float *mlr = ml + 2 * r * chunks;
            float Mx = kNegBig;
            for (int i = 0; i < chunks; ++i) Mx = __builtin_fmaxf(Mx, mlr[2 * i]);
            float Lx = 0.f, Ox = 0.f;
#pragma unroll
            for (int i = 0; i < kMaxChunksUnrolled; ++i) {
                if (i < chunks) {
                    const float w = __expf(mlr[2 * i] - Mx);
                    Lx += mlr[2 * i + 1] * w;
                    Ox += oi[r][i] * w;
                }
            }
            for (int i = kMaxChunksUnrolled; i < chunks; ++i) {  // many chunks: the rest one by one
                const float w = __expf(mlr[2 * i] - Mx);
                Lx += mlr[2 * i + 1] * w;
                Ox += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (r * chunks + i) * stride + (2 + tid) * 4, 0, 17)) * w;
            }
            a.out[qoff + r * kHD + tid] = (half_t)(Ox / Lx);
        }
    }
}

}  // namespace

// chunk of keys per workgroup: heads x chunks should cover the chip a couple of times; a multiple of 16 (4 waves x 4 keys per step)
static thread_local int g_attn_target_wgs = 0;  // 0: the fitted rule below; tuning: tce_w4a16_set_debug_mode(3000 + workgroups)
void set_attention_fast_target(int wgs) { g_attn_target_wgs = wgs >= 32 && wgs <= 8192 ? wgs : 0; }

// Measured (scripts/attention_step_sweep.py, profiles/r2/attention_step_sweep.jsonl; 32 heads, caches rotating through > 256 MB so the
// keys come from HBM).  Two costs pull against each other: a workgroup streams its keys at ~60 GB/s (2.2 us per 256 keys), and
// combining the chunks of a head costs ~2 us whatever their number (the acknowledged-store -> counter -> coherent-read chain):
//   128 keys: one chunk 4.5 us, two 6.0          256: one chunk 6.5, two or four 7.0        512: 8.9 / 8.6 / 7.6 / 8.5 for 1 / 2 / 4 / 8 chunks
//   1024: 9.0 with 4 chunks, 9.3 with 8, 12.5 with 16      2048: 12.0 / 11.8 / 13.7 with 4 / 8 / 16      4096: 18.6 / 16.7 / 18.3
//   8192: 30.0 / 28.9 / 33.4 with 8 / 16 / 32
// Rule: up to 320 keys one chunk per head and no combine; four chunks up to 1024 keys (up to 640 with 4+ query heads per key / value head: round 4); beyond, eight chunks of at most 512 keys.
// A combine without the acknowledged-store -> counter -> coherent-read chain (the head's last workgroup polling (value, tag) pairs)
// was tried and was SLOWER (13.2 us at 2048 keys): a poll is a full memory round trip, and the chain it replaces is three of them
// only on the LAST workgroup.  So was the hybrid -- (value, tag) pairs stored without waiting for acknowledgements, the counter only
// electing who combines, the elected workgroup re-reading the pairs whose tag is not there yet: 7.1-8.0 / 10.5-11.4 / 13.7-14.6 us at
// 128 / 512 / 2048 keys against 6.0 / 8.6 / 11.8 (profiles/r2/attention_merge_variants.jsonl): a write-through store takes longer to
// become visible to another XCD than the counter's round trip, so the first read pass misses and every further pass is a round
// trip of its own.
static thread_local int g_attn_waves = 0;  // 0: by the chunk's length; tuning: tce_w4a16_set_debug_mode(2900 + 4 / 8 / 16)
void set_attention_fast_waves(int nw) { g_attn_waves = (nw == 4 || nw == 8 || nw == 16) ? nw : 0; }

static thread_local int g_attn_probe_no_combine = 0;
void set_attention_fast_probe_no_combine(int on) { g_attn_probe_no_combine = on ? 1 : 0; }
static thread_local int g_attn_fuse = 0;  // tuning: query heads per workgroup for grouped-query attention (0: the rule; 1, 2, 4)
void set_attention_fast_fuse(int r) { g_attn_fuse = (r == 1 || r == 2 || r == 4) ? r : 0; }

// `heads` here = workgroup groups (query heads / heads per workgroup)
static void pick_chunk(int heads, int keys, int rep, int *chunk_out, int *waves_out) {
    int chunk;
    if (g_attn_target_wgs > 0) {
        const int target_chunks = heads >= g_attn_target_wgs ? 1 : g_attn_target_wgs / heads;
        chunk = (keys + target_chunks - 1) / target_chunks;
        if (chunk < 64) chunk = 64;
    } else if (keys <= 320) {
        chunk = keys;
    } else if (keys <= (rep >= 4 ? 640 : 1024)) {  // (round 4, after the block softmax, 4 query heads per key / value head: 1024 keys 8.7 us with four chunks, 8.2 with eight; 512 keys 7.25 / 7.45.
                                                   //  One query head per key / value head -- four times the cache bytes per query head --: 768 / 1024 keys 8.4 / 8.75 with four, 8.9 / 9.2 with eight)
        chunk = (keys + 3) >> 2;
    } else {
        chunk = (keys + 7) >> 3;
        if (chunk > 512) chunk = 512;
    }
    if (chunk > 1024) chunk = 1024;
    // Waves per workgroup: 4.  More waves on the same chunk (a shorter chain of 16-key blocks per wave, no extra partials) measured
    // SLOWER at every context -- 128 keys 4.5 / 5.2 / 7.4 us with 4 / 8 / 16 waves, 2048 keys 11.7 / 12.1 / 14.4
    // (profiles/r2/attention_step_waves_sweep.jsonl): the launch is at the floor of a dependent load -> compute -> store launch
    // (4.5 us; the 8 MiB GEMV's is 4.1) plus ~2 us for the combine plus the keys at 6.4 TB/s, and wider workgroups only add to the
    // fixed part (prologue loads per wave, barriers, the workgroup's own merge over 4 x waves states).
    int nw = g_attn_waves ? g_attn_waves : 4;
    chunk = (chunk + 4 * nw - 1) / (4 * nw) * (4 * nw);
    *chunk_out = chunk;
    *waves_out = nw;
}

// the cut launch_attention_decode_fast would use for `keys` keys (no HIP call)
// Query heads per workgroup for `rep` query heads per key / value head.  Measured (scripts/attention_gqa_sweep.py, 32 over 8 heads): the step is
// bound by latency and by the softmax / weighted-sum arithmetic per (query head, key), not by the cache bytes, so fusing the four query
// heads of a key / value head into one workgroup (a quarter of the bytes, a quarter of the workgroups, four times the arithmetic each) is
// SLOWER than one query head per workgroup reading the shared rows: 8.3 / 13.9 / 20.0 us against 4.6 / 7.9 / 11.9 at 128 / 512 / 2048 keys.
static int pick_fuse(int rep) {
    if (g_attn_fuse && rep % g_attn_fuse == 0) return g_attn_fuse;
    return 1;
}

void describe_attention_decode_fast(int heads, int keys, int *chunk, int *chunks, int *waves, int kv_heads) {
    const int rep = kv_heads > 0 ? heads / kv_heads : 1;
    pick_chunk(heads / pick_fuse(rep), keys, rep, chunk, waves);
    *chunks = (keys + *chunk - 1) / *chunk;
}

size_t attention_decode_workspace_bytes(int heads, int max_keys, int hd) {
    if (heads <= 0 || max_keys <= 0 || hd != kHD) return 0;
    const int chunk = 64;  // the smallest chunk bounds the number of partials
    const size_t chunks = (size_t)(max_keys + chunk - 1) / chunk;
    const size_t cnt_bytes = ((size_t)heads * 4 + 255) & ~(size_t)255;
    return cnt_bytes + (size_t)heads * chunks * kDeferStride * 4;  // (the deferred layout's stride; the combined form's 2 + hd fits inside)
}

int launch_attention_decode_fast(const void *qkv, void *kc, void *vc, const void *cosv, const void *sinv, const void *mask, void *out, void *workspace,
                                 int heads, int kv_heads, int hd, int max_keys, int pos, unsigned short alpha_bits, hipStream_t stream, hipError_t *hip_err,
                                 const int *pos_dev, AttnDeferred *deferred) {
    if (hd != kHD || kv_heads <= 0 || heads % kv_heads != 0) return TCE_ERR_UNSUPPORTED_SHAPE;
    const int rep = heads / kv_heads;

    FastAttnArgs a{};
    a.qkv = static_cast<const half_t *>(qkv);
    a.kc = static_cast<half_t *>(kc);
    a.vc = static_cast<half_t *>(vc);
    a.cosv = static_cast<const half_t *>(cosv);
    a.sinv = static_cast<const half_t *>(sinv);
    a.mask = static_cast<const half_t *>(mask);
    a.out = static_cast<half_t *>(out);
    const size_t cnt_bytes = ((size_t)heads * 4 + 255) & ~(size_t)255;
    a.cnt = static_cast<unsigned *>(workspace);
    a.part = reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + cnt_bytes);
    a.heads = heads;
    a.kv_heads = kv_heads;
    a.rep = rep;
    const int fuse = pick_fuse(rep);
    a.hd = hd;
    a.max_keys = max_keys;
    a.pos = pos;
    a.keys = pos + 1;
    a.pos_dev = pos_dev;
    int nw = 4;
    pick_chunk(heads / fuse, a.keys, rep, &a.chunk, &nw);
    a.chunks = (a.keys + a.chunk - 1) / a.chunk;
    if (a.chunks > 1024) return TCE_ERR_UNSUPPORTED_SHAPE;  // (the combine's LDS image; unreachable with the fitted rule below 500k keys)
    half_t ah;
    __builtin_memcpy(&ah, &alpha_bits, 2);
    a.alpha = (float)ah;
    a.probe_no_combine = g_attn_probe_no_combine;
    if (deferred) {
        // deferred combine: 2 .. kDeferMaxSlots chunk slots (one slot: the launch writes `out` itself, as it does whenever only one chunk of the bound is live)
        a.defer = a.chunks >= 2 && a.chunks <= kAttnDeferMaxSlots ? 1 : 0;
        deferred->slots = a.defer ? a.chunks : 1;
        deferred->chunk = a.chunk;
        deferred->heads = heads;
        deferred->stride = kDeferStride;
        deferred->part = a.part;
    }
    const dim3 grid((heads / fuse) * a.chunks);
    auto go = [&](auto has_mask) {
        constexpr bool MK = decltype(has_mask)::value;
        if (fuse == 4) hipLaunchKernelGGL((attn_decode_fast_kernel<MK, 4, 4>), grid, dim3(256), 0, stream, a);
        else if (fuse == 2) hipLaunchKernelGGL((attn_decode_fast_kernel<MK, 4, 2>), grid, dim3(256), 0, stream, a);
        else if (nw == 16) hipLaunchKernelGGL((attn_decode_fast_kernel<MK, 16, 1>), grid, dim3(1024), 0, stream, a);
        else if (nw == 8) hipLaunchKernelGGL((attn_decode_fast_kernel<MK, 8, 1>), grid, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((attn_decode_fast_kernel<MK, 4, 1>), grid, dim3(256), 0, stream, a);
    };
    if (a.mask) go(std::true_type{});
    else go(std::false_type{});
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
