// attention_prefill.hip -- the reference's Llama attention block between the fused q/k/v linear and o_proj for m > 1 new rows
// (a prompt, or a chunk of one on top of an existing context): the prefill form of attention_fast.hip's decode step.
//
// What it replaces (llm/src/nn_modules/cuda/Int4llamaAttention.cu:116-229, the same code path as decode, sqlen > 1):
//     shape_qkv_cuda (:41-64, 130-136)          the fused projection's rows [m][(heads + 2 kv_heads) * hd] are read where they lie
//     RotaryPosEmb_cuda_forward (:157-159)      the reference's binary16 arithmetic hfma(x, cos, hmul(rot, sin)) (RotaryPosEmb.cu:4-34) on q and
//                                               the new keys: the keys that enter the cache are bit-identical to the reference's (and to what m
//                                               decode steps would append)
//     the KV append (:161-181)                  rows pos .. pos + m - 1 of the fixed-capacity caches [kv_heads][max_keys][hd] are written; nothing is copied
//     qk_bmm -> batch_Add -> check_inf_half -> softmax -> transpose_1_2idx -> pv_bmm -> unshape (:184-217)
//                                               one pass over the keys per (query head, 64 query rows): scores on the matrix pipe
//                                               (v_mfma_f32_16x16x32_f16, fp32 accumulate), online softmax in fp32, probabilities x values on the
//                                               matrix pipe, the output rows [m][heads * hd] are o_proj's input
// Two launches: `prepare` (element-wise: rotation + append + the rotated queries into a workspace) and the attention kernel.
//
// This is NOT the reference's arithmetic (binary16 accumulation chains, one thread per output: tce_bmm_f16t + tce_softmax_half reproduce those bit for
// bit and stay the compat path); it computes the same function with fp32 accumulation and is held to a float64 evaluation with a stated tolerance
// (tests/test_gpu_attention.py), like the decode step.
//
// Attention kernel: workgroup = (block of 64 or 128 query rows, query head), 4 or 8 waves x 16 query rows.  Per tile of 64 keys:
//   * the K tile [64 keys][128] is staged in LDS as it lies (row stride 272 bytes: the MFMA B fragments -- key = lane % 16, 8 consecutive head
//     dimensions -- are conflict-free 16-byte reads); S = Q K^T: 4 key tiles x 4 k-steps of 32 head dimensions, the Q fragments stay in registers;
//   * scale, additive mask, causal cut, online softmax in fp32 in registers (log2 units: one v_exp_f32 per probability; the validity tests only on tiles
//     that hold keys past the context or on the causal diagonal);
//   * everything is computed transposed (see the kernel): S^T = K Q^T leaves ONE query row per lane, so P never touches LDS -- the S^T accumulator
//     layout is the B-operand layout of v_mfma_f32_16x16x16_f16 -- and O^T += V^T P^T;
//   * the V tile is staged as it lies too (row stride 288 bytes); the P V product contracts over keys and an MFMA operand wants its contraction index
//     contiguous per lane, which a row-major V does not give: gfx950's LDS transpose read (ds_read_b64_tr_b16) hands each lane 4 keys of one head dimension
//     out of a [4 keys][16 dims] block.  (The first version transposed V while staging it -- eight 2-byte LDS writes per 16 bytes, the global loads one
//     row per lane -- and ran 158 us at 2048 rows; this one 130.)
// Grouped-query attention: query head i reads key / value head i / (heads / kv_heads); the workgroups of a group stage the same tiles (L2).
// Rows of the caches at and beyond pos + m may hold anything (uninitialised memory): staged as zeros, and their scores are cut.
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

namespace {

constexpr int kHD = 128;
constexpr int kBK = 64;
constexpr int kKStride = 272;  // bytes per key row of the K tile in LDS
constexpr int kVStride = 288;  // bytes per key row of the V tile (72 dwords = 8 mod 64: the 8 key rows a 32-lane half reads with the transpose read hit 64 different banks)
constexpr float kNegBig = -1.0e30f;

struct PrepareArgs {
    const half_t *qkv;  // [m][ld_qkv]: per row the query heads, the key heads, the value heads
    int ld_qkv;
    half_t *qrot;       // [heads][m][hd]
    half_t *kc, *vc;    // [kv_heads][max_keys][hd]
    const half_t *cosv, *sinv;
    int heads, kv_heads, max_keys, pos, m;
};

// one thread per 16-byte piece of a (row, head slot): rotation as RotaryPosEmb_cuda_forward (hd = 128: the partner half is piece ^ 8)
__global__ __launch_bounds__(256) void attn_prefill_prepare_kernel(const PrepareArgs a) {
    const int slots = a.heads + 2 * a.kv_heads;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.m * slots * 16) return;
    const int piece = (int)(idx & 15);
    const int hs = (int)((idx >> 4) % slots), r = (int)((idx >> 4) / slots);
    const half_t *src = a.qkv + (size_t)r * a.ld_qkv + (size_t)hs * kHD;
    half8_t v = *reinterpret_cast<const half8_t *>(src + piece * 8);
    if (hs < a.heads + a.kv_heads && a.cosv) {
        const half8_t p = *reinterpret_cast<const half8_t *>(src + (piece ^ 8) * 8);
        const half8_t c = *reinterpret_cast<const half8_t *>(a.cosv + (size_t)(a.pos + r) * kHD + piece * 8);
        const half8_t s = *reinterpret_cast<const half8_t *>(a.sinv + (size_t)(a.pos + r) * kHD + piece * 8);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const half_t rot = piece < 8 ? (half_t)(-p[e]) : p[e];
            const half_t t = rot * s[e];             // hmul
            o[e] = __builtin_fmaf16(v[e], c[e], t);  // hfma, one rounding (the same statement as the decode step's and tce_rope_half's)
        }
        v = o;
    }
    half_t *dst;
    if (hs < a.heads) dst = a.qrot + ((size_t)hs * a.m + r) * kHD;
    else if (hs < a.heads + a.kv_heads) dst = a.kc + ((size_t)(hs - a.heads) * a.max_keys + a.pos + r) * kHD;
    else dst = a.vc + ((size_t)(hs - a.heads - a.kv_heads) * a.max_keys + a.pos + r) * kHD;
    *reinterpret_cast<half8_t *>(dst + piece * 8) = v;
}

struct PrefillArgs {
    const half_t *qrot;      // [heads][m][hd]
    const half_t *kc, *vc;   // [kv_heads][max_keys][hd]
    const half_t *mask;      // [m][ld_mask] additive, or null
    half_t *out;             // [m][ld_out]: row r, columns head * hd ..
    int ld_mask, ld_out;
    int heads, rep, max_keys, pos, m, causal;
    int pair;  // a workgroup takes two query blocks (see attn_prefill_kernel)
    float alpha;
};

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {  // over the 16 lanes of a DPP row, result in every lane of the row
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror
    v += dpp_f<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return v;
}

// NW: waves per workgroup; RT: 16-row MFMA tiles per wave (2: a K / V fragment feeds two MFMAs; kept for the sweep that ruled it out: > 200 registers,
// half the occupancy).  A block is 16 * RT * NW query rows: a staged K / V tile serves all of them.
//
// Everything is computed TRANSPOSED so that nothing but the operands' tiles touches LDS:
//   S^T = K Q^T    A = a K fragment (row = key, 8 head dimensions), B = the Q fragment (column = query row): the accumulator lane (n16, quad) holds
//                  query row n16, keys 16 j + 4 quad + r -- ONE query row per lane, so the online-softmax state is one (m, l) per lane and row tile,
//                  the row maximum is a register maximum over (j, r) plus two cross-quad exchanges, and the row sum stays a per-lane partial until the end
//                  (the rescale factor is the same in the four quads);
//   O^T += V^T P^T  on v_mfma_f32_16x16x16_f16 (contraction = 16 keys, 4 per lane): the B operand -- column = query row n16, keys 4 quad .. 4 quad + 3 -- IS
//                  the S^T accumulator layout of key tile j: the probabilities go from fp32 to packed fp16 in registers and never see LDS; A = a V^T
//                  fragment (row = head dimension, 4 keys: an 8-byte LDS read); the accumulator holds head dimensions 16 c + 4 quad + r of query row n16.
// The validity tests (keys past the context, the causal diagonal) run only on tiles that contain such keys.
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4_t lds_fp16x4_t;

__device__ __forceinline__ float quad_max(float v) {  // over the four lanes n16, n16 + 16, n16 + 32, n16 + 48
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <bool MASK, int NW, int RT>
__device__ __forceinline__ void attn_prefill_block(const PrefillArgs &a, const int qb, const int head, unsigned char *ks, unsigned char *vs) {
    constexpr int kBQ = 16 * RT * NW, NT = 64 * NW, KI = 1024 / NT;
    constexpr float kLog2e = 1.4426950408889634f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, quad = lane >> 4;
    const int kvh = head / a.rep;
    const int tgz = a.pos + a.m;
    const int r0 = qb * kBQ + wave * 16 * RT;
    // keys this block needs: all of them, or (causal) up to the block's last row's own position
    const int last_row = (qb * kBQ + kBQ < a.m ? qb * kBQ + kBQ : a.m) - 1;
    const int kend = a.causal ? a.pos + last_row + 1 : tgz;
    const int ntiles = (kend + kBK - 1) / kBK;
    const float scale2 = a.alpha * kLog2e;  // scores in units of log2: exp(x) = exp2(x * log2 e), one v_exp_f32 per probability

    // Q fragments of row tile t: row r0 + 16 t + n16 (clamped), head dimensions 32 s + 8 quad ..; that row is also the lane's softmax row
    half8_t qf[RT][4];
    int rowc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int row = r0 + 16 * t + n16;
        rowc[t] = row < a.m ? row : a.m - 1;
        const half_t *qrow = a.qrot + ((size_t)head * a.m + rowc[t]) * kHD;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[t][s] = *reinterpret_cast<const half8_t *>(qrow + 32 * s + 8 * quad);
    }
    float4_t o[RT][8];     // O^T: head dimensions 16 c + 4 quad + r of query row n16
    float m_i[RT], l_i[RT];  // the row's running maximum (log2 units; the same in the four quads) and this lane's share of its sum
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int c = 0; c < 8; ++c) o[t][c] = float4_t{0.f, 0.f, 0.f, 0.f};
        m_i[t] = kNegBig;
        l_i[t] = 0.f;
    }
    const half_t *kbase = a.kc + (size_t)kvh * a.max_keys * kHD, *vbase = a.vc + (size_t)kvh * a.max_keys * kHD;

    // a tile's global loads, K and V alike: piece idx = tid + NT i -> key idx / 16, piece idx % 16 (coalesced rows).  The loads of tile kt + 1 are requested
    // before tile kt is multiplied (registers) and written to LDS once every wave is done with tile kt.  (A ring of 3-4 register sets, tiles requested that
    // far ahead, measured SLOWER -- 151 -> 202 us at 2048 rows: the registers cost occupancy and the rows were not what the waves waited for.)
    half8_t kreg[KI], vreg[KI];
    auto fetch_tile = [&](int kt) {
        const int key0 = kt * kBK;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int idx = tid + NT * i, gk = key0 + (idx >> 4);
            kreg[i] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            vreg[i] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (gk < kend) {
                kreg[i] = *reinterpret_cast<const half8_t *>(kbase + (size_t)gk * kHD + (idx & 15) * 8);
                vreg[i] = *reinterpret_cast<const half8_t *>(vbase + (size_t)gk * kHD + (idx & 15) * 8);
            }
        }
    };
    fetch_tile(0);
    // this lane's piece of a [4 keys][16 head dimensions] block for the transpose read: key 4 quad + n16 / 4, head dimensions 4 (n16 % 4) ..
    const unsigned char *vfrag = vs + (4 * quad + (n16 >> 2)) * kVStride + (n16 & 3) * 8;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * kBK;
        __syncthreads();  // everybody has read the previous tiles
        // ---- both tiles as they lie ----
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int idx = tid + NT * i;
            *reinterpret_cast<half8_t *>(ks + (idx >> 4) * kKStride + (idx & 15) * 16) = kreg[i];
            *reinterpret_cast<half8_t *>(vs + (idx >> 4) * kVStride + (idx & 15) * 16) = vreg[i];
        }
        __syncthreads();
        if (kt + 1 < ntiles) fetch_tile(kt + 1);
        // ---- S^T = K Q^T (64 keys x this wave's 16 RT rows): a K fragment feeds the RT row tiles ----
        float4_t sacc[RT][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < RT; ++t) sacc[t][j] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const half8_t kf = *reinterpret_cast<const half8_t *>(ks + (16 * j + n16) * kKStride + (32 * s + 8 * quad) * 2);
#pragma unroll
                for (int t = 0; t < RT; ++t) sacc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[t][s], sacc[t][j], 0, 0, 0);
            }
        }
        // does this tile hold keys some row of the wave must not see?  (wave-uniform; the last tile, and the tiles on the wave's causal diagonal)
        const bool edge = key0 + kBK > tgz || (a.causal && key0 + kBK - 1 > a.pos + r0);
        // ---- scale, mask, cut; online softmax: the lane's row is rowc[t], its keys key0 + 16 j + 4 quad + r ----
        half4_t pb[RT][4];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            float best = kNegBig;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4_t mk4 = float4_t{0.f, 0.f, 0.f, 0.f};
                if constexpr (MASK) {
                    const int kb = key0 + 16 * j + 4 * quad;
                    const half_t *mrow = a.mask + (size_t)rowc[t] * a.ld_mask;
#pragma unroll
                    for (int r = 0; r < 4; ++r) mk4[r] = kb + r < tgz ? (float)mrow[kb + r] * kLog2e : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = sacc[t][j][r] * scale2;
                    if constexpr (MASK) s += mk4[r];
                    // check_inf_half (Int4llamaAttention.cu:105-115, applied by the decode step as well, attention_fast.hip): inf / NaN / a score beyond
                    // binary16 weighs nothing -- in EVERY tile, interior and unmasked ones included (a NaN from an overflowed q.k used to poison
                    // the row here and not in the decode step).  !(|s| <= bound) is true for NaN.
                    // (round 5, ADVICE r4: such a score becomes -65504 and still takes part in the softmax -- what the reference and the decode step do; a row of
                    //  nothing but such scores then weighs its keys equally in prefill as in decode -- while a key that is CUT, past the context or the causal
                    //  bound, weighs nothing)
                    const bool in_range = __builtin_fabsf(s) <= 65504.0f * kLog2e;
                    bool cut = false;
                    if (MASK || edge) {
                        const int key = key0 + 16 * j + 4 * quad + r;
                        cut = edge && !(key < tgz && !(a.causal && key > a.pos + rowc[t]));
                    }
                    s = cut ? kNegBig : (in_range ? s : -65504.0f * kLog2e);
                    sacc[t][j][r] = s;
                    best = fmaxf(best, s);
                }
            }
            const float m_new = fmaxf(m_i[t], quad_max(best));
            const float corr = __builtin_amdgcn_exp2f(m_i[t] - m_new);
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4_t p;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // a key that is cut weighs nothing even when the whole row so far is cut (m_new == kNegBig: exp2(0) would be 1)
                    p[r] = __builtin_amdgcn_exp2f(sacc[t][j][r] - m_new);
                    if (MASK || edge) p[r] = sacc[t][j][r] > kNegBig ? p[r] : 0.f;
                    rs += p[r];
                }
                pb[t][j] = half4_t{(half_t)p[0], (half_t)p[1], (half_t)p[2], (half_t)p[3]};
            }
            l_i[t] = l_i[t] * corr + rs;
            m_i[t] = m_new;
#pragma unroll
            for (int c = 0; c < 8; ++c) o[t][c] *= corr;
        }
        // ---- O^T += V^T P^T: 8 head-dimension tiles x 4 sub-tiles of 16 keys; a V^T fragment (4 keys: 8 bytes) feeds the RT row tiles ----
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // V[key 16 j + 4 quad + r][head dimension 16 c + n16], r = 0 .. 3: the 16 lanes of a quad read a [4 keys][16 dims] block of the row-major
                // tile, 8 bytes each, and ds_read_b64_tr_b16 hands lane n16 the block's column n16 (gfx950's LDS transpose read)
                const half4_t vf = __builtin_bit_cast(half4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (lds_fp16x4_t *)(vfrag + (16 * j) * kVStride + (16 * c) * 2)));
#pragma unroll
                for (int t = 0; t < RT; ++t) o[t][c] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pb[t][j], o[t][c], 0, 0, 0);
            }
    }
    // ---- out[row n16][head * hd + 16 c + 4 quad + r]: four consecutive halves per lane and column tile ----
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const float l = quad_sum(l_i[t]);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int row = r0 + 16 * t + n16;
        if (row >= a.m) continue;
        half_t *orow = a.out + (size_t)row * a.ld_out + (size_t)head * kHD;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const half4_t v = half4_t{(half_t)(o[t][c][0] * inv), (half_t)(o[t][c][1] * inv), (half_t)(o[t][c][2] * inv), (half_t)(o[t][c][3] * inv)};
            *reinterpret_cast<half4_t *>(orow + 16 * c + 4 * quad) = v;
        }
    }
}

// The launch: a workgroup takes query block nb - 1 - blockIdx.x (causal launches dispatch the blocks with the most key tiles first) and, when `pair` is set,
// block blockIdx.x after it: block b of a causal prompt walks b + 1 (x 2 for 128-row blocks) key tiles, so the pair (nb - 1 - i, i) is the same work for every
// i -- unpaired, all workgroups are resident at once and the launch lasts as long as its heaviest block while the CUs of the light ones idle.
template <bool MASK, int NW, int RT, bool PAIR>
__global__ __launch_bounds__(64 * NW) void attn_prefill_kernel(const PrefillArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char ks[kBK * kKStride];  // K tile
    __shared__ __attribute__((aligned(16))) unsigned char vs[kBK * kVStride];  // V tile
    constexpr int kBQ = 16 * RT * NW;
    const int nb = (a.m + kBQ - 1) / kBQ;
    const int first = nb - 1 - (int)blockIdx.x;
    if constexpr (PAIR) {
        // (two inlined copies, not a loop around one: 35 registers more -- the paired launch is one 8-wave workgroup per CU anyway -- and 107 against 118 us
        //  at 2048 rows; the unpaired kernel is its own instantiation because those registers would cost it its second workgroup per CU)
        attn_prefill_block<MASK, NW, RT>(a, first, blockIdx.y, ks, vs);
        if ((int)blockIdx.x < first) attn_prefill_block<MASK, NW, RT>(a, blockIdx.x, blockIdx.y, ks, vs);
    } else {
        attn_prefill_block<MASK, NW, RT>(a, first, blockIdx.y, ks, vs);
    }
}

thread_local int g_prefill_pair = 0;   // 0: by the rule; 1 / 2: pairing forced on / off
thread_local int g_prefill_waves = 0;  // 0: by the rule in launch_attention_prefill; forced (tests, sweeps): 4 / 8 waves with one row tile per wave, 14 / 18: with two

}  // namespace

void set_attention_prefill_waves(int w) {
    g_prefill_pair = w / 100;  // + 100: pairs forced on, + 200: off
    w %= 100;
    g_prefill_waves = (w == 4 || w == 8 || w == 14 || w == 18) ? w : 0;
}

size_t attention_prefill_workspace_bytes(int heads, int m, int hd) {
    if (hd != kHD || heads <= 0 || m <= 0) return 0;
    return (size_t)heads * m * kHD * sizeof(half_t);
}

int launch_attention_prefill(const void *qkv, int ld_qkv, void *kc, void *vc, const void *cosv, const void *sinv, const void *mask, int ld_mask, int causal,
                             void *out, int ld_out, void *workspace, int heads, int kv_heads, int max_keys, int pos, int m, float alpha, hipStream_t stream,
                             hipError_t *hip_err) {
    PrepareArgs p{};
    p.qkv = static_cast<const half_t *>(qkv);
    p.ld_qkv = ld_qkv;
    p.qrot = static_cast<half_t *>(workspace);
    p.kc = static_cast<half_t *>(kc);
    p.vc = static_cast<half_t *>(vc);
    p.cosv = static_cast<const half_t *>(cosv);
    p.sinv = static_cast<const half_t *>(sinv);
    p.heads = heads;
    p.kv_heads = kv_heads;
    p.max_keys = max_keys;
    p.pos = pos;
    p.m = m;
    const long long pieces = (long long)m * (heads + 2 * kv_heads) * 16;
    hipLaunchKernelGGL(attn_prefill_prepare_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, p);
    PrefillArgs a{};
    a.qrot = p.qrot;
    a.kc = p.kc;
    a.vc = p.vc;
    a.mask = static_cast<const half_t *>(mask);
    a.out = static_cast<half_t *>(out);
    a.ld_mask = ld_mask;
    a.ld_out = ld_out;
    a.heads = heads;
    a.rep = heads / kv_heads;
    a.max_keys = max_keys;
    a.pos = pos;
    a.m = m;
    a.causal = causal;
    a.alpha = alpha;
    // the block: 128 rows (8 waves) while that leaves two workgroups per CU, else 64 rows (forced: g_prefill_waves)
    auto blocks = [&](int rows) { return (long long)((m + rows - 1) / rows) * heads; };
    int form = g_prefill_waves;
    if (form == 0) form = blocks(128) >= 512 ? 8 : 4;
    // causal prompts: pair a heavy block with a light one while the pairs still fill the chip
    a.pair = g_prefill_pair == 1 || (g_prefill_pair == 0 && causal && blocks(form == 8 || form == 18 ? (form == 18 ? 256 : 128) : (form == 14 ? 128 : 64)) >= 512) ? 1 : 0;  // (two row tiles per wave -- forms 14 / 18 -- cost more in occupancy than the shared fragments give: 147 vs 130 us at 2048 rows)
    auto go = [&](auto nw_c, auto rt_c) {
        constexpr int NW = decltype(nw_c)::value, RT = decltype(rt_c)::value;
        const int nb = (m + 16 * RT * NW - 1) / (16 * RT * NW);
        const dim3 grid(a.pair ? (nb + 1) / 2 : nb, heads);
        if (a.pair) {
            if (mask) hipLaunchKernelGGL((attn_prefill_kernel<true, NW, RT, true>), grid, dim3(64 * NW), 0, stream, a);
            else hipLaunchKernelGGL((attn_prefill_kernel<false, NW, RT, true>), grid, dim3(64 * NW), 0, stream, a);
        } else {
            if (mask) hipLaunchKernelGGL((attn_prefill_kernel<true, NW, RT, false>), grid, dim3(64 * NW), 0, stream, a);
            else hipLaunchKernelGGL((attn_prefill_kernel<false, NW, RT, false>), grid, dim3(64 * NW), 0, stream, a);
        }
    };
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    if (form == 18) go(I8{}, I2{});
    else if (form == 14) go(I4{}, I2{});
    else if (form == 8) go(I8{}, I1{});
    else go(I4{}, I1{});
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
