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
//   * scale, additive mask, causal cut, online softmax on the accumulator layout (a lane holds 4 rows x 4 key columns; row maxima and sums over the
//     16 lanes of a DPP row);
//   * P goes through a per-wave LDS slab into A-fragment order (fp16); the V tile is TRANSPOSED while it is staged (V^T [128][64 keys], row stride
//     144 bytes), because the P V product contracts over keys and an MFMA operand wants its contraction index contiguous: a lane loads 16 bytes
//     of one key's row and scatters them as eight 2-byte LDS writes (lane = key: neighbouring lanes write neighbouring halves);
//   * O += P V: 8 column tiles x 2 k-steps of 32 keys.
// Grouped-query attention: query head i reads key / value head i / (heads / kv_heads); the workgroups of a group stage the same tiles (L2).
// Rows of the caches at and beyond pos + m may hold anything (uninitialised memory): staged as zeros, and their scores are cut.
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

namespace {

constexpr int kHD = 128;
constexpr int kBK = 64;
constexpr int kKStride = 272;  // bytes per key row of the K tile in LDS
constexpr int kVStride = 144;  // bytes per head-dimension row of the V^T tile
constexpr int kPStride = 144;  // bytes per query row of a wave's P slab
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

// NW: waves per workgroup = 16 query rows each (4: blocks of 64 rows; 8: blocks of 128 rows -- a staged tile serves twice the rows; taken when the
// launch still has two workgroups per CU's worth of blocks)
template <bool MASK, int NW>
__global__ __launch_bounds__(64 * NW) void attn_prefill_kernel(const PrefillArgs a) {
    constexpr int kBQ = 16 * NW, NT = 64 * NW, KI = 1024 / NT, VI = 16 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char ks[kBK * kKStride];      // K tile
    __shared__ __attribute__((aligned(16))) unsigned char vt[kHD * kVStride];      // V^T tile
    __shared__ __attribute__((aligned(16))) unsigned char ps[NW][16 * kPStride];   // a P slab per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, quad = lane >> 4;
    // (causal: the blocks with the most key tiles are dispatched first)
    const int qb = (int)gridDim.x - 1 - (int)blockIdx.x, head = blockIdx.y, kvh = head / a.rep;
    const int tgz = a.pos + a.m;
    const int r0 = qb * kBQ + wave * 16;
    // keys this block needs: all of them, or (causal) up to the block's last row's own position
    const int last_row = (qb * kBQ + kBQ < a.m ? qb * kBQ + kBQ : a.m) - 1;
    const int kend = a.causal ? a.pos + last_row + 1 : tgz;
    const int ntiles = (kend + kBK - 1) / kBK;

    // Q fragments: row r0 + n16 (clamped), head dimensions 32 s + 8 quad ..
    half8_t qf[4];
    {
        int row = r0 + n16;
        row = row < a.m ? row : a.m - 1;
        const half_t *qrow = a.qrot + ((size_t)head * a.m + row) * kHD;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const half8_t *>(qrow + 32 * s + 8 * quad);
    }
    float4_t o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = float4_t{0.f, 0.f, 0.f, 0.f};
    float m_i[4], l_i[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m_i[r] = kNegBig;
        l_i[r] = 0.f;
    }
    const half_t *kbase = a.kc + (size_t)kvh * a.max_keys * kHD, *vbase = a.vc + (size_t)kvh * a.max_keys * kHD;
    unsigned char *pw = ps[wave];

    // a tile's global loads: K piece idx = tid + 256 i -> key idx / 16, piece idx % 16 (coalesced rows); V: lane = key, piece = wave + 4 i.  The loads of
    // tile kt + 1 are requested before tile kt is multiplied (registers), and written to LDS once every wave is done with tile kt.
    half8_t kreg[KI], vreg[VI];
    auto fetch_tile = [&](int kt) {
        const int key0 = kt * kBK;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int idx = tid + NT * i, gk = key0 + (idx >> 4);
            kreg[i] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (gk < kend) kreg[i] = *reinterpret_cast<const half8_t *>(kbase + (size_t)gk * kHD + (idx & 15) * 8);
        }
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int gk = key0 + lane;
            vreg[i] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            if (gk < kend) vreg[i] = *reinterpret_cast<const half8_t *>(vbase + (size_t)gk * kHD + (wave + NW * i) * 8);
        }
    };
    fetch_tile(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * kBK;
        __syncthreads();  // everybody has read the previous tiles
        // ---- K as it lies; V transposed: eight 2-byte writes per piece (neighbouring lanes: neighbouring halves of a row of V^T) ----
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int idx = tid + NT * i;
            *reinterpret_cast<half8_t *>(ks + (idx >> 4) * kKStride + (idx & 15) * 16) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int piece = wave + NW * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) *reinterpret_cast<half_t *>(vt + (piece * 8 + e) * kVStride + lane * 2) = vreg[i][e];
        }
        __syncthreads();
        if (kt + 1 < ntiles) fetch_tile(kt + 1);
        // ---- S = Q K^T (this wave's 16 rows x 64 keys) ----
        float4_t sacc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sacc[j] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const half8_t kf = *reinterpret_cast<const half8_t *>(ks + (16 * j + n16) * kKStride + (32 * s + 8 * quad) * 2);
                sacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[s], kf, sacc[j], 0, 0, 0);
            }
        }
        // ---- scale, mask, cut; online softmax.  The lane holds rows 4 quad + r, key columns 16 j + n16 ----
        float mx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + 4 * quad + r;
            const int rowc = row < a.m ? row : a.m - 1;
            float best = kNegBig;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = key0 + 16 * j + n16;
                float s = sacc[j][r] * a.alpha;
                const bool valid = key < tgz && !(a.causal && key > a.pos + rowc);
                if constexpr (MASK) {
                    if (valid) s += (float)a.mask[(size_t)rowc * a.ld_mask + key];
                }
                s = valid ? s : kNegBig;
                s = s > kNegBig ? s : kNegBig;  // a mask of -inf / -65504 sums stays a finite "nothing"
                sacc[j][r] = s;
                best = fmaxf(best, s);
            }
            mx[r] = row16_max(best);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m_new = fmaxf(m_i[r], mx[r]);
            const float corr = __expf(m_i[r] - m_new);
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // a key that is cut weighs nothing even when the whole row so far is cut (m_new == kNegBig: exp(0) would be 1)
                const float p = sacc[j][r] > kNegBig ? __expf(sacc[j][r] - m_new) : 0.f;
                rs += p;
                *reinterpret_cast<half_t *>(pw + (4 * quad + r) * kPStride + (16 * j + n16) * 2) = (half_t)p;
            }
            l_i[r] = l_i[r] * corr + row16_sum(rs);
            m_i[r] = m_new;
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c][r] *= corr;
        }
        // ---- O += P V: P back in A-fragment order (the slab is this wave's own: LDS operations of a wave complete in order; the empty asm
        //      statements keep the compiler from moving the 16-byte reads across the 2-byte writes of another type) ----
        asm volatile("" ::: "memory");
        half8_t pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) pf[s2] = *reinterpret_cast<const half8_t *>(pw + n16 * kPStride + (32 * s2 + 8 * quad) * 2);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const half8_t vf = *reinterpret_cast<const half8_t *>(vt + (16 * c + n16) * kVStride + (32 * s2 + 8 * quad) * 2);
                o[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf[s2], vf, o[c], 0, 0, 0);
            }
        asm volatile("" ::: "memory");
    }
    // ---- out[row][head * hd + 16 c + n16] ----
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + 4 * quad + r;
        if (row >= a.m) continue;
        const float inv = l_i[r] > 0.f ? 1.0f / l_i[r] : 0.f;
        half_t *orow = a.out + (size_t)row * a.ld_out + (size_t)head * kHD;
#pragma unroll
        for (int c = 0; c < 8; ++c) orow[16 * c + n16] = (half_t)(o[c][r] * inv);
    }
}

int g_prefill_waves = 0;  // 0: by the rule in launch_attention_prefill; 4 / 8 forced (tests, sweeps)

}  // namespace

void set_attention_prefill_waves(int w) { g_prefill_waves = (w == 4 || w == 8) ? w : 0; }

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
    const bool wide = g_prefill_waves == 8 || (g_prefill_waves == 0 && (long long)((m + 127) / 128) * heads >= 512);
    if (wide) {
        const dim3 grid((m + 127) / 128, heads);
        if (mask) hipLaunchKernelGGL((attn_prefill_kernel<true, 8>), grid, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((attn_prefill_kernel<false, 8>), grid, dim3(512), 0, stream, a);
    } else {
        const dim3 grid((m + 63) / 64, heads);
        if (mask) hipLaunchKernelGGL((attn_prefill_kernel<true, 4>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((attn_prefill_kernel<false, 4>), grid, dim3(256), 0, stream, a);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
