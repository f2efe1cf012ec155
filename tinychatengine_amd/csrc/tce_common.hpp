// tce_common.hpp -- shared device helpers for the gfx950 kernels (wave64, CDNA4 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "tce_matmul.h"
#include "tce_tuning.h"

namespace tce {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;  // CDNA wavefront

// llm/src/nn_modules/cuda/utils.cu:162-178 (calculate_zeros_width), host + device
__host__ __device__ inline int zeros_width(int in_features, int group_size) {
    const int mult = group_size >= 128 ? 1 : (group_size == 64 ? 2 : 4);
    int w = (in_features / group_size + 7) / 8;
    return ((w + mult - 1) / mult) * mult;
}

__device__ __forceinline__ half2_t as_half2(unsigned v) { return __builtin_bit_cast(half2_t, v); }
__device__ __forceinline__ unsigned as_u32(half2_t v) { return __builtin_bit_cast(unsigned, v); }

// ---------------------------------------------------------------------------------------------
// int4 -> fp16 without integer->float conversions.
//
// A q4_6 word holds codes q0..q7 for k = 8j..8j+7, q_i in bits [4i, 4i+4).  OR-ing a nibble into the
// mantissa of 1024.0h (0x6400, ulp 1) yields the half 1024+q; a nibble at mantissa bits 4..7 yields
// 1024+16q.  Two nibbles 16 bits apart are converted at once:
//     (w      & 0x000F000F) | 0x64006400 -> (1024+q0,    1024+q4)
//     (w      & 0x00F000F0) | 0x64006400 -> (1024+16q1,  1024+16q5)
//     (w >> 8 & 0x000F000F) | 0x64006400 -> (1024+q2,    1024+q6)
//     (w >> 8 & 0x00F000F0) | 0x64006400 -> (1024+16q3,  1024+16q7)
// and the zero point is removed EXACTLY in fp16:  (1024+q) + (-(1024+z)) = q - z;
// (1024+16q) * (1/16) + (-(64+z)) = q - z (one fma, every intermediate representable).
// So d[0..3] = (q0-z,q4-z), (q1-z,q5-z), (q2-z,q6-z), (q3-z,q7-z): the activation vector is staged
// in the matching pair order (x0,x4),(x1,x5),(x2,x6),(x3,x7) so no lane ever shuffles weights.
// ---------------------------------------------------------------------------------------------
struct ZeroPair {
    half2_t lo;  // (-(1024+z), -(1024+z))
    half2_t hi;  // (-(64+z),   -(64+z))
};

__device__ __forceinline__ ZeroPair make_zero_pair(unsigned z /*0..15*/) {
    ZeroPair zp;
    zp.lo = as_half2(0xE400E400u | (z * 0x00010001u));  // -(1024+z): 0xE400 | z
    zp.hi = as_half2(0xD400D400u | (z * 0x00100010u));  // -(64+z):   0xD400 | z<<4  (ulp of [64,128) is 1/16)
    return zp;
}

// The two nibble masks live in VGPRs: gfx9 VOP3 instructions may read only ONE scalar/literal operand, so with both
// the mask and the magic constant as literals hipcc splits every (w & mask) | magic into v_and + v_or; with the mask
// in a VGPR it is a single v_and_or_b32 (mask VGPR, magic SGPR).  The empty asm makes the values opaque to the
// constant folder.
struct NibbleMasks {
    unsigned lo, hi;
};
__device__ __forceinline__ NibbleMasks make_nibble_masks() {
    NibbleMasks m;
    asm volatile("v_mov_b32 %0, 0x000F000F\n\tv_mov_b32 %1, 0x00F000F0" : "=v"(m.lo), "=v"(m.hi));
    return m;
}

__device__ __forceinline__ void dequant_word(unsigned w, const ZeroPair &zp, const NibbleMasks &nm, half2_t (&d)[4]) {
    const half2_t sixteenth = as_half2(0x2C002C00u);  // 0.0625h
    const unsigned w8 = w >> 8;
    const half2_t t0 = as_half2((w & nm.lo) | 0x64006400u);
    const half2_t t1 = as_half2((w & nm.hi) | 0x64006400u);
    const half2_t t2 = as_half2((w8 & nm.lo) | 0x64006400u);
    const half2_t t3 = as_half2((w8 & nm.hi) | 0x64006400u);
    d[0] = t0 + zp.lo;
    d[1] = __builtin_elementwise_fma(t1, sixteenth, zp.hi);
    d[2] = t2 + zp.lo;
    d[3] = __builtin_elementwise_fma(t3, sixteenth, zp.hi);
}

// Reorder 8 consecutive halves (x0..x7, one 16-byte piece) into the pair order the dequantizer produces:
// (x0,x4),(x1,x5),(x2,x6),(x3,x7).
__device__ __forceinline__ uint4_t pair_permute(uint4_t v) {
    // v.x = (x0,x1) v.y = (x2,x3) v.z = (x4,x5) v.w = (x6,x7), low half first (hipcc folds these to v_perm_b32 / v_and_or_b32)
    uint4_t r;
    r.x = (v.x & 0x0000FFFFu) | (v.z << 16);         // (x0, x4)
    r.y = (v.x >> 16) | (v.z & 0xFFFF0000u);         // (x1, x5)
    r.z = (v.y & 0x0000FFFFu) | (v.w << 16);         // (x2, x6)
    r.w = (v.y >> 16) | (v.w & 0xFFFF0000u);         // (x3, x7)
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// 64-lane sum on the VALU with DPP cross-lane operands: six dependent adds of ~8 cycles each, against six ds_bpermute
// round trips through the LDS crossbar (~100+ cycles each) for the shuffle tree above.  The total ends up in LANE 63
// only (rows 1..3 accumulate the rows before them).
__device__ __forceinline__ float wave_sum_dpp_lane63(float v) {
    auto dpp = [](float x, auto ctrl, auto row_mask) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                     decltype(row_mask)::value, 0xF, false));
    };
    using I = std::integral_constant<int, 0>;
    (void)sizeof(I);
    v += dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});   // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});  // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});  // row_mirror: every lane of a row = row sum
    v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});  // row_bcast15 into rows 1 and 3
    v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});  // row_bcast31 into rows 2 and 3
    return v;
}

// SiLuMul_half (llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu:20-30): v * (1 / (1 + hexp(-v))) * u with every operation
// rounded to binary16 (hexp = the float exponential rounded to nearest half).
__device__ __forceinline__ half_t silu_mul_half(half_t g, half_t u) {
    const half_t one = (half_t)1.0f;
    const half_t e = (half_t)expf((float)(-g));  // see oracle/tce_oracle.c orc_silu_mul_half for the 1-ulp caveat
    const half_t d = one + e;
    const half_t r = one / d;
    const half_t sv = g * r;
    return sv * u;
}

// One output of generalT5LayerNorm (llm/src/ops/cuda/LlamaRMSNorm.cu:89-92): half(clamp((x * rs) * gamma)), the clamp of
// clamp_inf_for_half (reduction.cuh:76-81).
__device__ __forceinline__ half_t rmsnorm_out(half_t x, float rs, float gamma) {
    float f = ((float)x * rs) * gamma;
    f = f > 0.0f ? fminf(f, 65504.f - 1000.f) : fmaxf(f, -65504.f + 1000.f);
    return (half_t)f;
}

// rs = 1 / sqrt(mean(x^2) + eps) of one fp16 row of n (n % 8 == 0) elements, formed by a whole workgroup in an order that
// does not depend on the workgroup's shape, so the fused GEMV prologues and the stand-alone kernel produce identical bits.
// THE ORDER (round 2; for n <= 8192 the same bits as round 1's):
//   s_p     of the row's p-th 16-byte piece: s = 0, then s = fmaf(v_e, v_e, s) for its 8 values in order;
//   slot[q] for q = 0 .. 1023: s_q + s_{q + 1024} + s_{q + 2048} + ... in that order (0 where the row has no such piece);
//   t_l     for lane l = 0 .. 63: slot[l] + slot[64 + l] + ... + slot[960 + l] in that order;
//   tot     = the fixed DPP tree over the 64 t_l (wave_sum_dpp_lane63);  rs = 1 / sqrtf(tot / n + eps).
// A piece's s_p needs nothing but the piece, so a kernel that already holds the row in registers (w4a16_gemv.hip) forms the slots
// from there -- rmsnorm_slots_finish() below is the part from `slot[]` on; rmsnorm_rs_block() is the whole thing for callers that
// read the row through `load_piece(p)` (the row's p-th piece as 8 halves): wave w fills slots 64 c + lane for c = w, w + nwaves, ...
// `part` = 4 KiB of LDS holding slot[]; both contain two barriers (the second one frees `part` for reuse).
// A workgroup barrier that publishes LDS writes and leaves global loads IN FLIGHT: __syncthreads() carries a fence the compiler implements as vmcnt(0) -- every
// barrier of a kernel that requested its weights ahead would wait for the last of them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float rmsnorm_piece_sum(half8_t v) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = __builtin_fmaf((float)v[e], (float)v[e], s);
    return s;
}
__device__ __forceinline__ float rmsnorm_slots_finish(const float *part, int n, float eps, int lane) {
    lds_barrier();  // (LDS only: the fused GEMV has its weight loads in flight here)
    float tot = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) tot += part[c * 64 + lane];
    tot = wave_sum_dpp_lane63(tot);
    tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 63));
    lds_barrier();
    return 1.0f / sqrtf(tot / (float)n + eps);
}
template <typename LoadPiece>
__device__ __forceinline__ float rmsnorm_rs_block(LoadPiece load_piece, int n, float eps, int wave, int nwaves, int lane, float *part) {
    const int pieces = n >> 3;
    for (int c = wave; c < 16; c += nwaves) {
        float slot = 0.f;
        for (int p = c * 64 + lane; p < pieces; p += 1024) slot += rmsnorm_piece_sum(load_piece(p));
        part[c * 64 + lane] = slot;
    }
    return rmsnorm_slots_finish(part, n, eps, lane);
}
__device__ __forceinline__ float rmsnorm_rs_block(const half_t *x, int n, float eps, int wave, int nwaves, int lane, float *part) {
    return rmsnorm_rs_block([&](int p) { return *reinterpret_cast<const half8_t *>(x + p * 8); }, n, eps, wave, nwaves, lane, part);
}

// Sequential fp32 sum  acc = (((x[0] + x[1]) + x[2]) + ...)  over n values that sit in LDS, in exactly that order --
// what a scalar host loop `for (k) acc += x[k]` computes (LayerNormQ.cc:27-30) -- at about one dependent v_add per element instead
// of one LDS round trip per few elements: lanes 0..15 of the wave each fetch one value of a 16-value group (a pipelined
// ds_read_b32, off the critical path) and lane 0 adds them in order through DPP row shifts (row_shl:j = "read lane i + j"), so
// the only dependence chain is the accumulator's.  `f` maps a value before it is added (identity, or the squared deviation,
// computed by the lane that fetched it).  The total is valid in lane 0 (and broadcast by the caller).  All 64 lanes must call.
template <typename F>
__device__ __forceinline__ float sequential_sum_lane0(const float *lds_x, int n, int lane, F &&f) {
    float acc = 0.f;
    const int l16 = lane & 15;
    const int n16 = n & ~15;
    float nxt = n16 ? f(lds_x[l16]) : 0.f;
    for (int base = 0; base < n16; base += 16) {
        const float cur = nxt;
        const int nb = base + 16 < n16 ? base + 16 : base;
        nxt = f(lds_x[nb + l16]);  // next group's value: in flight while this group is added
        // One statement, sixteen dependent adds: acc + cur(lane 0), + cur(lane 1), ...  Written as asm because hipcc pads every
        // DPP instruction that reads a just-written register with s_nop 1 -- the hazard (VALU write -> DPP read: 2 wait states)
        // concerns the DPP-shifted operand (src0 = cur, written a whole group ago; the s_nop 1 in front covers a late `f`), not
        // the plain second operand acc -- which doubles the length of the only dependence chain of this kernel.
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32 %0, %1, %0\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:9 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:10 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:11 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:12 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:13 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:14 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_add_f32_dpp %0, %1, %0 row_shl:15 row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "+v"(acc)
            : "v"(cur));
    }
    for (int k = n16; k < n; ++k) acc = acc + f(lds_x[k]);  // n % 16 values at the end: plain (every lane, same values)
    return acc;
}

// The same sequential sum with the values handed to EVERY lane by LDS broadcast reads (all lanes read the same 16-byte pieces:
// conflict-free, one LDS cycle per instruction) instead of to lane 0 by DPP shifts: the chain's step is then a plain
// `v_add_f32 acc, v[i], acc` -- 6.0 shader cycles per dependent addition against 11.3 for the DPP form
// (scripts/probes/dep_add_probe.hip: 2.86 against 4.68 ns per element with the reads included).  Two 16-value buffers in FIXED registers
// v96..v127, one being refilled while the other is added; the loop is ONE asm statement, so the compiler never sees a register whose data
// has not landed (and a kernel that calls this reports >= 128 VGPRs).  `vals` is 16-byte aligned; the loop reads up to 128 bytes past the
// last 32-value pair (LDS reads past the allocation return zero; nothing read there is added).  The total is valid in EVERY lane.  A
// mapped sum (the squared deviations of LayerNormQ.cc:33-36) is taken over values the wave computed into LDS beforehand, in parallel.
__device__ __forceinline__ float sequential_sum_bcast(const float *vals, int n, float init = 0.f) {  // init: the running value the additions start from (may differ per lane)
    float acc = init;
    int pairs = n >> 5;
    if (pairs > 0) {
        unsigned p = (unsigned)(size_t)vals;  // LDS byte address
        asm volatile(
            "ds_read_b128 v[96:99], %1\n\tds_read_b128 v[100:103], %1 offset:16\n\tds_read_b128 v[104:107], %1 offset:32\n\tds_read_b128 v[108:111], %1 offset:48\n\t"
            "ds_read_b128 v[112:115], %1 offset:64\n\tds_read_b128 v[116:119], %1 offset:80\n\tds_read_b128 v[120:123], %1 offset:96\n\tds_read_b128 v[124:127], %1 offset:112\n"
            "1:\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_add_f32 %0, v96, %0\n\tv_add_f32 %0, v97, %0\n\tv_add_f32 %0, v98, %0\n\tv_add_f32 %0, v99, %0\n\t"
            "v_add_f32 %0, v100, %0\n\tv_add_f32 %0, v101, %0\n\tv_add_f32 %0, v102, %0\n\tv_add_f32 %0, v103, %0\n\t"
            "v_add_f32 %0, v104, %0\n\tv_add_f32 %0, v105, %0\n\tv_add_f32 %0, v106, %0\n\tv_add_f32 %0, v107, %0\n\t"
            "v_add_f32 %0, v108, %0\n\tv_add_f32 %0, v109, %0\n\tv_add_f32 %0, v110, %0\n\tv_add_f32 %0, v111, %0\n\t"
            "ds_read_b128 v[96:99], %1 offset:128\n\tds_read_b128 v[100:103], %1 offset:144\n\tds_read_b128 v[104:107], %1 offset:160\n\tds_read_b128 v[108:111], %1 offset:176\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_add_f32 %0, v112, %0\n\tv_add_f32 %0, v113, %0\n\tv_add_f32 %0, v114, %0\n\tv_add_f32 %0, v115, %0\n\t"
            "v_add_f32 %0, v116, %0\n\tv_add_f32 %0, v117, %0\n\tv_add_f32 %0, v118, %0\n\tv_add_f32 %0, v119, %0\n\t"
            "v_add_f32 %0, v120, %0\n\tv_add_f32 %0, v121, %0\n\tv_add_f32 %0, v122, %0\n\tv_add_f32 %0, v123, %0\n\t"
            "v_add_f32 %0, v124, %0\n\tv_add_f32 %0, v125, %0\n\tv_add_f32 %0, v126, %0\n\tv_add_f32 %0, v127, %0\n\t"
            "ds_read_b128 v[112:115], %1 offset:192\n\tds_read_b128 v[116:119], %1 offset:208\n\tds_read_b128 v[120:123], %1 offset:224\n\tds_read_b128 v[124:127], %1 offset:240\n\t"
            "v_add_u32 %1, 0x80, %1\n\t"
            "s_sub_u32 %2, %2, 1\n\t"
            "s_cmp_lg_u32 %2, 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "+v"(acc), "+v"(p), "+s"(pairs)
            :
            : "memory", "scc", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",
              "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    }
    // the last n % 32 values: two 16-value groups read at once (past the end: unused), added under wave-uniform conditions
    const int done = n & ~31, rest = n - done;
    if (rest > 0) {
        float t[32];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4_t v = *reinterpret_cast<const float4_t *>(vals + done + 4 * j);
            t[4 * j] = v.x; t[4 * j + 1] = v.y; t[4 * j + 2] = v.z; t[4 * j + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 31; ++j)
            if (j < rest) acc = acc + t[j];
    }
    return acc;
}

// Which form: the broadcast form is the shorter CHAIN (a launch of a few rows: decode), but its four ds_read_b128 per 16 values occupy the CU's LDS
// for 16 cycles per wave and 16 values -- with four or more waves per SIMD walking rows the LDS, not the adder, is then the limit, and the DPP form
// (one ds_read_b32 per 16 values) is the faster one (OPT-125M prefill softmax, 6144 rows: 26.4 us against 31.6).  BCAST is chosen per launch.
template <bool BCAST>
__device__ __forceinline__ float sequential_sum(const float *vals, int n, int lane) {
    if constexpr (BCAST) {
        return sequential_sum_bcast(vals, n);
    } else {
        const float s = sequential_sum_lane0(vals, n, lane, [](float v) { return v; });
        return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s)));
    }
}

// The form for MANY waves of a CU walking sums at once (sequential_sum_speculated): the four ds_read_b128 per 16 values of the form above occupy the LDS for
// ~32 cycles per wave -- with 16 waves the LDS, not the adders, sets the pace.  Here a lane of every 16-lane row reads ONE value (a single ds_read_b32 brings 16
// values, the same 16 in each row) and each addition takes its operand through the DPP row broadcast (row_newbcast:j = lane j of the row to all of its lanes): one
// VALU instruction per addition and one LDS instruction per 16, so 4 waves per SIMD advance at one addition per 16 cycles each: n cycles per sum and CU.
// Same contract as sequential_sum_bcast (the tail of n % 32 values is added from plain broadcast reads); reads up to 256 bytes past the last 32 values.
__device__ __forceinline__ float sequential_sum_rowbcast(const float *vals, int n, float init, int lane) {
    float acc = init;
    int groups = n >> 5;
    if (groups > 0) {
        unsigned p = (unsigned)(size_t)vals + ((unsigned)(lane & 15) << 2);
        float va, vb;
        asm volatile(
            "ds_read_b32 %3, %1\n\tds_read_b32 %4, %1 offset:64\n"
            "1:\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %3, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
            "ds_read_b32 %3, %1 offset:128\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %0, %4, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
            "ds_read_b32 %4, %1 offset:192\n\t"
            "v_add_u32 %1, 0x80, %1\n\t"
            "s_sub_u32 %2, %2, 1\n\t"
            "s_cmp_lg_u32 %2, 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "+v"(acc), "+v"(p), "+s"(groups), "=&v"(va), "=&v"(vb)
            :
            : "memory", "scc");
    }
    const int done = n & ~31;
    for (int j = done; j < n; ++j) acc = acc + vals[j];  // (every lane reads the same address; the additions stay in order)
    return acc;
}

// The same sum by ALL NW waves of a workgroup, for ONE long row (LayerNormQ at OPT-1.3B / 6.7B widths: 2 x 4096 dependent additions = 22 us for a single wave).
// An fp32 sum rounded after every addition cannot be re-associated -- but it can be SPECULATED: the row is cut into NW segments; wave w takes segment w and starts
// it from 64 different running values at once, one per lane: the fp32 neighbours (-32 .. +31 units in the last place) of the exactly rounded sum of everything
// before the segment, which is known at once from a double-precision reduction.  (A wave's 64 lanes are idle during a dependent chain anyway: every lane adds the
// same broadcast value to ITS running value.)  Then wave 0 walks the segments in order: the true running value at the start of segment w is the end value of
// segment w - 1; if one of the 64 candidates of segment w is that very number (bit pattern), the lane's end value IS the sequential sum over the segment, else
// (the running value came closer to zero than the rounding errors accumulated so far, signed zeros, inf / NaN) the segment is re-added from the true value.
// Every addition that contributes to the result is the reference's, in the reference's order, on the reference's operands: bit-identical by construction, the
// speculation only decides how much of it was done ahead.  Cost: n / NW dependent additions per wave (with 16 waves the CU's four SIMDs are the limit: n cycles)
// + one segment per miss (zero-mean noise, 4096 values: ~11 % of the boundaries of the plain sum, ~0.4 % of the sum of squares).  All threads of the workgroup
// must call it (three barriers, which leave global loads in flight); `sp` is kSpecScratchFloats(NW) floats of LDS scratch.
// the segment of wave `wave`: [b, b + len), a multiple of 32 long (the chain loops' unit, and 128-byte alignment) except the last one
template <int NW>
__device__ __forceinline__ void speculated_segment(int n, int wave, int &b, int &len) {
    const int L = ((n + NW * 32 - 1) / (NW * 32)) * 32;
    b = wave * L;
    len = n - b < 0 ? 0 : (n - b < L ? n - b : L);
}

// `ds_mine`: the double-precision sum of the wave's own segment, valid in every lane (the caller forms it -- while it computes the values, if they are computed);
// the chain of a wave reads only its own segment (and up to 256 bytes behind it, unused), so values a wave wrote itself need no barrier before the call.
struct SpecNoOp {
    __device__ __forceinline__ void operator()() const {}
};
// `mid` is called by every wave between its chain and the barrier behind it (a place to issue memory requests that should not all be in flight at once)
// MORE_WAVES: the workgroup has waves beyond the NW that walk the sum; they only keep the barriers' count
template <int NW, bool ROWB = true, class Mid = SpecNoOp, bool MORE_WAVES = false>
__device__ __forceinline__ float sequential_sum_speculated(const float *vals, int n, float *sp, int wave, int lane, double ds_mine, unsigned long long *stamps = nullptr,
                                                            Mid mid = Mid()) {
    static_assert(NW <= 16, "segment sums are exchanged through 16 doubles");
    double *segsum = reinterpret_cast<double *>(sp);  // [16]
    float *cand = sp + 32, *fin = sp + 32 + NW * 64, *res = sp + 32 + 2 * NW * 64;
    int b, len;
    speculated_segment<NW>(n, wave, b, len);
    const int L = ((n + NW * 32 - 1) / (NW * 32)) * 32;
    if (MORE_WAVES && wave >= NW) {
        lds_barrier();
        lds_barrier();
        lds_barrier();
        return res[0];
    }
    if (lane == 0) segsum[wave] = ds_mine;
    lds_barrier();
    // the exactly rounded sum of everything before the segment (the order of THIS sum is free: it only centres the candidates)
    double before = lane < wave ? segsum[lane & 15] : 0.0;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) before += __shfl_xor(before, off, 64);
    before = __shfl(before, 0, 64);
    float c = 0.f;  // wave 0 starts from the sum's own start
    if (wave > 0) {
        // the fp32 neighbours of the estimate: bit patterns in monotone order (an involution: negative values run backwards)
        int i = __builtin_bit_cast(int, (float)before);
        i ^= (i >> 31) & 0x7fffffff;
        i += lane - 32;
        i ^= (i >> 31) & 0x7fffffff;
        c = __builtin_bit_cast(float, i);
    }
    const float f = len <= 0 ? c : (ROWB ? sequential_sum_rowbcast(vals + b, len, c, lane) : sequential_sum_bcast(vals + b, len, c));
    mid();
    cand[wave * 64 + lane] = c;
    fin[wave * 64 + lane] = f;
    if (stamps && wave == 0 && lane == 0) stamps[0] = wall_clock64();  // (timing experiments: wave 0's chain done; all chains done; the walk done)
    lds_barrier();
    if (stamps && wave == 0 && lane == 0) stamps[1] = wall_clock64();
    if (wave == 0) {
        // four segments' candidates and end values into registers at once (one LDS round trip), then a short dependent walk: compare, first hit, read that lane
        float s = f;  // (segment 0: every lane started from 0)
#pragma unroll 1
        for (int base = 1; base < NW; base += 4) {
            float cw[4], fw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = base + j < NW ? base + j : NW - 1;
                cw[j] = cand[w * 64 + lane];
                fw[j] = fin[w * 64 + lane];
            }
            int start = base;
            for (;;) {  // (one pass unless a segment misses: then that segment is re-added from the true value and the walk resumes behind it)
                int miss = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // branch-free: compare -> ballot -> first set bit -> read that lane -> scalar selects (a dependent chain of ~6 instructions per segment)
                    const int w = base + j;
                    const unsigned long long hit = __ballot(__builtin_bit_cast(int, cw[j]) == __builtin_bit_cast(int, s));
                    const int src = __builtin_ctzll(hit | (1ull << 63));
                    const float fs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fw[j]), src));
                    const bool active = (w < NW) & (w >= start) & (miss == 0) & (w * L < n);
                    s = (active & (hit != 0)) ? fs : s;
                    miss = (active & (hit == 0)) ? w : miss;
                }
                if (miss == 0) break;
                const int bw = miss * L, lw = n - bw < L ? n - bw : L;
                s = ROWB ? sequential_sum_rowbcast(vals + bw, lw, s, lane) : sequential_sum_bcast(vals + bw, lw, s);
                start = miss + 1;
            }
        }
        if (lane == 0) res[0] = s;
        if (stamps && lane == 0) stamps[2] = wall_clock64();
    }
    lds_barrier();
    return res[0];
}

// ... with the segment sums formed here (values that already lie in LDS, visible to all waves)
template <int NW, bool ROWB = true, class Mid = SpecNoOp, bool MORE_WAVES = false>
__device__ __forceinline__ float sequential_sum_speculated(const float *vals, int n, float *sp, int wave, int lane, unsigned long long *stamps = nullptr, Mid mid = Mid()) {
    int b, len;
    speculated_segment<NW>(n, wave, b, len);
    double ds = 0.0;
    for (int k = b + lane; k < b + len; k += 64) ds += (double)vals[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ds += __shfl_xor(ds, off, 64);
    return sequential_sum_speculated<NW, ROWB, Mid, MORE_WAVES>(vals, n, sp, wave, lane, ds, stamps, mid);
}
constexpr int kSpecScratchFloats(int nw) { return 32 + 2 * nw * 64 + 16; }

constexpr int kSeqSumBcastMaxWaves = 256 * 8;  // up to ~two row-walking waves per SIMD: the chain is the limit; beyond: the LDS

// non-temporal 16-byte load: streamed weights are read exactly once by exactly one CU
__device__ __forceinline__ uint4_t load_nt(const uint4_t *p) { return __builtin_nontemporal_load(p); }

}  // namespace tce
