// opt_attention.hip -- one decode step of the reference's SmoothQuant OPT attention between the q/k/v projections and out_proj as ONE launch.
//
// What it replaces per layer and step (llm/src/nn_modules/Int8OPTAttention.cc:203-281, m <= 8 new rows):
//     the KV append                  the reference copies the whole past into the other cache buffer and transposes the value cache (:214-236, 262-270);
//                                    here row pos + j of k_cache [heads][max_keys][hd] and column pos + j of vt_cache [heads][hd][max_keys] are written
//     qk_bmm (BMM_S8T_S8N_F32T)      scores[h][j][t] = fmul_rn((float) dot_int32(q_hj, key_ht), alpha_qk)          (kernels/ref/matmul_ref_int8.cc:108)
//     batch_Add + softmax + -> int8  exactly tce_opt_softmax_q's arithmetic (glue.hip): the in-place quirk of softmax.cc:13 included -- a row whose own maximum is
//                                    below 1 starts its running maximum from row (0, 0)'s first probability, which every workgroup can recompute for itself
//     pv_bmm (BMM_S8T_S8N_S8T)       out[j][h * hd + d] = clamp(roundf(fmul_rn((float) dot_int32(probs_hj, v_h[.][d]), alpha_pv)))   (:29-34)
// Every dot product is an int32 sum (exact in any order) and every floating-point step is the same operation in the same order as in the separate launches,
// so the attention rows and both caches are BIT-IDENTICAL to tce_opt_kv_append -> tce_w8a8_matmul -> tce_opt_softmax_q -> tce_w8a8_matmul (tests/test_gpu_w8a8.py).
//
// Workgroup = (head, new row j), 5 waves: waves 0-3 take the row's keys (a thread per key: a cache row of hd = 64 or 128 bytes -- OPT-125M / 1.3B and 6.7B -- against the query held in registers, 256 bytes of keys'
// loads in flight), wave 4 appends; row (0, 0) is evaluated -- by all five waves -- only if this row's maximum is below 1.  The softmax sums are sequential
// fp32 additions in key order; waves 0-3 walk them together (sequential_sum_speculated, tce_common.hpp: bit-identical, a quarter of the dependent chain).
// New keys / values (t >= pos) are read from the projections' output rows, not from the caches, so the m rows' workgroups do not depend on each other.
#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

namespace tce {

namespace {


struct OptAttnArgs {
    const int8_t *q, *kn, *vn;  // [m][ld]: the projections' output rows (head h: columns h * hd ..)
    int8_t *kc, *vtc;           // [heads][max_keys][hd], [heads][hd][max_keys]
    const float *mask;          // [m][tgz] additive
    int8_t *out;                // [m][ld]
    int heads, m, pos, max_keys, ld, tgz;
    float a_qk, a_pv;
};

template <int HD>
__device__ __forceinline__ int dot_row(const int4_t (&a)[HD / 16], const int8_t *b) {  // HD int8 products; b 16-byte aligned
    int acc = 0;
#pragma unroll
    for (int i = 0; i < HD / 16; ++i) {
        const int4_t w = *reinterpret_cast<const int4_t *>(b + 16 * i);
        acc = __builtin_amdgcn_sdot4(a[i].x, w.x, acc, false);
        acc = __builtin_amdgcn_sdot4(a[i].y, w.y, acc, false);
        acc = __builtin_amdgcn_sdot4(a[i].z, w.z, acc, false);
        acc = __builtin_amdgcn_sdot4(a[i].w, w.w, acc, false);
    }
    return acc;
}

// the masked scores of row (hs, js) for the keys t0, t0 + tstep, ... into ew[], four keys' loads in flight per thread; returns the thread's maximum
template <int HD>
__device__ __forceinline__ float score_row(const OptAttnArgs &a, int hs, int js, int t0, int tstep, float *ew) {
    constexpr int QW = HD / 16, U = 256 / HD;  // 16-byte pieces per row; keys in flight per thread (256 bytes of cache rows either way)
    int4_t qv[QW];
    const int8_t *qr = a.q + (size_t)js * a.ld + hs * HD;
#pragma unroll
    for (int i = 0; i < QW; ++i) qv[i] = *reinterpret_cast<const int4_t *>(qr + 16 * i);
    const float *mrow = a.mask + (size_t)js * a.tgz;
    float rmax = -__builtin_inff();
    for (int tb = t0; tb < a.tgz; tb += U * tstep) {
        int4_t kv[U][QW];
        float mk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tb + u * tstep, tc = t < a.tgz ? t : a.tgz - 1;
            const int8_t *key = tc < a.pos ? a.kc + ((size_t)hs * a.max_keys + tc) * HD : a.kn + (size_t)(tc - a.pos) * a.ld + hs * HD;
#pragma unroll
            for (int i = 0; i < QW; ++i) kv[u][i] = *reinterpret_cast<const int4_t *>(key + 16 * i);
            mk[u] = mrow[tc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tb + u * tstep;
            if (t < a.tgz) {
                int acc = 0;
#pragma unroll
                for (int i = 0; i < QW; ++i) {
                    acc = __builtin_amdgcn_sdot4(qv[i].x, kv[u][i].x, acc, false);
                    acc = __builtin_amdgcn_sdot4(qv[i].y, kv[u][i].y, acc, false);
                    acc = __builtin_amdgcn_sdot4(qv[i].z, kv[u][i].z, acc, false);
                    acc = __builtin_amdgcn_sdot4(qv[i].w, kv[u][i].w, acc, false);
                }
                const float v = __fmul_rn((float)acc, a.a_qk) + mk[u];  // BMM_S8T_S8N_F32T's epilogue, then batch_Add
                ew[t] = v;
                rmax = v > rmax ? v : rmax;
            }
        }
    }
    return rmax;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

template <int HD>
__global__ __launch_bounds__(320) void opt_attn_decode_kernel(const OptAttnArgs a) {
    constexpr int QW = HD / 16, DW = HD / 4;  // 16-byte pieces per row; head dimensions per wave in the last step (in passes of 16)
    extern __shared__ __attribute__((aligned(16))) float sm[];  // e[tgzp] | e0[tgzp] | probs int8 [tgz16] | red[16] | spec[kSpecScratchFloats(4)]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x, j = blockIdx.y;
    const int tgz = a.tgz, pos = a.pos;
    const int tgzp = (tgz + 3) & ~3, tgz16 = (tgz + 15) & ~15;
    float *e = sm, *e0 = sm + tgzp;
    int8_t *pq = reinterpret_cast<int8_t *>(sm + 2 * tgzp);
    float *red = reinterpret_cast<float *>(pq + tgz16);
    float *spec = red + 16;  // the speculated sums' scratch
    const bool row00 = h == 0 && j == 0;
    // ---- wave 4: the append (this row's key as a cache row, its value as a cache column) and the masked score [0][0][0]; waves 0-3: the row's scores ----
    if (wave == 4) {
        const int8_t *kr = a.kn + (size_t)j * a.ld + h * HD, *vr = a.vn + (size_t)j * a.ld + h * HD;
        if (lane < QW) *reinterpret_cast<int4_t *>(a.kc + ((size_t)h * a.max_keys + pos + j) * HD + 16 * lane) = *reinterpret_cast<const int4_t *>(kr + 16 * lane);
#pragma unroll
        for (int d = lane; d < HD; d += 64) a.vtc[((size_t)h * HD + d) * a.max_keys + pos + j] = vr[d];
        if (lane == 0) {
            int4_t q0[QW];
#pragma unroll
            for (int i = 0; i < QW; ++i) q0[i] = *reinterpret_cast<const int4_t *>(a.q + 16 * i);
            const int8_t *key0 = 0 < pos ? a.kc : a.kn;
            red[8] = __fmul_rn((float)dot_row<HD>(q0, key0), a.a_qk) + a.mask[0];
        }
    }
    // the cached value rows this lane will contract in the last step (16 head dimensions x its 16 keys) are requested NOW: they depend on nothing, and twelve
    // workgroups on twelve CUs have nothing else to hide a memory round trip behind
    const int pieces = (pos + 15) >> 4;
    const int8_t *vbase = a.vtc + ((size_t)h * HD + DW * (wave & 3)) * a.max_keys;  // (the wave's DW dimensions, consumed in passes of 16)
    int4_t vfirst[DW / 16][16];
    if (wave < 4) {
#pragma unroll
        for (int ps = 0; ps < DW / 16; ++ps)
#pragma unroll
            for (int dd = 0; dd < 16; ++dd)
                vfirst[ps][dd] = lane < pieces ? *reinterpret_cast<const int4_t *>(vbase + (size_t)(16 * ps + dd) * a.max_keys + 16 * lane) : int4_t{0, 0, 0, 0};
        const float r = wave_max(score_row<HD>(a, h, j, tid, 256, e));
        if (lane == 0) red[wave] = r;
    }
    __syncthreads();
    const float r03 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));  // this row's maximum
    const float v000 = red[8];
    const bool independent = row00 || r03 >= 1.0f;  // (tce_opt_softmax_q: a probability is <= 1, so such a row's maximum does not depend on row (0, 0))
    // ---- row (0, 0) up to its first probability, only if this row needs it: all five waves score it, waves 0-3 walk its sum ----
    if (!independent) {
        const float r = wave_max(score_row<HD>(a, 0, 0, tid, 320, e0));
        if (lane == 0) red[9 + wave] = r;
        __syncthreads();
        {
            float mx0 = v000;
#pragma unroll
            for (int w = 0; w < 5; ++w) mx0 = red[9 + w] > mx0 ? red[9 + w] : mx0;
            for (int t = tid; t < tgz; t += 320) e0[t] = expf(e0[t] - mx0);
            __syncthreads();
            // its sum: sequential fp32 additions in key order (softmax.cc:21-26) -- walked by waves 0-3 at once (sequential_sum_speculated, tce_common.hpp)
            const float sum0 = sequential_sum_speculated<4, false, SpecNoOp, true>(e0, tgz, spec, wave, lane);  // (wave 4 only keeps the barriers' count)
            if (tid == 0) red[5] = (float)((double)e0[0] / ((double)sum0 + 1e-10));
        }
        __syncthreads();
    }
    if (wave == 4) return;
    const float init = independent ? (row00 ? v000 : r03) : red[5];
    const float mx = r03 > init ? r03 : init;
    for (int t = tid; t < tgz; t += 256) e[t] = expf(e[t] - mx);
    __syncthreads();  // waves 0-3 only from here on (wave 4 has left: a barrier counts the waves still running)
    // the row's sum: sequential fp32 additions in key order, speculated across the four waves (bit-identical; 512 keys: 128 dependent additions per wave, not 512)
    const float sum = sequential_sum_speculated<4, false>(e, tgz, spec, wave, lane);
    const double denom = (double)sum + 1e-10;
    for (int t = tid; t < tgz16; t += 256) {
        int8_t qb = 0;
        if (t < tgz) {
            const float p = (float)((double)e[t] / denom);
            qb = (int8_t)(int)roundf(p * 127.0f);
        }
        pq[t] = qb;
    }
    __syncthreads();
    // ---- probabilities x values: wave w takes head dimensions DW w .. DW w + DW - 1 (DW = hd / 4, in passes of 16); a lane takes 16 keys of the cached part (the 16 rows' pieces requested
    //      together), lane 0 the new rows.  The piece that straddles pos: what lies behind pos in the cache is not part of the context -- its probabilities
    //      are masked out of the lane's copy (the bytes exist: max_keys is a multiple of 16) ----
#pragma unroll
    for (int pass = 0; pass < DW / 16; ++pass) {  // (head dimension 128: two passes of 16 dimensions per wave, both requested ahead)
        const int8_t *vpass = vbase + (size_t)16 * pass * a.max_keys;
        int acc[16];
#pragma unroll
        for (int dd = 0; dd < 16; ++dd) acc[dd] = 0;
        for (int p = lane, it = 0; p < pieces; p += 64, ++it) {
            int4_t pv = *reinterpret_cast<const int4_t *>(pq + 16 * p);
            const int nvalid = pos - 16 * p;  // >= 1
            if (nvalid < 16) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int vb = nvalid - 4 * w;
                    const unsigned mk = vb >= 4 ? 0xFFFFFFFFu : (vb <= 0 ? 0u : ((1u << (8 * vb)) - 1u));
                    pv[w] = (int)((unsigned)pv[w] & mk);
                }
            }
            int4_t vv[16];
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) vv[dd] = it == 0 ? vfirst[pass][dd] : *reinterpret_cast<const int4_t *>(vpass + (size_t)dd * a.max_keys + 16 * p);
#pragma unroll
            for (int dd = 0; dd < 16; ++dd) {
                acc[dd] = __builtin_amdgcn_sdot4(pv.x, vv[dd].x, acc[dd], false);
                acc[dd] = __builtin_amdgcn_sdot4(pv.y, vv[dd].y, acc[dd], false);
                acc[dd] = __builtin_amdgcn_sdot4(pv.z, vv[dd].z, acc[dd], false);
                acc[dd] = __builtin_amdgcn_sdot4(pv.w, vv[dd].w, acc[dd], false);
            }
        }
        if (lane == 0) {
            for (int t = pos; t < tgz; ++t) {
                const int pr = pq[t];
                const int8_t *vr = a.vn + (size_t)(t - pos) * a.ld + h * HD + DW * wave + 16 * pass;
#pragma unroll
                for (int dd = 0; dd < 16; ++dd) acc[dd] += pr * (int)vr[dd];
            }
        }
        int mine = 0;
#pragma unroll
        for (int dd = 0; dd < 16; ++dd) {
            int v = acc[dd];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == dd) mine = v;
        }
        if (lane < 16) {  // BMM_S8T_S8N_S8T's epilogue (matmul_ref_int8.cc:29-34, no bias)
            float r = roundf(__fmul_rn((float)mine, a.a_pv));
            r = fmaxf(r, -128.0f);
            r = fminf(r, 127.0f);
            a.out[(size_t)j * a.ld + h * HD + DW * wave + 16 * pass + lane] = (int8_t)(int)r;
        }
    }
}

}  // namespace

int launch_opt_attention_decode(const void *q, const void *kn, const void *vn, void *kc, void *vtc, const float *mask, void *out, int heads, int hd, int m, int pos,
                                int max_keys, int ld, float a_qk, float a_pv, hipStream_t stream, hipError_t *hip_err) {
    OptAttnArgs a{};
    a.q = static_cast<const int8_t *>(q);
    a.kn = static_cast<const int8_t *>(kn);
    a.vn = static_cast<const int8_t *>(vn);
    a.kc = static_cast<int8_t *>(kc);
    a.vtc = static_cast<int8_t *>(vtc);
    a.mask = mask;
    a.out = static_cast<int8_t *>(out);
    a.heads = heads;
    a.m = m;
    a.pos = pos;
    a.max_keys = max_keys;
    a.ld = ld;
    a.tgz = pos + m;
    a.a_qk = a_qk;
    a.a_pv = a_pv;
    const int tgzp = (a.tgz + 3) & ~3, tgz16 = (a.tgz + 15) & ~15;
    const size_t lds = (size_t)2 * tgzp * sizeof(float) + tgz16 + 16 * sizeof(float) + (size_t)kSpecScratchFloats(4) * sizeof(float);
    if (lds > 160 * 1024) return TCE_ERR_UNSUPPORTED_SHAPE;
    if (hd != 64 && hd != 128) return TCE_ERR_UNSUPPORTED_SHAPE;
    const auto kernel = hd == 64 ? opt_attn_decode_kernel<64> : opt_attn_decode_kernel<128>;
    if (lds > 64 * 1024) {
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) {
            if (hip_err) *hip_err = ea;
            return TCE_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kernel, dim3(heads, m), dim3(320), lds, stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (hip_err) *hip_err = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
