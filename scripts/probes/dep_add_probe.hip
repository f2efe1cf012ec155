// dep_add_probe.hip -- how long is ONE step of a chain of dependent fp32 additions on gfx950, by the form of the addition?
// The reference's LayerNormQ / softmax sums are sequential fp32 additions (llm/src/ops/LayerNormQ.cc:12-52, softmax.cc:8-33); bit-exactness
// keeps the order, so the length of that chain is the floor of those kernels.  One wave, N dependent additions, wall clock (100 MHz) and
// s_memtime (shader clock) around the chain.  Forms:
//   0  v_add_f32 acc, v[i], acc                       operands in the lane's own registers (values read from LDS by every lane: broadcast)
//   1  v_add_f32_dpp acc, cur, acc row_shl:j          what sequential_sum_lane0 does (a 16-value group held one value per lane)
//   2  as 0, with the values fetched by ds_read_b128 inside the loop, a 16-value buffer ahead (the candidate replacement, complete)
// Prints one JSON line: ns and shader cycles per addition.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int FORM>
__global__ __launch_bounds__(64) void chain(const float *x, int n, float mean, float *out, unsigned long long *ticks) {
    __shared__ __attribute__((aligned(16))) float lx[4096];
    for (int i = threadIdx.x; i < n; i += 64) lx[i] = x[i];
    __syncthreads();
    const int lane = threadIdx.x;
    float acc = 0.f;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long c0 = __builtin_readcyclecounter();
    if constexpr (FORM == 0) {
        float4 v[4];
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4 *>(lx + 4 * j);
        for (int base = 0; base < n; base += 16) {
            asm volatile(
                "v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0\n\tv_add_f32 %0, %3, %0\n\tv_add_f32 %0, %4, %0\n\t"
                "v_add_f32 %0, %5, %0\n\tv_add_f32 %0, %6, %0\n\tv_add_f32 %0, %7, %0\n\tv_add_f32 %0, %8, %0\n\t"
                "v_add_f32 %0, %9, %0\n\tv_add_f32 %0, %10, %0\n\tv_add_f32 %0, %11, %0\n\tv_add_f32 %0, %12, %0\n\t"
                "v_add_f32 %0, %13, %0\n\tv_add_f32 %0, %14, %0\n\tv_add_f32 %0, %15, %0\n\tv_add_f32 %0, %16, %0"
                : "+v"(acc)
                : "v"(v[0].x), "v"(v[0].y), "v"(v[0].z), "v"(v[0].w), "v"(v[1].x), "v"(v[1].y), "v"(v[1].z), "v"(v[1].w), "v"(v[2].x), "v"(v[2].y),
                  "v"(v[2].z), "v"(v[2].w), "v"(v[3].x), "v"(v[3].y), "v"(v[3].z), "v"(v[3].w));
        }
    } else if constexpr (FORM == 1) {
        const int l16 = lane & 15;
        float nxt = lx[l16];
        for (int base = 0; base < n; base += 16) {
            const float cur = nxt;
            const int nb = base + 16 < n ? base + 16 : base;
            nxt = lx[nb + l16];
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
    } else {
        // the candidate replacement: every lane reads the same 16-byte pieces (an LDS broadcast), two 16-value buffers in fixed registers, the next
        // buffer's reads in flight while the current one is added; ONE asm statement, so the compiler cannot move a register before its data landed
        unsigned p = (unsigned)(size_t)lx;  // LDS byte address
        int pairs = n >> 5;
        asm volatile(
            "ds_read_b128 v[200:203], %1\n\tds_read_b128 v[204:207], %1 offset:16\n\tds_read_b128 v[208:211], %1 offset:32\n\tds_read_b128 v[212:215], %1 offset:48\n\t"
            "ds_read_b128 v[216:219], %1 offset:64\n\tds_read_b128 v[220:223], %1 offset:80\n\tds_read_b128 v[224:227], %1 offset:96\n\tds_read_b128 v[228:231], %1 offset:112\n"
            "1:\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_add_f32 %0, v200, %0\n\tv_add_f32 %0, v201, %0\n\tv_add_f32 %0, v202, %0\n\tv_add_f32 %0, v203, %0\n\t"
            "v_add_f32 %0, v204, %0\n\tv_add_f32 %0, v205, %0\n\tv_add_f32 %0, v206, %0\n\tv_add_f32 %0, v207, %0\n\t"
            "v_add_f32 %0, v208, %0\n\tv_add_f32 %0, v209, %0\n\tv_add_f32 %0, v210, %0\n\tv_add_f32 %0, v211, %0\n\t"
            "v_add_f32 %0, v212, %0\n\tv_add_f32 %0, v213, %0\n\tv_add_f32 %0, v214, %0\n\tv_add_f32 %0, v215, %0\n\t"
            "ds_read_b128 v[200:203], %1 offset:128\n\tds_read_b128 v[204:207], %1 offset:144\n\tds_read_b128 v[208:211], %1 offset:160\n\tds_read_b128 v[212:215], %1 offset:176\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_add_f32 %0, v216, %0\n\tv_add_f32 %0, v217, %0\n\tv_add_f32 %0, v218, %0\n\tv_add_f32 %0, v219, %0\n\t"
            "v_add_f32 %0, v220, %0\n\tv_add_f32 %0, v221, %0\n\tv_add_f32 %0, v222, %0\n\tv_add_f32 %0, v223, %0\n\t"
            "v_add_f32 %0, v224, %0\n\tv_add_f32 %0, v225, %0\n\tv_add_f32 %0, v226, %0\n\tv_add_f32 %0, v227, %0\n\t"
            "v_add_f32 %0, v228, %0\n\tv_add_f32 %0, v229, %0\n\tv_add_f32 %0, v230, %0\n\tv_add_f32 %0, v231, %0\n\t"
            "ds_read_b128 v[216:219], %1 offset:192\n\tds_read_b128 v[220:223], %1 offset:208\n\tds_read_b128 v[224:227], %1 offset:224\n\tds_read_b128 v[228:231], %1 offset:240\n\t"
            "v_add_u32 %1, 0x80, %1\n\t"
            "s_sub_u32 %2, %2, 1\n\t"
            "s_cmp_lg_u32 %2, 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "+v"(acc), "+v"(p), "+s"(pairs)
            :
            : "memory", "scc", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217",
              "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231");
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    if (lane == 0) {
        out[0] = acc;
        ticks[0] = w1 - w0;
        ticks[1] = c1 - c0;
    }
}

int main() {
    const int n = 4096;
    std::vector<float> hx(n);
    for (int i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.f - 0.5f;
    float *x, *out;
    unsigned long long *ticks;
    hipMalloc(&x, n * 4); hipMalloc(&out, 16); hipMalloc(&ticks, 16);
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    int wall_khz = 100000;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("{\"n\": %d, \"wall_clock_khz\": %d", n, wall_khz);
    const char *names[3] = {"plain_registers", "dpp_row_shl", "plain_lds_broadcast_two_buffers"};
    for (int form = 0; form < 3; ++form) {
        unsigned long long best_w = ~0ull, best_c = ~0ull;
        float res = 0;
        for (int rep = 0; rep < 20; ++rep) {
            switch (form) {
                case 0: chain<0><<<1, 64>>>(x, n, 0.01f, out, ticks); break;
                case 1: chain<1><<<1, 64>>>(x, n, 0.01f, out, ticks); break;
                default: chain<2><<<1, 64>>>(x, n, 0.01f, out, ticks); break;
            }
            unsigned long long t[2];
            hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost);
            hipMemcpy(&res, out, 4, hipMemcpyDeviceToHost);
            if (t[0] < best_w) best_w = t[0];
            if (t[1] < best_c) best_c = t[1];
        }
        printf(", \"%s\": {\"ns_per_add\": %.2f, \"cycles_per_add\": %.2f, \"sum\": %.6f}", names[form], best_w * 1e6 / wall_khz / n, (double)best_c / n, res);
    }
    printf("}\n");
    return 0;
}
