// VALU / MFMA issue probe for gfx950: does a wave64 VALU instruction occupy its SIMD for 4 cycles
//  or 2 (MI355X_MICROARCH.md "SIMD-32")?  And how many VALU instructions issue in the shadow of one
// MFMA, from the same wave and from a second wave on the SIMD?  Pure register kernels, no memory traffic.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/probes/valu_probe scripts/probes/valu_probe.hip   (run through gpurun)
// output: one JSON line per (instruction mix, waves per SIMD): wall ns per wave-instruction per SIMD and, with the shader clock read
// from s_memtime deltas (ticks = shader cycles per the guide; the 100 MHz s_memrealtime gives the wall time), cycles per instruction.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef half_t half8_t __attribute__((ext_vector_type(8)));
typedef half_t half4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

enum Op { AND_OR = 0, FMA_F32 = 1, PK_FMA_F16 = 2, LSHR = 3, PK_ADD_F16 = 4, MFMA16 = 5, MFMA4 = 6, MIX = 7, PERM = 8, CVT_PK = 9,
          DOT8_I4 = 10, MFMA_I8 = 11, CVT_F32_I32 = 12, AND_LIT = 13, MIX_I8 = 14, DOT8_LOADS = 15 };  // round 4: the int4-dot / int8-MFMA decode formulations
typedef int int4v_t __attribute__((ext_vector_type(4)));

#define REP8(X) X X X X X X X X

// 8 independent registers, one instruction each per REP: a dependent chain per register of length 8 * iters (latency 4-ish cycles is
// covered by the 7 other chains for a 2- or 4-cycle issue).
template <int OP, int VALU_PER_MFMA>
__global__ __launch_bounds__(1024) void probe(unsigned *out, int iters, unsigned long long *ticks) {
    extern __shared__ unsigned lds_pad[];  // 100 KiB of dynamic LDS: exactly one workgroup per CU, whatever its size
    if (iters < 0) lds_pad[threadIdx.x] = 1;
    unsigned r0 = threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19;
    const unsigned m = 0x000F000Fu + blockIdx.x;
    float4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    half8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {(half_t)threadIdx.x, 1, 1, 1, 1, 1, 1, 1};
    half4_t a4 = {1, 2, 3, 4}, b4 = {(half_t)threadIdx.x, 1, 1, 1};
    int4v_t i0 = {0, 0, 0, 0}, i1 = i0, i2 = i0, i3 = i0, ia = {(int)threadIdx.x, 1, 2, 3}, ib = {0x01010101, 0x02020202, (int)blockIdx.x, 5};
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == AND_OR) {
            REP8(asm volatile("v_and_or_b32 %0, %0, %8, %1\n\tv_and_or_b32 %1, %1, %8, %2\n\tv_and_or_b32 %2, %2, %8, %3\n\tv_and_or_b32 %3, %3, %8, %4\n\t"
                              "v_and_or_b32 %4, %4, %8, %5\n\tv_and_or_b32 %5, %5, %8, %6\n\tv_and_or_b32 %6, %6, %8, %7\n\tv_and_or_b32 %7, %7, %8, %0"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m));)
        } else if (OP == FMA_F32) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %1\n\tv_fma_f32 %1, %1, %8, %2\n\tv_fma_f32 %2, %2, %8, %3\n\tv_fma_f32 %3, %3, %8, %4\n\t"
                              "v_fma_f32 %4, %4, %8, %5\n\tv_fma_f32 %5, %5, %8, %6\n\tv_fma_f32 %6, %6, %8, %7\n\tv_fma_f32 %7, %7, %8, %0"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m));)
        } else if (OP == PK_FMA_F16) {
            REP8(asm volatile("v_pk_fma_f16 %0, %0, %8, %1\n\tv_pk_fma_f16 %1, %1, %8, %2\n\tv_pk_fma_f16 %2, %2, %8, %3\n\tv_pk_fma_f16 %3, %3, %8, %4\n\t"
                              "v_pk_fma_f16 %4, %4, %8, %5\n\tv_pk_fma_f16 %5, %5, %8, %6\n\tv_pk_fma_f16 %6, %6, %8, %7\n\tv_pk_fma_f16 %7, %7, %8, %0"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m));)
        } else if (OP == PK_ADD_F16) {
            REP8(asm volatile("v_pk_add_f16 %0, %0, %8\n\tv_pk_add_f16 %1, %1, %8\n\tv_pk_add_f16 %2, %2, %8\n\tv_pk_add_f16 %3, %3, %8\n\t"
                              "v_pk_add_f16 %4, %4, %8\n\tv_pk_add_f16 %5, %5, %8\n\tv_pk_add_f16 %6, %6, %8\n\tv_pk_add_f16 %7, %7, %8"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m));)
        } else if (OP == LSHR) {
            REP8(asm volatile("v_lshrrev_b32 %0, 1, %1\n\tv_lshrrev_b32 %1, 1, %2\n\tv_lshrrev_b32 %2, 1, %3\n\tv_lshrrev_b32 %3, 1, %4\n\t"
                              "v_lshrrev_b32 %4, 1, %5\n\tv_lshrrev_b32 %5, 1, %6\n\tv_lshrrev_b32 %6, 1, %7\n\tv_lshrrev_b32 %7, 1, %0"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));)
        } else if (OP == PERM) {
            REP8(asm volatile("v_perm_b32 %0, %0, %1, %8\n\tv_perm_b32 %1, %1, %2, %8\n\tv_perm_b32 %2, %2, %3, %8\n\tv_perm_b32 %3, %3, %4, %8\n\t"
                              "v_perm_b32 %4, %4, %5, %8\n\tv_perm_b32 %5, %5, %6, %8\n\tv_perm_b32 %6, %6, %7, %8\n\tv_perm_b32 %7, %7, %0, %8"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m));)
        } else if (OP == CVT_PK) {  // v_cvt_pk_f16_f32-class conversion (v_cvt_pkrtz_f16_f32)
            REP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n\tv_cvt_pkrtz_f16_f32 %1, %1, %2\n\tv_cvt_pkrtz_f16_f32 %2, %2, %3\n\tv_cvt_pkrtz_f16_f32 %3, %3, %4\n\t"
                              "v_cvt_pkrtz_f16_f32 %4, %4, %5\n\tv_cvt_pkrtz_f16_f32 %5, %5, %6\n\tv_cvt_pkrtz_f16_f32 %6, %6, %7\n\tv_cvt_pkrtz_f16_f32 %7, %7, %0"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));)
        } else if (OP == DOT8_I4 || OP == DOT8_LOADS) {  // v_dot8_i32_i4: 8 signed int4 products into an int32 (the judge's round-4 candidate)
            REP8(asm volatile("v_dot8_i32_i4 %0, %0, %8, %1\n\tv_dot8_i32_i4 %1, %1, %8, %2\n\tv_dot8_i32_i4 %2, %2, %8, %3\n\tv_dot8_i32_i4 %3, %3, %8, %4\n\t"
                              "v_dot8_i32_i4 %4, %4, %8, %5\n\tv_dot8_i32_i4 %5, %5, %8, %6\n\tv_dot8_i32_i4 %6, %6, %8, %7\n\tv_dot8_i32_i4 %7, %7, %8, %0"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(m));)
            if (OP == DOT8_LOADS) {  // ... beside a 16-byte global load per 64 dots (the GEMV's ratio is one load per ~29 instructions)
                const int4v_t q = __builtin_nontemporal_load(reinterpret_cast<const int4v_t *>(out) + ((i * 1024 + threadIdx.x) & 0xFFFF));
                r0 ^= q[0] & 1; r1 ^= q[1] & 1; r2 ^= q[2] & 1; r3 ^= q[3] & 1;
            }
        } else if (OP == CVT_F32_I32) {
            REP8(asm volatile("v_cvt_f32_i32 %0, %1\n\tv_cvt_f32_i32 %1, %2\n\tv_cvt_f32_i32 %2, %3\n\tv_cvt_f32_i32 %3, %4\n\t"
                              "v_cvt_f32_i32 %4, %5\n\tv_cvt_f32_i32 %5, %6\n\tv_cvt_f32_i32 %6, %7\n\tv_cvt_f32_i32 %7, %0"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));)
        } else if (OP == AND_LIT) {  // VOP2 with a 32-bit literal (8-byte encoding)
            REP8(asm volatile("v_and_b32 %0, 0x0f0f0f0f, %1\n\tv_and_b32 %1, 0x0f0f0f0f, %2\n\tv_and_b32 %2, 0x0f0f0f0f, %3\n\tv_and_b32 %3, 0x0f0f0f0f, %4\n\t"
                              "v_and_b32 %4, 0x0f0f0f0f, %5\n\tv_and_b32 %5, 0x0f0f0f0f, %6\n\tv_and_b32 %6, 0x0f0f0f0f, %7\n\tv_and_b32 %7, 0x0f0f0f0f, %0"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));)
        } else if (OP == MFMA_I8) {
            REP8(i0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, i0, 0, 0, 0); i1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, i1, 0, 0, 0);
                 i2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, i2, 0, 0, 0); i3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, i3, 0, 0, 0);
                 i0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, i0, 0, 0, 0); i1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, i1, 0, 0, 0);
                 i2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, i2, 0, 0, 0); i3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, i3, 0, 0, 0);)
        } else if (OP == MIX_I8) {  // per REP: 4 x { one i8 16x16x64 MFMA + VALU_PER_MFMA independent v_and_b32 (literal) }
#define ONE_MFMA_I8(ACC) ACC = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia, ib, ACC, 0, 0, 0);
#define VALU_I8_ONE(R) asm volatile("v_and_b32 %0, 0x7f0f0f0f, %0" : "+v"(R));
#define VALU_I8                                                    \
    if (VALU_PER_MFMA >= 1) { VALU_I8_ONE(r0) }                    \
    if (VALU_PER_MFMA >= 2) { VALU_I8_ONE(r1) }                    \
    if (VALU_PER_MFMA >= 3) { VALU_I8_ONE(r2) }                    \
    if (VALU_PER_MFMA >= 4) { VALU_I8_ONE(r3) }                    \
    if (VALU_PER_MFMA >= 5) { VALU_I8_ONE(r4) }                    \
    if (VALU_PER_MFMA >= 6) { VALU_I8_ONE(r5) }                    \
    if (VALU_PER_MFMA >= 7) { VALU_I8_ONE(r6) }                    \
    if (VALU_PER_MFMA >= 8) { VALU_I8_ONE(r7) }                    \
    if (VALU_PER_MFMA >= 12) { VALU_I8_ONE(r0) VALU_I8_ONE(r1) VALU_I8_ONE(r2) VALU_I8_ONE(r3) }
            REP8(ONE_MFMA_I8(i0) VALU_I8 ONE_MFMA_I8(i1) VALU_I8 ONE_MFMA_I8(i2) VALU_I8 ONE_MFMA_I8(i3) VALU_I8)
        } else if (OP == MFMA16) {  // 8 MFMAs per REP on 4 independent accumulators
            REP8(acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc1, 0, 0, 0);
                 acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc3, 0, 0, 0);
                 acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc1, 0, 0, 0);
                 acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc3, 0, 0, 0);)
        } else if (OP == MFMA4) {
            REP8(acc0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc1, 0, 0, 0);
                 acc2 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc3, 0, 0, 0);
                 acc0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc1, 0, 0, 0);
                 acc2 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc3, 0, 0, 0);)
        } else if (OP == MIX) {  // per REP: 4 x { one 16x16x32 MFMA + VALU_PER_MFMA independent v_and_or }
#define ONE_MFMA(ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, ACC, 0, 0, 0);
#define VALU_K                                                                                                                      \
    if (VALU_PER_MFMA >= 1) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r0) : "v"(m));                                       \
    if (VALU_PER_MFMA >= 2) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r1) : "v"(m));                                       \
    if (VALU_PER_MFMA >= 3) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r2) : "v"(m));                                       \
    if (VALU_PER_MFMA >= 4) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r3) : "v"(m));                                       \
    if (VALU_PER_MFMA >= 5) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r4) : "v"(m));                                       \
    if (VALU_PER_MFMA >= 6) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r5) : "v"(m));                                       \
    if (VALU_PER_MFMA >= 7) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r6) : "v"(m));                                       \
    if (VALU_PER_MFMA >= 8) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r7) : "v"(m));
            REP8(ONE_MFMA(acc0) VALU_K ONE_MFMA(acc1) VALU_K ONE_MFMA(acc2) VALU_K ONE_MFMA(acc3) VALU_K)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    // slowest wave of workgroup 0 (the SIMD arbiter favours the oldest wave: wave 0 alone finishes early)
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        atomicMax(&ticks[0], t1 - t0);
        atomicMax(&ticks[1], w1 - w0);
    }
    const float s = acc0[0] + acc1[1] + acc2[2] + acc3[3] + (float)(i0[0] + i1[1] + i2[2] + i3[3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = (r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) + (unsigned)s;  // unconditional: nothing can be sunk or dropped
}

template <int OP, int V>
void run(const char *name, int instr_per_iter, int mfma_per_iter, unsigned *out, unsigned long long *ticks) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {  // waves per SIMD: one workgroup of 256 * wps threads per CU
        const int iters = 20000 / wps;
        static bool attr_set = false;
        if (!attr_set) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe<OP, V>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        hipLaunchKernelGGL((probe<OP, V>), dim3(256), dim3(256 * wps), 100 * 1024, 0, out, 200, ticks);
        (void)hipDeviceSynchronize();
        (void)hipMemset(ticks, 0, 16);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<OP, V>), dim3(256), dim3(256 * wps), 100 * 1024, 0, out, iters, ticks);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        const double n_valu = (double)iters * instr_per_iter * wps, n_mfma = (double)iters * mfma_per_iter * wps;  // per SIMD
        const double wall_ns = (double)h[1] * 10.0;  // s_memrealtime: 100 MHz
        const double ghz = wall_ns > 0 ? (double)h[0] / wall_ns : 0;
        printf("{\"mix\": \"%s\", \"waves_per_simd\": %d, \"valu_per_simd\": %.0f, \"mfma_per_simd\": %.0f, \"event_us\": %.1f, \"wave0_wall_us\": %.1f, "
               "\"shader_GHz_from_memtime\": %.3f, \"ns_per_instr_per_simd\": %.3f, \"cycles_per_instr_per_simd\": %.2f}\n",
               name, wps, n_valu, n_mfma, ms * 1000.0, wall_ns / 1000.0, ghz, wall_ns / (n_valu + n_mfma), (double)h[0] / (n_valu + n_mfma));
        fflush(stdout);
    }
}

int main() {
    unsigned *out;
    unsigned long long *ticks;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&ticks, 64);
    run<AND_OR, 0>("v_and_or_b32 x64", 64, 0, out, ticks);
    run<FMA_F32, 0>("v_fma_f32 x64", 64, 0, out, ticks);
    run<PK_FMA_F16, 0>("v_pk_fma_f16 x64", 64, 0, out, ticks);
    run<PK_ADD_F16, 0>("v_pk_add_f16 x64", 64, 0, out, ticks);
    run<LSHR, 0>("v_lshrrev_b32 x64", 64, 0, out, ticks);
    run<PERM, 0>("v_perm_b32 x64", 64, 0, out, ticks);
    run<CVT_PK, 0>("v_cvt_pkrtz_f16_f32 x64", 64, 0, out, ticks);
    run<DOT8_I4, 0>("v_dot8_i32_i4 x64", 64, 0, out, ticks);
    run<DOT8_LOADS, 0>("v_dot8_i32_i4 x64 + one 16-byte nt load per lane", 64, 0, out, ticks);
    run<CVT_F32_I32, 0>("v_cvt_f32_i32 x64", 64, 0, out, ticks);
    run<AND_LIT, 0>("v_and_b32 literal x64", 64, 0, out, ticks);
    run<MFMA_I8, 0>("v_mfma_i32_16x16x64_i8 x64", 0, 64, out, ticks);
    run<MIX_I8, 0>("mfma_i8 + 0 valu", 0, 32, out, ticks);
    run<MIX_I8, 4>("mfma_i8 + 4 valu", 128, 32, out, ticks);
    run<MIX_I8, 8>("mfma_i8 + 8 valu", 256, 32, out, ticks);
    run<MIX_I8, 12>("mfma_i8 + 12 valu", 384, 32, out, ticks);
    run<MFMA16, 0>("v_mfma_f32_16x16x32_f16 x64", 0, 64, out, ticks);
    run<MFMA4, 0>("v_mfma_f32_4x4x4_f16 x64", 0, 64, out, ticks);
    run<MIX, 0>("mfma16 + 0 valu", 0, 32, out, ticks);
    run<MIX, 1>("mfma16 + 1 valu", 32, 32, out, ticks);
    run<MIX, 2>("mfma16 + 2 valu", 64, 32, out, ticks);
    run<MIX, 3>("mfma16 + 3 valu", 96, 32, out, ticks);
    run<MIX, 4>("mfma16 + 4 valu", 128, 32, out, ticks);
    run<MIX, 6>("mfma16 + 6 valu", 192, 32, out, ticks);
    run<MIX, 8>("mfma16 + 8 valu", 256, 32, out, ticks);
    return 0;
}
