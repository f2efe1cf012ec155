// Round 6: a standalone reproducer for the lost-lanes defect of profiles/r5/pk_form2_g32_first_launch.txt.
//
// What the asm-variant sessions of round 6 established on the real kernel (profiles/r6/pk_lost_lanes_rule.md): the wrong outputs come from ONE instruction form,
//     v_pk_mul_f32 D, S0, S1 op_sel:[0,1]          (the LOW product takes S1's HIGH register: D.lo = S0.lo * S1.hi, D.hi = S0.hi * S1.hi)
// whose low product comes back as 0.0 in lanes 48-63, the more often the longer the wave's vector ALU sat idle in front of it (s_nop 64: 45-100 % of launches; the same
// 64 slots filled with v_mov: the base rate).  The same multiplies as plain v_mul_f32, or through a register pair {hi, hi} with op_sel_hi:[1,0], never failed.
//
// This file asks how little context the failure needs: waves per SIMD, MFMA traffic on the SIMD, the source being an MFMA result, idle length.  Every product is checked
// against the plain v_mul_f32 of the same registers inside the kernel; mismatches are counted per configuration and the first few recorded (lane, register, got, want).
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/pk_opsel_repro scripts/probes/pk_opsel_repro.hip     (run through gpurun)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

struct Rec {
    unsigned lane, reg, it, form;
    float got, want, s0, s1;
};
struct Args {
    const float *scales;  // [2]: (e0, e1) -- from memory, nothing folds
    unsigned *fails;      // [8]: mismatching LOW products, HIGH products per form (0: op_sel:[0,1]; 1: op_sel_hi:[1,0]; 2: op_sel:[1,0] on src0)
    Rec *recs;            // [64]
    unsigned *nrec;
    int iters;
    int other_stream_work;  // unused (the host runs copies on a second stream)
};

#define NOP16 "s_nop 15\n\t"
#define MOV16 "v_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\t"

// IDLE: idle wait states in front of the packed instructions (0: the 16 slots the MFMA -> VALU distance needs are filled with v_mov instead)
// MF: 1 = the sources are fresh MFMA results and the SIMD's matrix pipe is busy; 0 = plain vector values
// NT: tiles (accumulator quads) per wave
// FORM: the VOP3P instruction under test, D = op(S0 = an accumulator pair, S1 = the scale pair e)
//   0  v_pk_mul_f32 op_sel:[0,1]      D.lo = S0.lo * e.hi   D.hi = S0.hi * e.hi     <- the form hipcc emitted in the failing kernel
//   1  v_pk_mul_f32 op_sel_hi:[1,0]   D.lo = S0.lo * e.lo   D.hi = S0.hi * e.lo     <- the form the rescale in the loop has
//   2  v_pk_mul_f32 op_sel:[1,0]      D.lo = S0.hi * e.lo   D.hi = S0.hi * e.hi
//   3  v_pk_add_f32 op_sel:[0,1]      D.lo = S0.lo + e.hi   D.hi = S0.hi + e.hi
//   4  v_pk_fma_f32 op_sel:[0,1,0]    D.lo = fma(S0.lo, e.hi, c.lo)   D.hi = fma(S0.hi, e.hi, c.hi)
//   5  v_pk_mov_b32 op_sel:[1,0]      D.lo = S0.hi          D.hi = e.lo
//   6  v_pk_mov_b32 op_sel:[0,1]      D.lo = S0.lo          D.hi = e.hi
//   7  v_pk_mul_f32 op_sel:[1,1]      D.lo = S0.hi * e.hi   D.hi = S0.hi * e.hi
template <int IDLE, int MF, int NT, int FORM>
__global__ __launch_bounds__(1024) void repro_kernel(const Args a) {
    const int tid = threadIdx.x, lane = tid & 63;
    float2_t e = float2_t{a.scales[0], a.scales[1]};
    asm volatile("" : "+v"(e));
    float2_t cadd = float2_t{1.5f, -2.25f};
    asm volatile("" : "+v"(cadd));
    float4_t acc[NT];
    half8_t fa, fb;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        fa[k] = (_Float16)(float)((lane + k) % 5 - 2);
        fb[k] = (_Float16)(float)((lane * 3 + k) % 7 - 3);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = float4_t{(float)(tid + i), (float)(2 * tid + i) + 0.5f, (float)(3 * tid - i), (float)(tid - 7 * i) + 0.25f};
    unsigned bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < a.iters; ++it) {
        if constexpr (MF) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
        }
        // every accumulator pinned: nothing below is hoisted above the pad, and the pad (>= 16 wait states) covers the MFMA -> VALU distance
#pragma unroll
        for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(acc[i]));
        unsigned scratchv = 0;
        if constexpr (IDLE == 0) asm volatile(MOV16 : "+v"(scratchv));
        else if constexpr (IDLE == 16) asm volatile(NOP16 ::: "memory");
        else if constexpr (IDLE == 64) asm volatile(NOP16 NOP16 NOP16 NOP16 ::: "memory");
        else if constexpr (IDLE == 256) asm volatile(NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 NOP16 ::: "memory");
        else if constexpr (IDLE == 1000) {  // a real wait: a global load and its s_waitcnt (what the kernel's loop exit has)
            float tmp;
            asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(tmp) : "v"(a.scales) : "memory");
            asm volatile("" ::"v"(tmp));
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(acc[i]));
        // the instruction under test back to back over all tiles (the kernel's post-loop block: 32 packed multiplies in a row), results in fresh registers
        float2_t p[NT][2];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float2_t s = float2_t{acc[i][2 * h], acc[i][2 * h + 1]};
                if constexpr (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p[i][h]) : "v"(s), "v"(e));
                if constexpr (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p[i][h]) : "v"(s), "v"(e));
                if constexpr (FORM == 2) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(p[i][h]) : "v"(s), "v"(e));
                if constexpr (FORM == 3) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p[i][h]) : "v"(s), "v"(e));
                if constexpr (FORM == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(p[i][h]) : "v"(s), "v"(e), "v"(cadd));
                if constexpr (FORM == 5) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p[i][h]) : "v"(s), "v"(e));
                if constexpr (FORM == 6) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(p[i][h]) : "v"(s), "v"(e));
                if constexpr (FORM == 7) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]" : "=v"(p[i][h]) : "v"(s), "v"(e));
            }
        asm volatile("s_nop 7" ::: "memory");
        // the same values by plain (unpacked) instructions
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float s_lo = acc[i][2 * h], s_hi = acc[i][2 * h + 1];
                float wl, wh;
                if constexpr (FORM == 0) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(wl) : "v"(s_lo), "v"(e.y)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(wh) : "v"(s_hi), "v"(e.y)); }
                if constexpr (FORM == 1) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(wl) : "v"(s_lo), "v"(e.x)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(wh) : "v"(s_hi), "v"(e.x)); }
                if constexpr (FORM == 2) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(wl) : "v"(s_hi), "v"(e.x)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(wh) : "v"(s_hi), "v"(e.y)); }
                if constexpr (FORM == 3) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(wl) : "v"(s_lo), "v"(e.y)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(wh) : "v"(s_hi), "v"(e.y)); }
                if constexpr (FORM == 4) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(wl) : "v"(s_lo), "v"(e.y), "v"(cadd.x)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(wh) : "v"(s_hi), "v"(e.y), "v"(cadd.y)); }
                if constexpr (FORM == 5) { wl = s_hi; wh = e.x; }
                if constexpr (FORM == 6) { wl = s_lo; wh = e.y; }
                if constexpr (FORM == 7) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(wl) : "v"(s_hi), "v"(e.y)); wh = wl; }
                auto differ = [](float x, float y) { return __builtin_bit_cast(unsigned, x) != __builtin_bit_cast(unsigned, y); };
                auto rec = [&](unsigned reg, float got, float want) {
                    const unsigned n = atomicAdd(a.nrec, 1u);
                    if (n < 64) a.recs[n] = Rec{(unsigned)lane, (unsigned)(i * 4 + reg), (unsigned)it, (unsigned)FORM, got, want, s_lo, s_hi};
                };
                if (differ(p[i][h].x, wl)) ++bad_lo, rec(2 * h, p[i][h].x, wl);
                if (differ(p[i][h].y, wh)) ++bad_hi, rec(2 * h + 1, p[i][h].y, wh);
            }
        if constexpr (!MF) {  // keep the values moving (and finite)
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = acc[i] * 0.5f + float4_t{1.f, 2.f, 3.f, 4.f};
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = acc[i] * 0.25f;
        }
    }
    if (bad_lo) atomicAdd(a.fails + 0, bad_lo);
    if (bad_hi) atomicAdd(a.fails + 1, bad_hi);
}

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(2);                                                \
        }                                                           \
    } while (0)

static const char *kForms[8] = {"v_pk_mul_f32 op_sel:[0,1]", "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel:[1,0]", "v_pk_add_f32 op_sel:[0,1]", "v_pk_fma_f32 op_sel:[0,1,0]",
                                "v_pk_mov_b32 op_sel:[1,0]", "v_pk_mov_b32 op_sel:[0,1]", "v_pk_mul_f32 op_sel:[1,1]"};

template <int IDLE, int MF, int NT, int FORM>
static void run(Args a, int threads, size_t lds, hipStream_t st, hipStream_t side, void *big_a, void *big_b, size_t big) {
    CK(hipMemsetAsync(a.fails, 0, 32, st));
    CK(hipMemsetAsync(a.nrec, 0, 4, st));
    CK(hipStreamSynchronize(st));
    auto kfn = repro_kernel<IDLE, MF, NT, FORM>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int launches = 200;
    for (int l = 0; l < launches; ++l) {
        if (side && (l % 4) == 0) CK(hipMemcpyAsync(big_b, big_a, big, hipMemcpyDeviceToDevice, side));
        hipLaunchKernelGGL(kfn, dim3(256), dim3(threads), lds, st, a);
    }
    CK(hipStreamSynchronize(st));
    if (side) CK(hipStreamSynchronize(side));
    unsigned f[8], n;
    CK(hipMemcpy(f, a.fails, 32, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&n, a.nrec, 4, hipMemcpyDeviceToHost));
    const double products = (double)launches * 256 * threads * a.iters * NT * 2;
    printf("{\"form\": \"%s\", \"idle\": %d, \"mfma\": %d, \"tiles\": %d, \"threads\": %d, \"load\": %d, \"results_per_half\": %.3g, \"bad_lo\": %u, \"bad_hi\": %u",
           kForms[FORM], IDLE, MF, NT, threads, side ? 1 : 0, products, f[0], f[1]);
    if (n) {
        Rec r[64];
        CK(hipMemcpy(r, a.recs, sizeof(r), hipMemcpyDeviceToHost));
        unsigned lane_lo = 64, lane_hi = 0;
        for (unsigned i = 0; i < (n < 64 ? n : 64); ++i) {
            lane_lo = r[i].lane < lane_lo ? r[i].lane : lane_lo;
            lane_hi = r[i].lane > lane_hi ? r[i].lane : lane_hi;
        }
        printf(", \"lanes_of_first_64\": [%u, %u], \"first\": [", lane_lo, lane_hi);
        for (unsigned i = 0; i < (n < 4 ? n : 4); ++i)
            printf("%s{\"lane\": %u, \"reg\": %u, \"it\": %u, \"got\": %g, \"want\": %g}", i ? ", " : "", r[i].lane, r[i].reg, r[i].it, r[i].got, r[i].want);
        printf("]");
    }
    printf("}\n");
    fflush(stdout);
}

int main() {
    Args a{};
    float sc[2] = {3.0f, 0.0078125f * 5.0f};
    float *dsc;
    CK(hipMalloc(&dsc, 8));
    CK(hipMemcpy(dsc, sc, 8, hipMemcpyHostToDevice));
    a.scales = dsc;
    CK(hipMalloc(&a.fails, 32));
    CK(hipMalloc(&a.recs, 64 * sizeof(Rec)));
    CK(hipMalloc(&a.nrec, 4));
    a.iters = 64;
    hipStream_t st, side;
    CK(hipStreamCreate(&st));
    CK(hipStreamCreate(&side));
    const size_t big = (size_t)256 << 20;
    void *ba, *bb;
    CK(hipMalloc(&ba, big));
    CK(hipMalloc(&bb, big));
    CK(hipMemset(ba, 1, big));
    const size_t L = 128 << 10;  // one workgroup per CU
    // 1. every form, two waves per SIMD (512 threads: the failing kernel's residency), MFMA traffic, 64 / 256 idle wait states
#define ALL_FORMS(IDLE_, T_)                                                                                                     \
    run<IDLE_, 1, 8, 0>(a, T_, L, st, nullptr, ba, bb, big); run<IDLE_, 1, 8, 1>(a, T_, L, st, nullptr, ba, bb, big);           \
    run<IDLE_, 1, 8, 2>(a, T_, L, st, nullptr, ba, bb, big); run<IDLE_, 1, 8, 3>(a, T_, L, st, nullptr, ba, bb, big);           \
    run<IDLE_, 1, 8, 4>(a, T_, L, st, nullptr, ba, bb, big); run<IDLE_, 1, 8, 5>(a, T_, L, st, nullptr, ba, bb, big);           \
    run<IDLE_, 1, 8, 6>(a, T_, L, st, nullptr, ba, bb, big); run<IDLE_, 1, 8, 7>(a, T_, L, st, nullptr, ba, bb, big);
    ALL_FORMS(64, 512)
    ALL_FORMS(256, 512)
    // 2. the failing form across contexts: idle length, no MFMA traffic, one / four waves per SIMD, a loaded second stream
    run<0, 1, 8, 0>(a, 512, L, st, nullptr, ba, bb, big);
    run<16, 1, 8, 0>(a, 512, L, st, nullptr, ba, bb, big);
    run<1000, 1, 8, 0>(a, 512, L, st, nullptr, ba, bb, big);
    run<64, 0, 8, 0>(a, 512, L, st, nullptr, ba, bb, big);
    run<256, 0, 8, 0>(a, 512, L, st, nullptr, ba, bb, big);
    run<256, 1, 8, 0>(a, 256, L, st, nullptr, ba, bb, big);
    run<256, 1, 8, 0>(a, 1024, L, st, nullptr, ba, bb, big);
    run<256, 1, 8, 0>(a, 512, L, st, side, ba, bb, big);
    run<256, 1, 8, 1>(a, 512, L, st, side, ba, bb, big);
    return 0;
}
