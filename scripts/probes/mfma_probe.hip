// Bottom-up probe for the W4A16 GEMM inner loop on gfx950: what rate does each added ingredient leave?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/probes/mfma_probe scripts/probes/mfma_probe.hip  (run the binary through gpurun)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half_t;
typedef half_t half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
typedef half_t half2_t __attribute__((ext_vector_type(2)));

// LEVEL 0: MFMA only (operands in registers). 1: + A fragments from LDS (ds_read_b128). 2: + unpack of 2 words per k-step.
// 3: + per-group scale fma into acc. 4: + weight global loads (one 16-byte chunk per n-tile per k-block, prefetched one ahead).
// 5: + A tile staging (global -> regs -> permute -> LDS, double buffer, barrier per k-block).
template <int LEVEL, int MT, int NT>
__global__ __launch_bounds__(256, 2) void probe(const uint4_t *W, const half_t *A, float *out, int nkb, int K) {
    __shared__ __attribute__((aligned(16))) uint4_t lds_a[2][MT * 4 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n16 = lane & 15, q = lane >> 4;
    for (int i = tid; i < 2 * MT * 4 * 64; i += 256) (&lds_a[0][0])[i] = uint4_t{0x3c003c00u + (unsigned)i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    float4_t acc[MT][NT];
    for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j) acc[i][j] = float4_t{0, 0, 0, 0};
    unsigned mlo, mhi;
    asm volatile("v_mov_b32 %0, 0x000F000F\n\tv_mov_b32 %1, 0x00F000F0" : "=v"(mlo), "=v"(mhi));
    unsigned mhi2;
    asm volatile("v_mov_b32 %0, 0x00F0000F" : "=v"(mhi2));
    const int nchunks = K >> 5;
    const size_t rowbase = ((size_t)blockIdx.x * 4 * NT * 16 + wave * NT * 16 + n16) % 4096;
    uint4_t wreg[NT], wnext[NT];
    for (int j = 0; j < NT; ++j) wreg[j] = W[(rowbase + j * 16) * nchunks + q];
    uint4_t areg[MT];
    half8_t afix = half8_t{1, 2, 3, 4, 5, 6, 7, 8};
    const half2_t zlo = half2_t{(half_t)-1032.f, (half_t)-1032.f}, zhi = half2_t{(half_t)-72.f, (half_t)-72.f}, sixteenth = half2_t{(half_t)0.0625f, (half_t)0.0625f};
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = LEVEL == 5 ? (kb & 1) : 0;
        if (LEVEL >= 4) {
            const int nx = kb + 1 < nkb ? kb + 1 : kb;
            for (int j = 0; j < NT; ++j) wnext[j] = W[(rowbase + j * 16) * nchunks + nx * 4 + q];
        }
        if (LEVEL == 5) {
            const int nx = kb + 1 < nkb ? kb + 1 : kb;
            for (int i = 0; i < MT; ++i) {
                const int e = i * 256 + tid, row = e >> 4, pc = e & 15;
                areg[i] = *reinterpret_cast<const uint4_t *>(A + (size_t)((blockIdx.x & 7) * MT * 16 + row) * K + nx * 128 + pc * 8);
            }
        }
        if (LEVEL >= 4) __builtin_amdgcn_sched_barrier(0);
        float4_t blk[MT][NT];
        for (int i = 0; i < MT; ++i)
            for (int j = 0; j < NT; ++j) blk[i][j] = LEVEL >= 3 ? float4_t{0, 0, 0, 0} : acc[i][j];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            half8_t bf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (LEVEL == 6 || LEVEL == 8) {
                    const unsigned w = wreg[j][s];
                    half2_t d[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned rep = __builtin_amdgcn_perm(w, w, 0x01010101u * (unsigned)r);
                        d[r] = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, (rep & mhi2) | 0x64006400u), sixteenth, zhi);
                    }
                    bf[j] = half8_t{d[0].x, d[0].y, d[1].x, d[1].y, d[2].x, d[2].y, d[3].x, d[3].y};
                } else if (LEVEL >= 2) {
                    const unsigned w = wreg[j][s] + (LEVEL >= 4 ? 0u : (unsigned)kb);
                    const unsigned w8 = w >> 8;
                    const half2_t t0 = __builtin_bit_cast(half2_t, (w & mlo) | 0x64006400u) + zlo;
                    const half2_t t1 = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, (w & mhi) | 0x64006400u), sixteenth, zhi);
                    const half2_t t2 = __builtin_bit_cast(half2_t, (w8 & mlo) | 0x64006400u) + zlo;
                    const half2_t t3 = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, (w8 & mhi) | 0x64006400u), sixteenth, zhi);
                    bf[j] = half8_t{t0.x, t0.y, t1.x, t1.y, t2.x, t2.y, t3.x, t3.y};
                } else {
                    bf[j] = __builtin_bit_cast(half8_t, wreg[j]);
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                half8_t af = afix;
                if (LEVEL >= 1) af = __builtin_bit_cast(half8_t, lds_a[buf][(i * 4 + s) * 64 + q * 16 + (n16 ^ s ^ (q << 2))]);
#pragma unroll
                for (int j = 0; j < NT; ++j) blk[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], blk[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < MT; ++i)
            for (int j = 0; j < NT; ++j) {
                if (LEVEL >= 3) {
                    const float sc = 0.5f + (float)j;
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(sc, blk[i][j][r], acc[i][j][r]);
                } else acc[i][j] = blk[i][j];
            }
        if (LEVEL >= 4)
            for (int j = 0; j < NT; ++j) wreg[j] = wnext[j];
        if (LEVEL == 7 || LEVEL == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (LEVEL == 5) {
            for (int i = 0; i < MT; ++i) {
                const int e = i * 256 + tid, row = e >> 4, pc = e & 15, qq = pc >> 2, s = pc & 3;
                uint4_t v = areg[i], r;
                r.x = (v.x & 0xFFFFu) | (v.z << 16); r.y = (v.x >> 16) | (v.z & 0xFFFF0000u); r.z = (v.y & 0xFFFFu) | (v.w << 16); r.w = (v.y >> 16) | (v.w & 0xFFFF0000u);
                lds_a[buf ^ 1][((row >> 4) * 4 + s) * 64 + qq * 16 + ((row & 15) ^ s ^ (qq << 2))] = r;
            }
            __syncthreads();
        }
    }
    float t = 0;
    for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.678f) out[tid] = t;
}


typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_void_t;
// VARIANT 0: W through ordinary global loads (compiler-managed waits). 1: W loads by inline asm + counted waits.
template <int VARIANT, int MT, int NT>
__global__ __launch_bounds__(256, 2) void probe9(const uint4_t *W, const half_t *A, float *out, int nkb, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int A_BYTES = MT * 16 * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, q = lane >> 4;
    float4_t acc[MT][NT];
    for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j) acc[i][j] = float4_t{0, 0, 0, 0};
    unsigned mhi2;
    asm volatile("v_mov_b32 %0, 0x00F0000F" : "=v"(mhi2));
    const int nchunks = K >> 5;
    const size_t rowbase = ((size_t)blockIdx.x * 4 * NT * 16 + wave * NT * 16 + n16) % 4096;
    const uint4_t *wp[NT];
    for (int j = 0; j < NT; ++j) wp[j] = W + (rowbase + j * 16) * nchunks + q;
    const char *a_src[MT];
    for (int i = 0; i < MT; ++i) {
        const int row = (i * 4 + wave) * 4 + (lane >> 4), p = lane & 15;
        a_src[i] = reinterpret_cast<const char *>(A + (size_t)((blockIdx.x & 7) * MT * 16 + row) * K) + ((p ^ (row & 15)) << 4);
    }
    int a_off[4];
    for (int s = 0; s < 4; ++s) a_off[s] = n16 * 256 + (((q * 4 + s) ^ n16) << 4);
    auto issue = [&](int stage, int kb) {
        for (int i = 0; i < MT; ++i)
            __builtin_amdgcn_global_load_lds((global_void_t *)(a_src[i] + (size_t)kb * 256), (lds_void_t *)(smem + stage * A_BYTES + (i * 4 + wave) * 1024), 16, 0, 0);
    };
    const half2_t zc = __builtin_bit_cast(half2_t, 0xD480E408u), mulc = __builtin_bit_cast(half2_t, 0x2C003C00u);
    uint4_t wreg[NT], wnext[NT];
    for (int j = 0; j < NT; ++j) wreg[j] = wp[j][0];
    issue(0, 0);
    issue(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT) : "memory");
    __builtin_amdgcn_s_barrier();
    int stage = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        const int nx = kb + 1 < nkb ? kb + 1 : kb;
        const int nx2 = kb + 2 < nkb ? kb + 2 : nkb - 1;
        if (VARIANT == 0) {
            for (int j = 0; j < NT; ++j) wnext[j] = wp[j][nx * 4];
        } else {
            for (int j = 0; j < NT; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wnext[j]) : "v"(wp[j] + nx * 4) : "memory");
        }
        issue(stage >= 1 ? stage - 1 : 2, nx2);
        const unsigned char *st = smem + stage * A_BYTES;
        float4_t blk[MT][NT];
        for (int i = 0; i < MT; ++i)
            for (int j = 0; j < NT; ++j) blk[i][j] = float4_t{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            half8_t bf[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const unsigned w = wreg[j][s];
                half2_t d[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned rep = __builtin_amdgcn_perm(w, w, 0x01010101u * (unsigned)r);
                    d[r] = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, (rep & mhi2) | 0x64006400u), mulc, zc);
                }
                bf[j] = half8_t{d[0].x, d[0].y, d[1].x, d[1].y, d[2].x, d[2].y, d[3].x, d[3].y};
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const half8_t af = *reinterpret_cast<const half8_t *>(st + a_off[s] + i * 4096);
#pragma unroll
                for (int j = 0; j < NT; ++j) blk[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], blk[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < MT; ++i)
            for (int j = 0; j < NT; ++j) {
                const float sc = 0.5f + (float)j;
                for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(sc, blk[i][j][r], acc[i][j][r]);
            }
        if (VARIANT == 0) {
            for (int j = 0; j < NT; ++j) wreg[j] = wnext[j];
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT) : "memory");
        } else {
            // W(kb+1) was issued before the MT DMAs of block kb+2: all but those have landed
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT) : "memory");
            for (int j = 0; j < NT; ++j) {
                asm volatile("" : "+v"(wnext[j]));
                wreg[j] = wnext[j];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage = stage == 2 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0;
    for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.678f) out[tid] = t;
}

template <int VARIANT, int MT, int NT>
void run9(const uint4_t *W, const half_t *A, float *out, int blocks, int nkb, int K) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 3 * MT * 16 * 256;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe9<VARIANT, MT, NT>), dim3(blocks), dim3(256), lds, 0, W, A, out, nkb, K);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe9<VARIANT, MT, NT>), dim3(blocks), dim3(256), lds, 0, W, A, out, nkb, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / reps;
    const double flops = (double)blocks * 4 * MT * NT * 4 * nkb * 16384.0;
    printf("{\"level\": \"9.%d dma\", \"tile\": [%d, %d], \"blocks\": %d, \"us\": %.1f, \"TF\": %.0f}\n", VARIANT, MT, NT, blocks, us, flops / us / 1e6);
    fflush(stdout);
}

template <int LEVEL, int MT, int NT>
void run(const uint4_t *W, const half_t *A, float *out, int blocks, int nkb, int K) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<LEVEL, MT, NT>), dim3(blocks), dim3(256), 0, 0, W, A, out, nkb, K);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<LEVEL, MT, NT>), dim3(blocks), dim3(256), 0, 0, W, A, out, nkb, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / reps;
    const double flops = (double)blocks * 4 * MT * NT * 4 * nkb * 16384.0;
    printf("{\"level\": %d, \"tile\": [%d, %d], \"blocks\": %d, \"us\": %.1f, \"TF\": %.0f}\n", LEVEL, MT, NT, blocks, us, flops / us / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int K = 4096, nkb = K / 128;
    uint4_t *W; half_t *A; float *out;
    hipMalloc(&W, (size_t)4096 * K / 2 + 4096); hipMalloc(&A, (size_t)512 * K * 2 + 4096); hipMalloc(&out, 4096);
    hipMemset(W, 0x55, (size_t)4096 * K / 2); hipMemset(A, 0x3c, (size_t)512 * K * 2);
    for (int blocks : {256, 512, 2048}) {
        run<0, 4, 2>(W, A, out, blocks, nkb, K);
        run<1, 4, 2>(W, A, out, blocks, nkb, K);
        run<2, 4, 2>(W, A, out, blocks, nkb, K);
        run<3, 4, 2>(W, A, out, blocks, nkb, K);
        run<4, 4, 2>(W, A, out, blocks, nkb, K);
        run<5, 4, 2>(W, A, out, blocks, nkb, K);
        run<6, 4, 2>(W, A, out, blocks, nkb, K);
        run<7, 4, 2>(W, A, out, blocks, nkb, K);
        run<8, 4, 2>(W, A, out, blocks, nkb, K);
        run9<0, 4, 2>(W, A, out, blocks, nkb, K);
        run9<1, 4, 2>(W, A, out, blocks, nkb, K);
    }
    for (int blocks : {512, 2048}) {
        run<0, 4, 1>(W, A, out, blocks, nkb, K);
        run<1, 4, 1>(W, A, out, blocks, nkb, K);
        run<2, 4, 1>(W, A, out, blocks, nkb, K);
        run<3, 4, 1>(W, A, out, blocks, nkb, K);
        run<5, 4, 1>(W, A, out, blocks, nkb, K);
    }
    return 0;
}
