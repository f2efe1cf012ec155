// What does it cost to hand 4096 activations from every workgroup of a persistent grid to every workgroup of the same grid --
// the step a decode "token kernel" needs between two dependent GEMV launches -- against simply ending the kernel and starting
// the next one ("data-tagged hand-off ... kill criterion < 3.2 us per hand-off on the 4096^2 link")?
//
// A stage = [stream this stage's weights: P 16-byte pieces per thread, from a 1 GiB ring, requested BEFORE the wait because they
// depend on nothing] + [wait for the previous stage's 4096 outputs] + [publish this workgroup's 16 outputs].  Three ways to wait:
//   tagged    every output is one 32-bit store (tag << 16 | value) with device scope; consumers poll THE DATA with coherent
//             (sc0 sc1) 16-byte loads until all four tags of their piece match -- no counter, no store acknowledgement
//   counter   round 1's token kernel: device-scope stores, s_waitcnt vmcnt(0), workgroup barrier, atomic arrive, one polling lane,
//             workgroup barrier, coherent read
//   launches  one kernel per stage in a hipGraph (stream order); activations first, then the weights
// P = 0 (the bare hand-off), 2 (8 MiB per stage = a 4096 x 4096 int4 linear), 6 (24 MiB: qkv / down), 11 (46 MiB: gate + up).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/handoff_probe scripts/probes/handoff_probe.hip   (run through gpurun)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

constexpr int NX = 4096;        // values handed over per stage
constexpr int G = 256;          // persistent workgroups: one per CU
constexpr int PER_WG = NX / G;  // outputs per workgroup and stage
constexpr int THREADS = 1024;
constexpr size_t RING_PIECES = (size_t)1 << 26;  // 1 GiB of 16-byte pieces

struct Args {
    unsigned *tagged;  // [2][NX]
    unsigned *arrive;  // [stages]
    const uint4_t *weights;
    unsigned *fail;
    unsigned long long *ticks;  // [2]: 100 MHz wall clock at the start and the end of workgroup 0
    unsigned *sink;
    int stages;
    int first_stage;  // launches mode: the stage this kernel is
};

template <int P>
__device__ __forceinline__ void request_weights(const Args &a, int s, uint4_t (&w)[P > 0 ? P : 1]) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const size_t piece = ((size_t)s * P + i) * (size_t)(G * THREADS) + (size_t)blockIdx.x * THREADS + threadIdx.x;
        w[i] = __builtin_nontemporal_load(a.weights + (piece & (RING_PIECES - 1)));
    }
}

// MODE 0 tagged, 1 counter
template <int MODE, int P>
__global__ __launch_bounds__(THREADS) void chain_kernel(const Args a) {
    __shared__ unsigned xs[NX];
    const int tid = threadIdx.x, b = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.tagged, 0, 2 * NX * 4, 0x00020000);
    unsigned acc = 0;
    if (b == 0 && tid == 0) a.ticks[0] = wall_clock64();
    bool dead = false;
    for (int s = 0; s < a.stages && !dead; ++s) {
        uint4_t w[P > 0 ? P : 1];
        request_weights<P>(a, s, w);
        __builtin_amdgcn_sched_barrier(0);
        uint4_t x = uint4_t{(unsigned)tid, 1u, 2u, 3u};
        if (s > 0) {
            const int off = ((s - 1) & 1) * NX * 4 + tid * 16;
            if constexpr (MODE == 0) {
                const unsigned tag = (unsigned)s;
                int tries = 0;
                for (;;) {
                    x = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, /*sc0 sc1*/ 17);
                    if ((x.x >> 16) == tag && (x.y >> 16) == tag && (x.z >> 16) == tag && (x.w >> 16) == tag) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (++tries > 40000) {
                        atomicExch(a.fail, 1u);
                        dead = true;
                        break;
                    }
                }
            } else {
                if (tid == 0) {
                    int tries = 0;
                    while (__hip_atomic_load(a.arrive + (s - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++tries > 40000) {
                            atomicExch(a.fail, 1u);
                            break;
                        }
                    }
                }
                __syncthreads();
                x = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 17);
                if (__hip_atomic_load(a.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dead = true;
            }
        }
        reinterpret_cast<uint4_t *>(xs)[tid] = uint4_t{x.x & 0xFFFFu, x.y & 0xFFFFu, x.z & 0xFFFFu, x.w & 0xFFFFu};
#pragma unroll
        for (int i = 0; i < P; ++i) acc ^= w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
        asm volatile("" : "+v"(acc));
        dead = __syncthreads_or(dead);  // all of this workgroup's weights have been consumed, the image is complete
        if (tid < PER_WG) {
            const unsigned v = (xs[(b * PER_WG + tid + 17) & (NX - 1)] + 1u + (acc & 0u)) & 0xFFFFu;
            __hip_atomic_store(a.tagged + (s & 1) * NX + b * PER_WG + tid, ((unsigned)(s + 1) << 16) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (MODE == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(a.arrive + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();  // the image may be overwritten
        }
    }
    if (b == 0 && tid == 0) a.ticks[1] = wall_clock64();
    if (acc == 0x12345678u) a.sink[0] = acc;
}

// one stage as a kernel of its own: the boundary orders it behind its predecessor
template <int P>
__global__ __launch_bounds__(THREADS) void stage_kernel(const Args a) {
    __shared__ unsigned xs[NX];
    const int tid = threadIdx.x, b = blockIdx.x, s = a.first_stage;
    uint4_t x = uint4_t{(unsigned)tid, 1u, 2u, 3u};
    if (s > 0) x = reinterpret_cast<const uint4_t *>(a.tagged + ((s - 1) & 1) * NX)[tid];
    uint4_t w[P > 0 ? P : 1];
    request_weights<P>(a, s, w);
    reinterpret_cast<uint4_t *>(xs)[tid] = uint4_t{x.x & 0xFFFFu, x.y & 0xFFFFu, x.z & 0xFFFFu, x.w & 0xFFFFu};
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) acc ^= w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
    asm volatile("" : "+v"(acc));
    __syncthreads();
    if (tid < PER_WG) {
        const unsigned v = (xs[(b * PER_WG + tid + 17) & (NX - 1)] + 1u + (acc & 0u)) & 0xFFFFu;
        a.tagged[(s & 1) * NX + b * PER_WG + tid] = ((unsigned)(s + 1) << 16) | v;
    }
    if (acc == 0x12345678u) a.sink[0] = acc;
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(2);                                                               \
        }                                                                          \
    } while (0)

template <int P>
static void run(Args a, hipStream_t st) {
    const int stages = a.stages;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned expect_last = 0;
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        unsigned long long best_ticks = ~0ull;
        unsigned fail = 0, last = 0;
        hipGraphExec_t ge = nullptr;
        hipGraph_t g = nullptr;
        if (mode == 2) {
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int s = 0; s < stages; ++s) {
                Args b = a;
                b.first_stage = s;
                hipLaunchKernelGGL(stage_kernel<P>, dim3(G), dim3(THREADS), 0, st, b);
            }
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemsetAsync(a.tagged, 0, 2 * NX * 4, st));
            CK(hipMemsetAsync(a.arrive, 0, stages * 4, st));
            CK(hipMemsetAsync(a.fail, 0, 4, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            if (mode == 0) hipLaunchKernelGGL((chain_kernel<0, P>), dim3(G), dim3(THREADS), 0, st, a);
            else if (mode == 1) hipLaunchKernelGGL((chain_kernel<1, P>), dim3(G), dim3(THREADS), 0, st, a);
            else CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
            unsigned long long t[2];
            CK(hipMemcpy(t, a.ticks, 16, hipMemcpyDeviceToHost));
            if (mode < 2 && rep > 0 && t[1] - t[0] < best_ticks) best_ticks = t[1] - t[0];
            CK(hipMemcpy(&fail, a.fail, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&last, a.tagged + ((stages - 1) & 1) * NX + 5, 4, hipMemcpyDeviceToHost));
            if (fail) break;
        }
        if (mode == 0) expect_last = last;
        printf("{\"wait\": \"%s\", \"weight_MiB_per_stage\": %.1f, \"stages\": %d, \"us_per_stage_events\": %.3f, \"us_per_stage_device_clock\": %.3f, "
               "\"timed_out\": %u, \"chain_value\": %u, \"same_value_as_tagged\": %s}\n",
               mode == 0 ? "tagged" : mode == 1 ? "counter" : "launches", P * 16.0 * G * THREADS / 1048576.0, stages, best * 1000.0 / stages,
               mode < 2 ? best_ticks * 0.01 / stages : 0.0, fail, last & 0xFFFFu, last == expect_last ? "true" : "false");
        fflush(stdout);
        if (ge) {
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    }
}

int main() {
    Args a{};
    const int stages = 400;
    a.stages = stages;
    CK(hipMalloc(&a.tagged, 2 * NX * 4));
    CK(hipMalloc(&a.arrive, stages * 4));
    CK(hipMalloc(&a.fail, 4));
    CK(hipMalloc(&a.ticks, 16));
    CK(hipMalloc(&a.sink, 4));
    void *ring;
    CK(hipMalloc(&ring, RING_PIECES * 16));
    CK(hipMemset(ring, 1, RING_PIECES * 16));
    a.weights = static_cast<const uint4_t *>(ring);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    run<0>(a, st);
    run<2>(a, st);
    run<6>(a, st);
    run<11>(a, st);
    return 0;
}
