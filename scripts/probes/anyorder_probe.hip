// anyorder_probe.hip -- does hipExtAnyOrderLaunch drop the barrier between consecutive kernels of ONE stream on gfx950 / ROCm 7.2?
// (hip_ext.h says the flag is "not supported on AMD GFX9xx boards".)  Two kernels that each keep 64 workgroups busy for ~20 us: overlapped they take
// ~20 us together, serialised ~40 us.  Prints one JSON line.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (sink && ticks == 0) *sink = 1;
}
static float run(int flags, int n, hipStream_t s) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    for (int i = 0; i < n; ++i) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, flags, 2000ull, (unsigned *)nullptr);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f;
}
int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    run(0, 2, s);
    const float plain = run(0, 4, s), any = run(hipExtAnyOrderLaunch, 4, s);
    printf("{\"four_20us_kernels_plain_us\": %.1f, \"four_20us_kernels_any_order_us\": %.1f, \"last_error\": \"%s\"}\n", plain, any, hipGetErrorString(hipGetLastError()));
    return 0;
}
