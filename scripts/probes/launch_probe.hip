// How long does it take to get N workgroups of W waves onto the chip and off again?  Kernels that do nothing (one store
// that never happens), back to back on one stream, HIP events around 200 launches.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/launch_probe scripts/probes/launch_probe.hip  (run through gpurun)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void nop_kernel(int *p, int lds_words) {
    extern __shared__ int sm[];
    if (lds_words > 0 && threadIdx.x == 0) sm[0] = 1;
    if (p && blockIdx.x == 0x7fffffff) *p = sm[0];
}
int main() {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int cfgs[][3] = {{1, 64, 0}, {256, 64, 0}, {256, 256, 0}, {256, 1024, 0}, {1376, 256, 0}, {1376, 256, 16384}, {2752, 128, 0}, {5504, 64, 0}, {688, 512, 0}, {344, 1024, 0}, {768, 256, 0}, {8016, 256, 0}};
    for (auto &c : cfgs) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(nop_kernel, dim3(c[0]), dim3(c[1]), c[2], 0, nullptr, c[2]);
        (void)hipDeviceSynchronize();
        // graph of 200 launches: no host launch cost in the timed region
        hipStream_t s; (void)hipStreamCreate(&s);
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(nop_kernel, dim3(c[0]), dim3(c[1]), c[2], s, nullptr, c[2]);
        (void)hipStreamEndCapture(s, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        (void)hipEventRecord(e0, s);
        for (int r = 0; r < 5; ++r) (void)hipGraphLaunch(ge, s);
        (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("{\"workgroups\": %d, \"threads\": %d, \"lds_bytes\": %d, \"waves\": %d, \"us_per_launch\": %.2f}\n", c[0], c[1], c[2], c[0] * c[1] / 64, ms * 1000.0 / 1000.0);
        fflush(stdout);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(s);
    }
    return 0;
}
