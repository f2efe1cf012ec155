// spin.hip -- the device side of aql_probe: a kernel that keeps its workgroups busy for `ticks` of the 100 MHz wall clock
#include <hip/hip_runtime.h>
extern "C" __global__ void spin(unsigned long long ticks, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (sink && ticks == 0) *sink = 1;
}
