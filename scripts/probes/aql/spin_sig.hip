// spin_sig.hip -- device side of aql_signal_probe: spin for `ticks` of the 100 MHz wall clock; one lane stores 0 into an HSA signal's value word at the
// start (when == 1) or at the end (when == 2) of the kernel -- what a producer launch would do when its last generation of workgroups starts.  Every workgroup
// folds its start / end wall-clock reading into stamps[2 id] (min) / stamps[2 id + 1] (max).
#include <hip/hip_runtime.h>
extern "C" __global__ void spin_sig(unsigned long long ticks, long long *sig_value, int when, int id, unsigned long long *stamps) {
    const bool me = blockIdx.x == 0 && threadIdx.x == 0;
    const unsigned long long t0 = wall_clock64();
    if (me && when == 1) __hip_atomic_store(sig_value, 0ll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) atomicMin(stamps + 2 * id, t0);
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) atomicMax(stamps + 2 * id + 1, wall_clock64());
    if (me && when == 2) __hip_atomic_store(sig_value, 0ll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
