// aql_signal_probe.cpp -- can a launch on a SECOND hardware queue be released by the device itself, early? Kernel A (queue 1) stores 0 into an HSA signal from
// device code at its start or at its end; queue 2 holds a barrier-AND packet on that signal followed by kernel B.  Both spin 20 us on 64 workgroups.
// A then B in one queue: ~45 us.  Released at A's start: ~25 us if the packet processor wakes up quickly.  Released at A's end: 45 us + the wake-up latency.
//   hipcc --genco --offload-arch=gfx950 spin.hip -o spin.hsaco ; g++ -O2 aql_probe.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o aql_probe
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/amd_hsa_signal.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>

#define CHECK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char *m = nullptr; hsa_status_string(s_, &m); fprintf(stderr, "%s failed: %s\n", #x, m ? m : "?"); exit(1); } } while (0)

static hsa_agent_t g_gpu; static bool g_have_gpu = false;
static hsa_status_t find_gpu(hsa_agent_t a, void *) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_kernarg_pool; static bool g_have_pool = false;
static hsa_status_t find_pool(hsa_amd_memory_pool_t p, void *) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_pool) { g_kernarg_pool = p; g_have_pool = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_dev_pool; static bool g_have_dev_pool = false;
static hsa_status_t find_dev_pool(hsa_amd_memory_pool_t p, void *) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    bool alloc = false; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (alloc && !g_have_dev_pool) { g_dev_pool = p; g_have_dev_pool = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_agent_t g_cpu; static bool g_have_cpu = false;
static hsa_status_t find_cpu(hsa_agent_t a, void *) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}

int main(int argc, char **argv) {
    const char *hsaco = argc > 1 ? argv[1] : "spin_sig.hsaco";
    CHECK(hsa_init());
    CHECK(hsa_iterate_agents(find_gpu, nullptr));
    CHECK(hsa_iterate_agents(find_cpu, nullptr));
    CHECK(hsa_amd_agent_iterate_memory_pools(g_cpu, find_pool, nullptr));
    if (!g_have_gpu || !g_have_cpu || !g_have_pool) { fprintf(stderr, "agents / pool missing\n"); return 1; }
    hsa_queue_t *q1 = nullptr, *q2 = nullptr;
    CHECK(hsa_queue_create(g_gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q1));
    CHECK(hsa_queue_create(g_gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q2));
    const int fd = open(hsaco, O_RDONLY);
    if (fd < 0) { perror(hsaco); return 1; }
    hsa_code_object_reader_t reader; CHECK(hsa_code_object_reader_create_from_file(fd, &reader));
    hsa_executable_t exe; CHECK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    CHECK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    CHECK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym; CHECK(hsa_executable_get_symbol_by_name(exe, "spin_sig.kd", &g_gpu, &sym));
    uint64_t kobj = 0; uint32_t group = 0, priv = 0;
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
    struct Args { unsigned long long ticks; long long *sig; int when; int id; unsigned long long *stamps; };
    char *kernarg = nullptr;
    CHECK(hsa_amd_memory_pool_allocate(g_kernarg_pool, 4096, 0, reinterpret_cast<void **>(&kernarg)));
    CHECK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kernarg));
    CHECK(hsa_amd_agent_iterate_memory_pools(g_gpu, find_dev_pool, nullptr));
    unsigned long long *stamps = nullptr;  // device memory (atomics on host memory cost ~100 us here), copied back after each run
    CHECK(hsa_amd_memory_pool_allocate(g_dev_pool, 4096, 0, reinterpret_cast<void **>(&stamps)));
    unsigned long long hstamps[4];
    hsa_signal_t rel, doneA, doneB;
    CHECK(hsa_signal_create(1, 0, nullptr, &rel));
    CHECK(hsa_signal_create(1, 0, nullptr, &doneA));
    CHECK(hsa_signal_create(1, 0, nullptr, &doneB));
    long long *rel_value = const_cast<long long *>(reinterpret_cast<volatile long long *>(&reinterpret_cast<amd_signal_t *>(rel.handle)->value));

    auto put_kernel = [&](hsa_queue_t *q, uint64_t idx, int argslot, hsa_signal_t done, bool barrier) {
        hsa_kernel_dispatch_packet_t *p = reinterpret_cast<hsa_kernel_dispatch_packet_t *>(q->base_address) + (idx & (q->size - 1));
        memset(reinterpret_cast<char *>(p) + 4, 0, sizeof(*p) - 4);
        p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p->workgroup_size_x = 64; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->grid_size_x = 64 * 64; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = priv; p->group_segment_size = group;
        p->kernel_object = kobj; p->kernarg_address = kernarg + 64 * argslot;
        p->completion_signal = done;
        uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_ACQUIRE_FENCE_SCOPE) |
                          (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_RELEASE_FENCE_SCOPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER);
        __atomic_store_n(reinterpret_cast<uint16_t *>(p), header, __ATOMIC_RELEASE);
    };
    auto put_barrier = [&](hsa_queue_t *q, uint64_t idx, hsa_signal_t dep) {
        hsa_barrier_and_packet_t *p = reinterpret_cast<hsa_barrier_and_packet_t *>(q->base_address) + (idx & (q->size - 1));
        memset(reinterpret_cast<char *>(p) + 4, 0, sizeof(*p) - 4);
        p->dep_signal[0] = dep;
        uint16_t header = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER);
        __atomic_store_n(reinterpret_cast<uint16_t *>(p), header, __ATOMIC_RELEASE);
    };
    double gap_start = 0, gap_end = 0;
    auto wait0 = [&](hsa_signal_t s) { while (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {} };
    // mode 0: A, B in queue 1 (the in-queue boundary); 1: B on queue 2 released by A at its start; 2: ... at its end
    auto run = [&](int mode) -> double {
        Args a{2000ull, rel_value, mode, 0, stamps}, b{2000ull, rel_value, 0, 1, stamps};
        hstamps[0] = hstamps[2] = ~0ull; hstamps[1] = hstamps[3] = 0ull;
        CHECK(hsa_memory_copy(stamps, hstamps, sizeof(hstamps)));
        memcpy(kernarg, &a, sizeof(a));
        memcpy(kernarg + 64, &b, sizeof(b));
        hsa_signal_store_relaxed(rel, 1);
        hsa_signal_store_relaxed(doneA, 1);
        hsa_signal_store_relaxed(doneB, 1);
        const auto t0 = std::chrono::steady_clock::now();
        if (mode == 0) {
            const uint64_t i = hsa_queue_add_write_index_relaxed(q1, 2);
            put_kernel(q1, i, 0, doneA, true);
            put_kernel(q1, i + 1, 1, doneB, true);
            hsa_signal_store_screlease(q1->doorbell_signal, i + 1);
        } else {
            const uint64_t j = hsa_queue_add_write_index_relaxed(q2, 2);
            put_barrier(q2, j, rel);
            put_kernel(q2, j + 1, 1, doneB, true);
            hsa_signal_store_screlease(q2->doorbell_signal, j + 1);
            const uint64_t i = hsa_queue_add_write_index_relaxed(q1, 1);
            put_kernel(q1, i, 0, doneA, true);
            hsa_signal_store_screlease(q1->doorbell_signal, i);
        }
        wait0(doneA);
        wait0(doneB);
        const auto t1 = std::chrono::steady_clock::now();
        CHECK(hsa_memory_copy(hstamps, stamps, sizeof(hstamps)));
        gap_start = (double)(long long)(hstamps[2] - hstamps[0]) / 100.0;  // B's first workgroup start minus A's, us
        gap_end = (double)(long long)(hstamps[2] - hstamps[1]) / 100.0;    // ... minus A's last workgroup end
        return std::chrono::duration<double, std::micro>(t1 - t0).count();
    };
    run(0); run(1);
    double r[3] = {1e9, 1e9, 1e9}, gs[3] = {1e9, 1e9, 1e9}, ge[3] = {1e9, 1e9, 1e9};
    for (int rep = 0; rep < 9; ++rep)
        for (int m = 0; m < 3; ++m) {
            r[m] = std::min(r[m], run(m));
            gs[m] = std::min(gs[m], gap_start);
            ge[m] = std::min(ge[m], gap_end);
        }
    printf("{\"host_clock_us\": {\"A_then_B_in_one_queue\": %.1f, \"B_on_queue_2_released_at_A_start\": %.1f, \"B_on_queue_2_released_at_A_end\": %.1f}, "
           "\"device_clock_B_start_minus_A_start_us\": [%.2f, %.2f, %.2f], \"device_clock_B_start_minus_A_end_us\": [%.2f, %.2f, %.2f]}\n",
           r[0], r[1], r[2], gs[0], gs[1], gs[2], ge[0], ge[1], ge[2]);
    return 0;
}
