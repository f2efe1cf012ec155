// aql_probe.cpp -- do consecutive kernel dispatch packets of ONE hardware queue overlap on gfx950 when their BARRIER bit is clear?
// (HIP sets it on every same-stream launch; hipExtAnyOrderLaunch, which would clear it, is a no-op here: scripts/probes/anyorder_probe.hip.)
// Pure HSA: own queue, the code object of spin.hip loaded through the HSA loader, N packets of 64 workgroups spinning 20 us each.
// Serialised: N x 20 us.  Overlapped: ~20 us.  Prints one JSON line.
//   hipcc --genco --offload-arch=gfx950 spin.hip -o spin.hsaco ; g++ -O2 aql_probe.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o aql_probe
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
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
static hsa_agent_t g_cpu; static bool g_have_cpu = false;
static hsa_status_t find_cpu(hsa_agent_t a, void *) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}

int main(int argc, char **argv) {
    const char *hsaco = argc > 1 ? argv[1] : "spin.hsaco";
    CHECK(hsa_init());
    CHECK(hsa_iterate_agents(find_gpu, nullptr));
    CHECK(hsa_iterate_agents(find_cpu, nullptr));
    if (!g_have_gpu || !g_have_cpu) { fprintf(stderr, "no gpu / cpu agent\n"); return 1; }
    CHECK(hsa_amd_agent_iterate_memory_pools(g_cpu, find_pool, nullptr));
    if (!g_have_pool) { fprintf(stderr, "no kernarg pool\n"); return 1; }
    hsa_queue_t *q = nullptr;
    CHECK(hsa_queue_create(g_gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    // code object
    const int fd = open(hsaco, O_RDONLY);
    if (fd < 0) { perror(hsaco); return 1; }
    hsa_code_object_reader_t reader; CHECK(hsa_code_object_reader_create_from_file(fd, &reader));
    hsa_executable_t exe; CHECK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    CHECK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    CHECK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym; CHECK(hsa_executable_get_symbol_by_name(exe, "spin.kd", &g_gpu, &sym));
    uint64_t kobj = 0; uint32_t kernarg_size = 0, group = 0, priv = 0;
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kernarg_size));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group));
    CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
    struct Args { unsigned long long ticks; unsigned *sink; };
    void *kernarg = nullptr;
    CHECK(hsa_amd_memory_pool_allocate(g_kernarg_pool, 4096, 0, &kernarg));
    CHECK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kernarg));
    memset(kernarg, 0, 4096);
    Args a{2000ull, nullptr};  // 20 us
    memcpy(kernarg, &a, sizeof(a));
    hsa_signal_t done; CHECK(hsa_signal_create(1, 0, nullptr, &done));

    auto run = [&](int n, bool barrier, bool fences = true) -> double {
        hsa_signal_store_relaxed(done, 1);
        const uint64_t base = hsa_queue_add_write_index_relaxed(q, n);
        for (int i = 0; i < n; ++i) {
            hsa_kernel_dispatch_packet_t *p = reinterpret_cast<hsa_kernel_dispatch_packet_t *>(q->base_address) + ((base + i) & (q->size - 1));
            memset(reinterpret_cast<char *>(p) + 4, 0, sizeof(*p) - 4);
            p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            p->workgroup_size_x = 64; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
            p->grid_size_x = 64 * 64; p->grid_size_y = 1; p->grid_size_z = 1;
            p->private_segment_size = priv; p->group_segment_size = group;
            p->kernel_object = kobj; p->kernarg_address = kernarg;
            p->completion_signal = i == n - 1 ? done : hsa_signal_t{0};
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {  // headers last (the packet becomes valid when its type is written)
            hsa_kernel_dispatch_packet_t *p = reinterpret_cast<hsa_kernel_dispatch_packet_t *>(q->base_address) + ((base + i) & (q->size - 1));
            uint16_t header = HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE;
            if (fences || i == 0) header |= HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_ACQUIRE_FENCE_SCOPE;
            if (fences || i == n - 1) header |= HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_RELEASE_FENCE_SCOPE;
            if (barrier || i == 0) header |= 1 << HSA_PACKET_HEADER_BARRIER;
            __atomic_store_n(reinterpret_cast<uint16_t *>(p), header, __ATOMIC_RELEASE);
        }
        hsa_signal_store_screlease(q->doorbell_signal, base + n - 1);
        while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) {}
        const auto t1 = std::chrono::steady_clock::now();
        return std::chrono::duration<double, std::micro>(t1 - t0).count();
    };
    run(2, true);
    double with_b = 1e9, without_b = 1e9, one = 1e9, bare = 1e9;
    for (int r = 0; r < 5; ++r) {
        one = std::min(one, run(1, true));
        with_b = std::min(with_b, run(4, true));
        without_b = std::min(without_b, run(4, false));
        bare = std::min(bare, run(4, false, false));
    }
    printf("{\"one_20us_kernel_us\": %.1f, \"four_with_barrier_bit_us\": %.1f, \"four_without_barrier_bit_us\": %.1f, \"four_without_barrier_bit_and_without_fences_between_us\": %.1f}\n", one, with_b, without_b, bare);
    return 0;
}
