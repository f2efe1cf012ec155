// comm.hip -- the multi-GPU side of the path behind the C ABI (no reference counterpart: TinyChatEngine is
// single-device).  Every linear is sharded by output rows (tce_w4a16_shard: a contiguous byte range of the three q4_6 arrays),
// activations are replicated, and the ranks' fp16 output slices are joined by an ALL-GATHER.  At decode (M = 1) a slice is
// 1-4 KB: the exchange is pure latency, and a library collective costs O(10 us) per call against ~3 us of HBM time per block at
// 8 ranks.  So the gather here is a PEER WRITE over xGMI, one small kernel per exchange, capturable in the token's hipGraph:
//
//   window    per rank, fine-grained device memory, exported with hipIpcGetMemHandle and mapped by every peer
//             (one process per GPU), or handed over directly when several ranks live in one process (tests):
//             [slots][2 parities][vector bytes] data + [slots][2][8] arrival flags
//   exchange  the kernel of rank r (a) writes its slice into EVERY rank's window -- buffer (slot, parity of the slot's epoch) at
//             the slice's offset -- with 16-byte stores straight into peer memory, (b) after a system-scope release writes the
//             epoch into flag (slot, parity, r) of every window, (c) waits until its own window shows all ranks' flags at this
//             epoch (bounded: a rank that never arrives sets the status word after 2 s -- tce_comm_set_timeout_ms -- instead of hanging the queue),
//             (d) copies the complete vector from the window to an ordinary device buffer -- the next linear's activation.
//             Epochs are counted on the device (one word per slot), so the same captured kernel node is correct on every
//             replay; two buffers per slot are enough: a rank can start exchange e+1 (and write into peers' buffers of parity
//             e+1) while a peer still reads the vector of exchange e, but exchange e+2 cannot start before every rank has sent
//             its flags of e+1, i.e. has finished with e.
// M >= 17 (prefill) moves megabytes per exchange: that regime belongs to RCCL (ncclAllGather through torch.distributed in this
// repository's host code, tinychatengine_amd/decode.py); nothing here replaces it.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include "tce_common.hpp"
#include "w4a16_kernels.hpp"

#include <dlfcn.h>

#include <cstring>

namespace tce {

constexpr int kMaxRanks = 8;
constexpr int kFlagStride = 16;  // words between flags (their own 64-byte lines)
static_assert(kMaxRanks == kCommMaxRanks && kFlagStride == kCommFlagStride, "w4a16_kernels.hpp: the in-launch exchange addresses the same flag table");

struct Comm {
    int rank = 0, world = 1, slots = 0;
    int device = 0;                      // the HIP device the window lives on: launches and peer-access calls run with it current
    unsigned long long timeout_ticks = 200000000ull;  // bound of a wait, 100 MHz ticks (2 s; tce_comm_set_timeout_ms)
    size_t vec_bytes = 0, window_bytes = 0;
    unsigned char *window = nullptr;     // this rank's window
    unsigned char *peer[kMaxRanks] = {};  // every rank's window as mapped here (peer[rank] == window)
    bool ipc_opened[kMaxRanks] = {};
    unsigned *epochs = nullptr;          // [slots] exchanges completed per slot, + [1] status
    bool finegrained = false;
    void *nccl = nullptr;                // ncclComm_t once tce_comm_rccl_init ran: exchanges too large for the peer-write kernel go through RCCL
};

namespace {

size_t data_bytes(const Comm &c) { return (size_t)c.slots * 2 * c.vec_bytes; }
size_t flags_bytes(const Comm &c) { return (size_t)c.slots * 2 * kMaxRanks * kFlagStride * 4; }

struct GatherArgs {
    unsigned char *peer[kMaxRanks];
    unsigned *epochs;  // [slots] + status at [slots]
    const uint4_t *src;
    uint4_t *dst;
    int rank, world, slot, slots;
    unsigned slice16;      // 16-byte pieces per rank's slice
    size_t vec_bytes, flags_off;
    unsigned long long timeout_ticks;
};

__global__ __launch_bounds__(1024) void allgather_peer_kernel(const GatherArgs a) {
    const int tid = threadIdx.x;
    const unsigned e = a.epochs[a.slot] + 1u;  // this exchange's epoch (same on every rank: they all run the same sequence)
    const unsigned par = e & 1u;
    const size_t buf_off = ((size_t)a.slot * 2 + par) * a.vec_bytes;
    // (a) my slice into every window
    for (int p = 0; p < a.world; ++p) {
        uint4_t *dstp = reinterpret_cast<uint4_t *>(a.peer[p] + buf_off) + (size_t)a.rank * a.slice16;
        for (unsigned i = tid; i < a.slice16; i += 1024) dstp[i] = a.src[i];
    }
    // (b) release at system scope, then the flags
    __threadfence_system();
    __syncthreads();
    if (tid < a.world) {
        unsigned *flag = reinterpret_cast<unsigned *>(a.peer[tid] + a.flags_off) + (((size_t)a.slot * 2 + par) * kMaxRanks + a.rank) * kFlagStride;
        __hip_atomic_store(flag, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // (c) every rank's flag in MY window
    if (tid < a.world) {
        const unsigned *flag = reinterpret_cast<const unsigned *>(a.peer[a.rank] + a.flags_off) + (((size_t)a.slot * 2 + par) * kMaxRanks + tid) * kFlagStride;
        const unsigned long long t0 = wall_clock64();  // 100 MHz
        // a communicator that is already flagged does not wait again: one lost exchange costs one bound, not one bound per exchange after it (tce_comm_reset re-arms)
        const bool dead = __hip_atomic_load(a.epochs + a.slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        while (!dead && (int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > a.timeout_ticks) {  // 2 s unless tce_comm_set_timeout_ms says otherwise (ranks are processes of their own: a host-side stall -- a
                // collector pause, a code-object load -- must not void an exchange): give up, flag the communicator (tce_comm_status), tce_comm_reset re-arms it
                __hip_atomic_store(a.epochs + a.slots, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();  // acquire: the peers' data behind their flags
    // (d) the complete vector to the consumer's buffer
    const uint4_t *full = reinterpret_cast<const uint4_t *>(a.peer[a.rank] + buf_off);
    const unsigned total16 = a.slice16 * (unsigned)a.world;
    for (unsigned i = tid; i < total16; i += 1024) {
        uint4_t v;
        const unsigned *w = reinterpret_cast<const unsigned *>(full + i);
        v.x = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        v.y = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        v.z = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        v.w = __hip_atomic_load(w + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        a.dst[i] = v;
    }
    __syncthreads();
    if (tid == 0) a.epochs[a.slot] = e;
}

}  // namespace

int comm_create(int rank, int world, int max_vector_elems, int slots, Comm **out, hipError_t *he) {
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world || max_vector_elems <= 0 || slots < 1) return TCE_ERR_BAD_ARG;
    Comm *c = new (std::nothrow) Comm();
    if (!c) return TCE_ERR_BAD_ARG;
    c->rank = rank;
    c->world = world;
    c->slots = slots;
    if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
    c->vec_bytes = (((size_t)max_vector_elems * 2) + 255) & ~(size_t)255;
    c->window_bytes = data_bytes(*c) + flags_bytes(*c);
    void *p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, c->window_bytes, hipDeviceMallocFinegrained);
    c->finegrained = e == hipSuccess;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&p, c->window_bytes);  // single-device setups (tests): coherence within the device is all that is needed
    }
    if (e == hipSuccess) e = hipMemset(p, 0, c->window_bytes);
    // [slots] epochs, the status word, [slots] arrival counters of the exchanges that run inside a linear's launch (PeerGatherEpi)
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->epochs), (size_t)(2 * slots + 1) * 4);
    if (e == hipSuccess) e = hipMemset(c->epochs, 0, (size_t)(2 * slots + 1) * 4);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (p) (void)hipFree(p);
        if (c->epochs) (void)hipFree(c->epochs);
        delete c;
        if (he) *he = e;
        return TCE_ERR_HIP;
    }
    c->window = static_cast<unsigned char *>(p);
    c->peer[rank] = c->window;
    *out = c;
    return TCE_OK;
}

// RAII: make the communicator's device current for the calls below, restore the caller's afterwards
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

int comm_export(Comm *c, void *handle64, hipError_t *he) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI promises 64-byte handles");
    // a coarse-grained window mapped by another GPU is not coherent for the flag protocol: such a window is never exported
    // (TCE_COMM_ALLOW_COARSE=1 lifts this for ranks that share ONE device, where device-scope coherence is all that is needed)
    if (!c->finegrained && !getenv("TCE_COMM_ALLOW_COARSE")) return TCE_ERR_UNSUPPORTED_SHAPE;
    DeviceGuard guard(c->device);
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, c->window);
    if (e != hipSuccess) {
        if (he) *he = e;
        return TCE_ERR_HIP;
    }
    std::memcpy(handle64, &h, 64);
    return TCE_OK;
}

int comm_connect_ipc(Comm *c, const void *handles, hipError_t *he) {
    DeviceGuard guard(c->device);
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank || c->peer[p]) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const unsigned char *>(handles) + (size_t)p * 64, 64);
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            if (he) *he = e;
            return TCE_ERR_HIP;
        }
        c->peer[p] = static_cast<unsigned char *>(ptr);
        c->ipc_opened[p] = true;
        // the mapping must be usable from THIS device before a kernel stores through it (a bad mapping inside the gather kernel would
        // be a GPU memory fault, i.e. the end of the process; a failed copy is an error code): read one flag word of the peer's window
        unsigned probe = 0;
        const hipError_t e2 = hipMemcpy(&probe, c->peer[p] + data_bytes(*c), sizeof(probe), hipMemcpyDeviceToHost);
        if (e2 != hipSuccess) {
            if (he) *he = e2;
            return TCE_ERR_HIP;
        }
    }
    return TCE_OK;
}

// All ranks in ONE process (the reference's host is one process, one thread: Int4llamaForCausalLM.cu:40-44), one rank per device or several
// per device: the windows are handed over as pointers.  A peer on ANOTHER device needs peer access from this rank's device
// (hipDeviceEnablePeerAccess; "already enabled" is fine) and a fine-grained window (its flags are polled across the link).
int comm_connect_local(Comm *c, Comm *const *all, hipError_t *he) {
    for (int p = 0; p < c->world; ++p)
        if (!all[p] || all[p]->world != c->world || all[p]->rank != p || all[p]->window_bytes != c->window_bytes || all[p]->slots != c->slots) return TCE_ERR_BAD_ARG;
    DeviceGuard guard(c->device);
    for (int p = 0; p < c->world; ++p) {
        if (all[p]->device != c->device) {
            if (!all[p]->finegrained || !c->finegrained) return TCE_ERR_UNSUPPORTED_SHAPE;
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, c->device, all[p]->device);
            if (e == hipSuccess && !can) e = hipErrorPeerAccessUnsupported;
            if (e == hipSuccess) {
                e = hipDeviceEnablePeerAccess(all[p]->device, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) {
                    (void)hipGetLastError();
                    e = hipSuccess;
                }
            }
            if (e != hipSuccess) {
                if (he) *he = e;
                return TCE_ERR_HIP;
            }
        }
        c->peer[p] = all[p]->window;
    }
    return TCE_OK;
}

int comm_set_timeout_ms(Comm *c, int ms) {
    if (ms < 1 || ms > 600000) return TCE_ERR_BAD_ARG;
    c->timeout_ticks = (unsigned long long)ms * 100000ull;
    return TCE_OK;
}

// Re-arms a communicator whose status word was set by a timed-out wait.  COLLECTIVE in the host's sense: every rank calls it after the
// host has made sure that no exchange is in flight on any rank (a barrier + device synchronisation); the epochs stay as they are -- every
// rank's gather kernels ran to their end, timed out or not, so the per-slot counts still agree.
int comm_reset(Comm *c, hipError_t *he) {
    DeviceGuard guard(c->device);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemset(c->epochs + c->slots, 0, (size_t)(c->slots + 1) * 4);  // the status word and the arrival counters (a launch that was cut short may have left one)
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (he) *he = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

int comm_device(const Comm *c) { return c->device; }

void *comm_window(Comm *c) { return c->window; }
int comm_rank(const Comm *c) { return c->rank; }
int comm_world(const Comm *c) { return c->world; }
int comm_world_of(const Comm *c) { return c->world; }

int comm_status(Comm *c, hipError_t *he) {
    DeviceGuard guard(c->device);
    unsigned st = 0;
    const hipError_t e = hipMemcpy(&st, c->epochs + c->slots, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        if (he) *he = e;
        return TCE_ERR_HIP;
    }
    return st ? 1 : 0;
}

void comm_rccl_destroy(Comm *c);
void comm_destroy(Comm *c) {
    if (c) comm_rccl_destroy(c);
    if (!c) return;
    DeviceGuard guard(c->device);
    for (int p = 0; p < c->world; ++p)
        if (c->ipc_opened[p] && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
    if (c->window) (void)hipFree(c->window);
    if (c->epochs) (void)hipFree(c->epochs);
    delete c;
}

int launch_allgather_f16(Comm *c, int slot, const void *src_slice, void *dst_full, int n_total, hipStream_t stream, hipError_t *he) {
    if (slot < 0 || slot >= c->slots || n_total <= 0 || n_total % c->world) return TCE_ERR_BAD_ARG;
    const size_t slice_bytes = (size_t)(n_total / c->world) * 2;
    if (slice_bytes % 16 || (size_t)n_total * 2 > c->vec_bytes) return TCE_ERR_UNSUPPORTED_SHAPE;
    if ((reinterpret_cast<uintptr_t>(src_slice) | reinterpret_cast<uintptr_t>(dst_full)) & 15) return TCE_ERR_UNSUPPORTED_SHAPE;
    GatherArgs a{};
    for (int p = 0; p < c->world; ++p) {
        if (!c->peer[p]) return TCE_ERR_BAD_ARG;  // not connected
        a.peer[p] = c->peer[p];
    }
    a.epochs = c->epochs;
    a.src = static_cast<const uint4_t *>(src_slice);
    a.dst = static_cast<uint4_t *>(dst_full);
    a.rank = c->rank;
    a.world = c->world;
    a.slot = slot;
    a.slots = c->slots;
    a.slice16 = (unsigned)(slice_bytes / 16);
    a.vec_bytes = c->vec_bytes;
    a.flags_off = data_bytes(*c);
    a.timeout_ticks = c->timeout_ticks;
    DeviceGuard guard(c->device);  // one host thread driving several devices: the launch goes to the communicator's device (the stream must be one of that device's)
    hipLaunchKernelGGL(allgather_peer_kernel, dim3(1), dim3(1024), 0, stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (he) *he = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}


int comm_peer_gather_epi(Comm *c, int slot, void *dst_full, int n_total, PeerGatherEpi *out) {
    if (slot < 0 || slot >= c->slots || n_total <= 0 || n_total % c->world || !dst_full) return TCE_ERR_BAD_ARG;
    // 16-byte pieces of the complete vector; the slices themselves are written element by element by the tiles that computed them
    if (n_total % 8 || (size_t)n_total * 2 > c->vec_bytes || (reinterpret_cast<uintptr_t>(dst_full) & 15)) return TCE_ERR_UNSUPPORTED_SHAPE;
    PeerGatherEpi g{};
    for (int p = 0; p < c->world; ++p) {
        if (!c->peer[p]) return TCE_ERR_BAD_ARG;  // not connected
        g.peer[p] = c->peer[p];
    }
    g.epochs = c->epochs;
    g.dst = dst_full;
    g.rank = c->rank;
    g.world = c->world;
    g.slot = slot;
    g.slots = c->slots;
    g.slice_elems = (unsigned)(n_total / c->world);
    g.vec_bytes = c->vec_bytes;
    g.flags_off = data_bytes(*c);
    g.timeout_ticks = c->timeout_ticks;
    *out = g;
    return TCE_OK;
}

// ---- RCCL behind the same communicator (round 4): exchanges beyond the latency regime ----
// A decode exchange is 1-4 KB per rank: one peer-write kernel.  A prompt's exchange (0.65-1.97 MB per rank at M = 512) is a bandwidth problem, and
// a ring all-gather over the xGMI links is RCCL's job.  librccl is opened on first use (dlopen: the library itself does not link against it; a host that never
// asks for the large regime never loads it); the communicator comes from an opaque 128-byte ncclUniqueId blob the host exchanges exactly like the IPC handles.
struct Id128 {  // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
    char bytes[128];
};
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
RcclApi &rccl() {
    static RcclApi api = [] {
        RcclApi a;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
        }
        if (!a.lib) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.lib, "ncclAllGather"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.AllGather && a.CommDestroy;
        return a;
    }();
    return api;
}
constexpr int kNcclHalf = 6;  // ncclFloat16 / ncclHalf (rccl.h: ncclDataType_t)

// [P][M][n_loc] (what an all-gather of whole [M][n_loc] blocks produces) -> [M][P * n_loc]: the column-sharded rows side by side
__global__ __launch_bounds__(256) void interleave_rows_kernel(const uint4_t *__restrict__ ws, uint4_t *__restrict__ dst, int P, int M, int loc16, long long ld16) {
    const long long total = (long long)P * M * loc16;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i % loc16);
        const long long r = i / loc16;
        const int m = (int)(r % M), p = (int)(r / M);
        dst[(long long)m * ld16 + (long long)p * loc16 + j] = ws[i];
    }
}
}  // namespace

int comm_rccl_unique_id(void *id128) {
    RcclApi &a = rccl();
    if (!a.ok) return TCE_ERR_UNSUPPORTED_KIND;
    return a.GetUniqueId(id128) == 0 ? TCE_OK : TCE_ERR_HIP;
}
int comm_rccl_init(Comm *c, const void *id128) {
    RcclApi &a = rccl();
    if (!a.ok) return TCE_ERR_UNSUPPORTED_KIND;
    if (c->nccl) return TCE_OK;
    DeviceGuard guard(c->device);
    Id128 id;
    std::memcpy(id.bytes, id128, 128);
    void *comm = nullptr;
    if (a.CommInitRank(&comm, c->world, id, c->rank) != 0 || !comm) return TCE_ERR_HIP;
    c->nccl = comm;
    return TCE_OK;
}
bool comm_has_rccl(const Comm *c) { return c->nccl != nullptr; }
void comm_rccl_destroy(Comm *c) {
    if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl);
    c->nccl = nullptr;
}
// the peer-write kernel's regime: slices of at most 64 KiB (beyond: one workgroup copying is slower than the links) that fit the window
bool comm_peer_regime(const Comm *c, int n_total) {
    const size_t slice_bytes = (size_t)(n_total / c->world) * 2;
    return slice_bytes <= 64 * 1024 && (size_t)n_total * 2 <= c->vec_bytes;
}
// the last RCCL failure of this thread as text (ncclGetErrorString), for tce_last_error(): a failed ncclAllGather is not a HIP error
thread_local char g_rccl_error[160] = "";
const char *comm_rccl_last_error() { return g_rccl_error; }
int launch_allgather_rccl(Comm *c, const void *src_slice, void *dst_full, size_t n_per_rank, hipStream_t stream) {
    RcclApi &a = rccl();
    if (!a.ok || !c->nccl) return TCE_ERR_UNSUPPORTED_KIND;
    DeviceGuard guard(c->device);
    const int rc = a.AllGather(src_slice, dst_full, n_per_rank, kNcclHalf, c->nccl, stream);
    if (rc == 0) return TCE_OK;
    std::snprintf(g_rccl_error, sizeof g_rccl_error, "ncclAllGather failed: %s (ncclResult_t %d)", a.GetErrorString ? a.GetErrorString(rc) : "?", rc);
    return TCE_ERR_RCCL;
}
// fits the window at all (whatever the slice size): the peer-write kernel can carry it when there is no RCCL communicator -- slower than the links beyond 64 KiB
// slices, but correct (what tce_allgather_f16 has always done)
static bool comm_fits_window(const Comm *c, long long n_total) { return n_total > 0 && (size_t)n_total * 2 <= c->vec_bytes; }
size_t allgather_rows_workspace_bytes(int M, int n_total) { return (size_t)M * (size_t)n_total * 2; }
// src [M][n_total / world] (this rank's columns of M rows) -> dst [M][ldd] on every rank.  M = 1 is tce_allgather_f16.
int launch_allgather_rows_f16(Comm *c, int slot, const void *src, void *dst, int M, int n_total, int ldd, void *workspace, hipStream_t stream, hipError_t *he) {
    if (M < 1 || n_total <= 0 || n_total % c->world || ldd < n_total) return TCE_ERR_BAD_ARG;
    const int n_loc = n_total / c->world;
    if (n_loc % 8 || ldd % 8 || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15)) return TCE_ERR_UNSUPPORTED_SHAPE;
    if (M == 1) {
        if (comm_peer_regime(c, n_total) || (!c->nccl && comm_fits_window(c, n_total))) return launch_allgather_f16(c, slot, src, dst, n_total, stream, he);
        return launch_allgather_rccl(c, src, dst, (size_t)n_loc, stream);
    }
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15)) return TCE_ERR_BAD_ARG;
    // stage 1: every rank's whole [M][n_loc] block, rank-major, into the workspace
    const long long flat = (long long)M * n_total;
    int rc;
    if (flat <= 0x7FFFFFFF && (comm_peer_regime(c, (int)flat) || (!c->nccl && comm_fits_window(c, flat)))) rc = launch_allgather_f16(c, slot, src, workspace, (int)flat, stream, he);
    else rc = launch_allgather_rccl(c, src, workspace, (size_t)M * n_loc, stream);
    if (rc != TCE_OK) return rc;
    // stage 2: rows side by side
    DeviceGuard guard(c->device);
    const long long total16 = flat / 8;
    const int grid = (int)((total16 + 255) / 256 < 2048 ? (total16 + 255) / 256 : 2048);
    hipLaunchKernelGGL(interleave_rows_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const uint4_t *>(workspace), static_cast<uint4_t *>(dst), c->world, M, n_loc / 8,
                       (long long)ldd / 8);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (he) *he = e;
        return TCE_ERR_HIP;
    }
    return TCE_OK;
}

}  // namespace tce
