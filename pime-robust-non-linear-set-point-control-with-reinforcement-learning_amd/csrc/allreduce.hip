// One-shot all-reduce (mean) of the flat gradient buffer over peer-mapped memory: the hand-written replacement of the RCCL
// all-reduce for the 270 KB message of SURVEY.md section 8(e) / f4 ("a custom one-shot xGMI all-reduce").
//
// The reference has NO collective (it is one process: /root/reference/elegantrl/run.py:232-247 is an unused mp.Pipe); this is
// new.  Why one-shot: the message is 67 460 floats.  A ring all-reduce over G ranks is 2 (G - 1) dependent steps of a 34 KB
// chunk -- latency times 14 at G = 8 --, while xGMI is a full point-to-point mesh: every rank can write its whole vector into
// every peer's inbox at once (G - 1 concurrent 270 KB writes over G - 1 different links), raise a flag, wait for the G - 1 flags
// raised on itself, and sum the G inbox rows locally IN RANK ORDER -- one step of latency, and every rank computes
// bit-identical sums (the replicas stay identical, as SURVEY section 8(e) requires).
//
// Protocol (one kernel launch per all-reduce, on the caller's stream, replayable from a HIP graph: the call sequence number lives
// in device memory):
//   region of rank r (fine-grained device memory, exported by hipIpcGetMemHandle, opened by every peer):
//     flags [2 parities][G sources] uint32, 64 B apart: chunk-arrival counters, monotonically increasing
//     inbox [2 parities][G sources][n] float
//   call k (parity p = k & 1), grid = G x C workgroups, workgroup (d, c):
//     1. push   chunk c of `data` -> inbox[p][my rank] of rank d (d = my rank: the local copy); __threadfence_system();
//               one system-scope atomic add on flags[p][my rank] of rank d
//     2. wait   until my flags[p][s] >= (calls of parity p so far + 1) * C for every source s (bounded spin: a peer that never
//               arrives sets the error word instead of hanging the device), then a device-local grid barrier (every push of
//               this rank has READ `data` before anybody overwrites it)
//     3. reduce its 1 / (G C) share: data[i] = (inbox[p][0][i] + inbox[p][1][i] + ...) * (1 / G), sources in rank order
//   Parity double-buffering: a rank can be at most one call ahead of a peer (it cannot finish call k + 1 without the peer's push of
//   call k + 1, which the peer issues after its own reduce of call k), so the rows of call k are never overwritten while read.
// Validated here with two processes on ONE device (tests/test_gpu_oneshot_allreduce.py: IPC mapping, flags, parity, graph
// replay, bit-equality with gloo); cross-DEVICE visibility (fine-grained memory over xGMI) cannot be exercised on this pool's
// one-GPU boxes, so data parallelism defaults to RCCL and this path is opt-in (PIME_ONESHOT_ALLREDUCE=1).
#include "pime_common.hpp"

#include <cstring>
#include <vector>

namespace pime {

constexpr int kArMaxWorld = 8;
constexpr int kArChunks = 8;          // workgroups per destination
constexpr int kArThreads = 256;
constexpr int kArFlagStride = 16;     // uint32 words between two flags (64 B)
constexpr unsigned kArSpinLimit = 4000000u;   // x ~0.5 us: about two seconds

struct ArRegion {          // layout of one rank's exported region
    __host__ __device__ static size_t flag_words() { return 2 * kArMaxWorld * kArFlagStride; }
    __host__ __device__ static size_t bytes(long long n) { return flag_words() * 4 + (size_t)2 * kArMaxWorld * n * 4; }
};

struct ArArgs {
    int rank, world;
    long long n;
    unsigned* flags[kArMaxWorld];   // peer d's flag block (d = rank: my own)
    float* inbox[kArMaxWorld];      // peer d's inbox block
    unsigned* local;                // device-local control words: [0] calls of parity 0, [1] calls of parity 1, [2] grid-barrier
                                    // arrivals, [3] grid-barrier generation, [4] error
    float* data;
};

__device__ __forceinline__ unsigned ld_sys(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ __launch_bounds__(kArThreads) void oneshot_allreduce_kernel(ArArgs a) {
    const int G = a.world, C = kArChunks;
    const int d = blockIdx.x / C, c = blockIdx.x % C, tid = threadIdx.x;
    const long long n = a.n;
    // the call's parity: calls so far (both parities) = local[0] + local[1]; every workgroup reads them before anyone updates them
    // (the update happens behind the grid barrier below)
    const unsigned done0 = a.local[0], done1 = a.local[1];
    const int p = (int)((done0 + done1) & 1u);
    const unsigned want = ((p ? done1 : done0) + 1u) * (unsigned)C;
    // 1. push chunk c to destination d
    {
        const long long per = (n + C - 1) / C, lo = c * per, hi = lo + per < n ? lo + per : n;
        float* dst = a.inbox[d] + ((size_t)p * kArMaxWorld + a.rank) * n;
        for (long long i = lo + tid; i < hi; i += kArThreads) dst[i] = a.data[i];
        __threadfence_system();
        __syncthreads();
        if (tid == 0)
            __hip_atomic_fetch_add(a.flags[d] + (p * kArMaxWorld + a.rank) * kArFlagStride, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. wait for every source's C chunks in MY inbox
    __shared__ int failed;
    if (tid == 0) failed = 0;
    __syncthreads();
    if (tid < G) {
        const unsigned* f = a.flags[a.rank] + (p * kArMaxWorld + tid) * kArFlagStride;
        unsigned spins = 0;
        while (ld_sys(f) < want) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > kArSpinLimit) { failed = 1; break; }
        }
    }
    __syncthreads();
    if (failed) {
        if (tid == 0) atomicExch(a.local + 4, 1u);
        // fall through: the grid barrier below must still be reached by every workgroup
    }
    // device-local grid barrier (G * C <= 64 workgroups of 256 threads: all resident): every push has read `data`
    if (tid == 0) {
        const unsigned gen = __hip_atomic_load(a.local + 3, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(a.local + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            a.local[2] = 0;
            a.local[p] = (p ? done1 : done0) + 1u;    // this call is counted
            __hip_atomic_store(a.local + 3, gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(a.local + 3, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > kArSpinLimit) { atomicExch(a.local + 4, 2u); break; }
            }
        }
    }
    __syncthreads();
    __threadfence_system();   // acquire side of the flags for every thread of the workgroup
    if (failed) return;
    // 3. reduce this workgroup's share, sources in rank order (the same order on every rank: bit-identical results)
    {
        const long long W = (long long)gridDim.x, per = (n + W - 1) / W, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
        const float* base = a.inbox[a.rank] + (size_t)p * kArMaxWorld * n;
        const float inv = 1.0f / (float)G;
        for (long long i = lo + tid; i < hi; i += kArThreads) {
            float s = __builtin_nontemporal_load(base + i);
            for (int r = 1; r < G; ++r) s += __builtin_nontemporal_load(base + (size_t)r * n + i);
            a.data[i] = s * inv;
        }
    }
}

}  // namespace pime

using namespace pime;

struct pime_oneshot {
    int rank = 0, world = 1, device = 0;
    long long n = 0;
    void* region = nullptr;            // my exported region
    void* peer[kArMaxWorld] = {};      // opened peer regions (peer[rank] = region)
    unsigned* local = nullptr;
    bool connected = false;
    bool fine_grained = false;         // region obtained with hipDeviceMallocFinegrained (coherent for peers on OTHER devices while kernels run)
};

extern "C" {

pime_oneshot* pime_oneshot_create(int32_t rank, int32_t world, int64_t n_floats, int32_t device) {
    if (rank < 0 || world < 1 || world > kArMaxWorld || rank >= world || n_floats < 1) {
        set_error("pime_oneshot_create: rank %d / world %d (max %d) / n %lld", rank, world, kArMaxWorld, (long long)n_floats);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { set_error("pime_oneshot_create: hipSetDevice(%d) failed", device); return nullptr; }
    auto* h = new pime_oneshot();
    h->rank = rank; h->world = world; h->device = device; h->n = n_floats;
    const size_t bytes = ArRegion::bytes(n_floats);
    // fine-grained memory: peer writes and system-scope atomics are coherent while the kernels run; plain hipMalloc (coarse-grained)
    // if the runtime refuses (still correct between processes that share one device and its L2)
    hipError_t e = hipExtMallocWithFlags(&h->region, bytes, hipDeviceMallocFinegrained);
    h->fine_grained = e == hipSuccess;
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&h->region, bytes); }   // recorded: pime_oneshot_info, and refused across devices
    if (e != hipSuccess || hipMalloc(reinterpret_cast<void**>(&h->local), 8 * sizeof(unsigned)) != hipSuccess) {
        set_error("pime_oneshot_create: allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        delete h;
        return nullptr;
    }
    (void)hipMemset(h->region, 0, bytes);
    (void)hipMemset(h->local, 0, 8 * sizeof(unsigned));
    (void)hipDeviceSynchronize();
    h->peer[rank] = h->region;
    return h;
}

int pime_oneshot_export(pime_oneshot* h, void* handle_out) {
    PIME_REQUIRE(h && handle_out, "pime_oneshot_export: NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    PIME_HIP_TRY(hipSetDevice(h->device));
    hipIpcMemHandle_t ipc;
    if (hipIpcGetMemHandle(&ipc, h->region) != hipSuccess) {   // a runtime that does not export fine-grained allocations: coarse-grained
        (void)hipGetLastError();
        const size_t bytes = ArRegion::bytes(h->n);
        (void)hipFree(h->region);
        h->region = nullptr;
        h->fine_grained = false;
        PIME_HIP_TRY(hipMalloc(&h->region, bytes));
        PIME_HIP_TRY(hipMemset(h->region, 0, bytes));
        PIME_HIP_TRY(hipDeviceSynchronize());
        h->peer[h->rank] = h->region;
        PIME_HIP_TRY(hipIpcGetMemHandle(&ipc, h->region));
    }
    std::memcpy(handle_out, &ipc, sizeof(ipc));
    return PIME_OK;
}

int pime_oneshot_connect(pime_oneshot* h, const void* handles) {
    PIME_REQUIRE(h && handles, "pime_oneshot_connect: NULL argument");
    PIME_HIP_TRY(hipSetDevice(h->device));
    for (int r = 0; r < h->world; ++r) {
        if (r == h->rank) continue;
        hipIpcMemHandle_t ipc;
        std::memcpy(&ipc, static_cast<const char*>(handles) + 64 * r, sizeof(ipc));
        PIME_HIP_TRY(hipIpcOpenMemHandle(&h->peer[r], ipc, hipIpcMemLazyEnablePeerAccess));
    }
    h->connected = true;
    return PIME_OK;
}

int pime_oneshot_allreduce_mean(pime_oneshot* h, float* data, pime_stream stream) {
    PIME_REQUIRE(h && data, "pime_oneshot_allreduce_mean: NULL argument");
    PIME_REQUIRE(h->connected || h->world == 1, "pime_oneshot_allreduce_mean before pime_oneshot_connect");
    ArArgs a{};
    a.rank = h->rank; a.world = h->world; a.n = h->n; a.local = h->local; a.data = data;
    for (int r = 0; r < h->world; ++r) {
        a.flags[r] = static_cast<unsigned*>(h->peer[r]);
        a.inbox[r] = reinterpret_cast<float*>(static_cast<char*>(h->peer[r]) + ArRegion::flag_words() * 4);
    }
    hipLaunchKernelGGL(oneshot_allreduce_kernel, dim3(h->world * kArChunks), dim3(kArThreads), 0, static_cast<hipStream_t>(stream), a);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

/* info[0] = 1 if the exported region is fine-grained device memory (coherent for peers on other devices while kernels run), 0 if the
 * runtime only gave coarse-grained memory (coherent between processes that share ONE device and its L2 -- wrong means, silently, across
 * devices); info[1] = device ordinal; info[2..4] = PCI domain, bus, device of that GPU (ranks compare them to see whether they share it). */
int pime_oneshot_info(pime_oneshot* h, int32_t* info) {
    PIME_REQUIRE(h && info, "pime_oneshot_info: NULL argument");
    int dom = 0, bus = 0, dev = 0;
    PIME_HIP_TRY(hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, h->device));
    PIME_HIP_TRY(hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, h->device));
    PIME_HIP_TRY(hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, h->device));
    info[0] = h->fine_grained ? 1 : 0; info[1] = h->device; info[2] = dom; info[3] = bus; info[4] = dev;
    return PIME_OK;
}

/* 0 = every call so far completed; 1 = a peer's data did not arrive within the spin limit; 2 = the local grid barrier timed out.
 * Synchronises the device. */
int pime_oneshot_status(pime_oneshot* h) {
    PIME_REQUIRE(h != nullptr, "pime_oneshot_status: NULL handle");
    PIME_HIP_TRY(hipSetDevice(h->device));
    unsigned w[8];
    PIME_HIP_TRY(hipMemcpy(w, h->local, sizeof(w), hipMemcpyDeviceToHost));
    return (int)w[4];
}

void pime_oneshot_destroy(pime_oneshot* h) {
    if (!h) return;
    // (parked while a stream capture is open: pime_common.hpp release_device)
    for (int r = 0; r < h->world; ++r)
        if (r != h->rank && h->peer[r]) release_device(h->peer[r], h->device, RELEASE_IPC_CLOSE, false);
    release_device(h->region, h->device, RELEASE_FREE, true);
    release_device(h->local, h->device, RELEASE_FREE, false);
    delete h;
    (void)drain_releases();
}

}  // extern "C"
