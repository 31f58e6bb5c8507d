// Internal helpers shared by the HIP translation units of libpime_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>

#include "pime_hip.h"

namespace pime {

void set_error(const char* fmt, ...);

#define PIME_HIP_TRY(expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            ::pime::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PIME_ERR_DEVICE;                                                                 \
        }                                                                                           \
    } while (0)

#define PIME_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            ::pime::set_error(__VA_ARGS__); \
            return PIME_ERR_ARG;         \
        }                                \
    } while (0)

// ---- frees that must not run under stream capture ---------------------------------------------------------------------------
// hipFree / hipDeviceSynchronize / hipIpcCloseMemHandle called while a stream capture is open on the calling thread abort the
// process (seen once: a garbage-collected env handle finalised inside torch.cuda.graph, gpurun_out/r03b_pytest.log).  The library
// keeps a capture depth (pime_capture_begin / pime_capture_end, set by the Python capture wrapper around EVERY torch.cuda.graph of
// this package; the Python finalisers also ask torch whether the current stream is capturing); while it is non-zero, release_device
// parks the pointer in a queue that the next entry point outside a capture drains.
enum ReleaseKind { RELEASE_FREE = 0, RELEASE_IPC_CLOSE = 1 };
bool capture_active();
void release_device(void* p, int device, ReleaseKind kind, bool synchronize_first);   // now, or queued while a capture is open
int drain_releases();                                                                // frees what is queued (no-op under capture); returns how many
int queued_releases();

// hipFuncSetAttribute acts on the current device: one bit per device ordinal records where a kernel's dynamic-LDS limit has
// been raised, so that a process driving several GPUs raises it on each (one static LdsLimit per kernel instantiation).
struct LdsLimit {
    std::atomic<unsigned long long> raised{0};
};

#define PIME_RAISE_LDS(once, kernel, bytes)                                                                     \
    do {                                                                                                        \
        int _dev = 0;                                                                                           \
        PIME_HIP_TRY(hipGetDevice(&_dev));                                                                      \
        const unsigned long long _bit = 1ull << (_dev & 63);                                                    \
        if (!((once).raised.load(std::memory_order_relaxed) & _bit)) {                                          \
            PIME_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                             \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));        \
            (once).raised.fetch_or(_bit, std::memory_order_relaxed);                                            \
        }                                                                                                       \
    } while (0)

// ---- Philox4x32-10 (Salmon et al., SC'11).  Counter layout: (global env id, episode, slot, stream). -------
enum : uint32_t { STREAM_RESET = 0, STREAM_NOISE = 1 };

struct Philox4 {
    uint32_t v[4];
};

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// 53-bit uniform in [0,1): the two-word construction numpy's random_sample uses.
__host__ __device__ inline double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

__host__ __device__ inline void philox_pair(uint64_t seed, uint32_t env, uint32_t episode, uint32_t slot,
                                            uint32_t stream, double& ua, double& ub) {
    const Philox4 o = philox4x32_10(env, episode, slot, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    ua = u53(o.v[0], o.v[1]);
    ub = u53(o.v[2], o.v[3]);
}

constexpr int kMaxObsDim = 32;  // Stacking10 -> 30 floats

struct PriorK {
    double k[kMaxObsDim];
};

}  // namespace pime
