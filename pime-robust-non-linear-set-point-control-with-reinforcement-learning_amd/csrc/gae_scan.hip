// Reverse-time reward-sum / advantage recursion over a [T, N] time-major trajectory buffer (gfx950).
//
// replaces: AgentPPO.compute_reward_gae / compute_reward_adv, /root/reference/elegantrl/agent.py:666-708 --
// a Python loop issuing two scalar tensor writes per transition.  Here one thread owns one env lane and scans
// its T steps backwards; at every t the wave reads three contiguous 256-B rows (reward, mask, value) and writes
// two, so the kernel streams 20 B per transition.  The recurrence is sequential in t but the loads are not:
// they are issued CHUNK steps ahead of the arithmetic so each lane keeps CHUNK*3 loads in flight.
#include "pime_common.hpp"

namespace pime {

constexpr int kGaeChunk = 10;

template <bool USE_GAE>
__global__ void gae_scan_kernel(const float* __restrict__ reward, const float* __restrict__ mask,
                                const float* __restrict__ value, int T, int N, float lambda,
                                float* __restrict__ r_sum, float* __restrict__ adv) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float pre_r = 0.f, pre_a = 0.f;
    for (int t1 = T; t1 > 0; t1 -= kGaeChunk) {
        const int t0 = t1 - kGaeChunk > 0 ? t1 - kGaeChunk : 0;
        float rw[kGaeChunk], mk[kGaeChunk], vl[kGaeChunk];
#pragma unroll
        for (int j = 0; j < kGaeChunk; ++j) {
            const int t = t1 - 1 - j;
            if (t >= t0) {
                const size_t k = (size_t)t * N + n;
                rw[j] = reward[k]; mk[j] = mask[k]; vl[j] = value[k];
            }
        }
#pragma unroll
        for (int j = 0; j < kGaeChunk; ++j) {
            const int t = t1 - 1 - j;
            if (t >= t0) {
                const size_t k = (size_t)t * N + n;
                const float rs = rw[j] + mk[j] * pre_r;            // agent.py:701
                pre_r = rs;
                r_sum[k] = rs;
                float a;
                if constexpr (USE_GAE) {
                    a = rw[j] + mk[j] * (pre_a - vl[j]);           // :704
                    pre_a = vl[j] + a * lambda;                    // :705
                } else {
                    a = rs - mk[j] * vl[j];                        // :681
                }
                adv[k] = a;
            }
        }
    }
}

int launch_gae_scan(const float* reward, const float* mask, const float* value, int T, int N, float lambda,
                    int use_gae, float* r_sum, float* adv, hipStream_t s) {
    const int block = N <= 65536 ? 64 : 256;
    const dim3 grid((unsigned)((N + block - 1) / block));
    if (use_gae)
        hipLaunchKernelGGL(gae_scan_kernel<true>, grid, dim3(block), 0, s, reward, mask, value, T, N, lambda, r_sum, adv);
    else
        hipLaunchKernelGGL(gae_scan_kernel<false>, grid, dim3(block), 0, s, reward, mask, value, T, N, lambda, r_sum, adv);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

}  // namespace pime
