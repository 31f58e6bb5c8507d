// HIP kernels for the vectorised pH-neutralisation and two-tank water-level envs (gfx950).
//
// One thread advances one env instance; state is SoA in HBM so every load/store of a wave is one
// contiguous 256-B (f32) or 512-B (f64) segment.  Both kernels are HBM/latency bound (76 B and
// 84-112 B of algorithmic traffic per env-step); the only non-streaming access is the 4/8-B titration-LUT
// gather, which is served by the XCD L2 (the reachable part of the table is 300-600 KB).
//
// Reference semantics (file:line under /root/reference):
//   pH step   gym_control/envs/ph.py:320-348 (+ :155-159 action map, :187-189 LUT, :202-225 reward), NoBound :448-478,
//             TimeLimit from gym_control/__init__.py:6
//   pH reset  ph.py:409-445, ZOH :114-121
//   WT step   gym_control/envs/nonlinear_watertank.py:800-826 (+ :258-260, :271-272, :484-514), Stacking :1118-1147
//   WT reset  :890-939, :1166-1208
//   residual  elegantrl/agent_residual.py:61
#include "env_device.hpp"

namespace pime {

// ============================================================================================ pH
// (lane arithmetic in env_device.hpp; the kernels below are load -> step/reset -> store)
// Template parameters of every kernel below: S = arithmetic / storage type of the slow state words, SI = storage type of the
// integrated error (S, or binary16 in PIME_STATE_MIXED16), OT = type of the observation / reward buffers (float, or binary16
// through the *_h entry points: SURVEY §8(d) cfg 5 "fp16 storage for obs / reward / I, f32 math, f64 x").
template <typename S, typename SI, typename OT>
__global__ void ph_reset_kernel(PhParams p, PhPtrs<S, SI> st, const uint8_t* __restrict__ mask,
                                const double* __restrict__ draws, OT* __restrict__ obs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    if (mask && !mask[i]) return;
    PhLane<S> L;
    ph_lane_load<S>(p, st, i, L);
    float o[3];
    ph_lane_reset<S>(p, st.table, p.env_offset + (uint32_t)i, draws ? draws + 4 * (size_t)i : nullptr, L, o);
    ph_lane_store<S>(p, st, i, L);
    obs[3 * (size_t)i + 0] = (OT)o[0]; obs[3 * (size_t)i + 1] = (OT)o[1]; obs[3 * (size_t)i + 2] = (OT)o[2];
}

template <typename S, typename SI, typename ActT, typename OT, bool RESIDUAL>
__global__ void ph_step_kernel(PhParams p, PhPtrs<S, SI> st, const ActT* __restrict__ act,
                               const OT* __restrict__ obs_in, PriorK K, const double* __restrict__ reset_draws,
                               OT* __restrict__ obs, OT* __restrict__ reward, uint8_t* __restrict__ done) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    double a;
    if constexpr (RESIDUAL) {
        const float o_in[3] = {(float)obs_in[3 * (size_t)i], (float)obs_in[3 * (size_t)i + 1], (float)obs_in[3 * (size_t)i + 2]};
        a = ph_residual_action((float)act[i], o_in, K);
    } else {
        a = (double)act[i];
    }
    PhLane<S> L;
    ph_lane_load<S>(p, st, i, L);
    float o[3], rew;
    const bool d = ph_lane_step<S>(p, st.table, a, L, o, rew);
    reward[i] = (OT)rew;
    done[i] = (uint8_t)d;
    if (d && p.auto_reset)
        ph_lane_reset<S>(p, st.table, p.env_offset + (uint32_t)i, reset_draws ? reset_draws + 4 * (size_t)i : nullptr, L, o);
    ph_lane_store<S>(p, st, i, L);
    obs[3 * (size_t)i + 0] = (OT)o[0]; obs[3 * (size_t)i + 1] = (OT)o[1]; obs[3 * (size_t)i + 2] = (OT)o[2];
}

template <typename S, typename SI, typename OT>
__global__ void ph_observe_kernel(PhParams p, PhPtrs<S, SI> st, OT* __restrict__ obs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    obs[3 * (size_t)i + 0] = (OT)(float)ph_lookup<S>(p, st.table, st.C[i], st.x[i]);
    obs[3 * (size_t)i + 1] = (OT)(float)st.r[i];
    obs[3 * (size_t)i + 2] = (OT)(float)st.I[i];
}

// ============================================================================================ water tank
// Stacking variant (nonlinear_watertank.py:1056-1208): frame ring in SoA order with a per-lane head (env_state.hpp).
template <typename S, typename SI>
__device__ __forceinline__ void wt_frames_fill(const WtParams& p, const WtPtrs<S, SI>& st, int i, const WtLane<S>& L) {
    const size_t n = (size_t)p.n;   // every frame = the first frame (:1181-1183)
    for (int s = 0; s < p.num_stack; ++s) {
        st.frames[(3 * s + 0) * n + i] = L.h1; st.frames[(3 * s + 1) * n + i] = L.h2; st.frames[(3 * s + 2) * n + i] = L.r;
    }
    st.head[i] = 0;
}
template <typename S, typename SI>
__device__ __forceinline__ void wt_frames_push(const WtParams& p, const WtPtrs<S, SI>& st, int i, const WtLane<S>& L) {
    const size_t n = (size_t)p.n;   // deque(maxlen=S).append([h1,h2,r]) (:1143-1144): overwrite the oldest slot
    const int h = st.head[i];
    st.frames[(3 * h + 0) * n + i] = L.h1; st.frames[(3 * h + 1) * n + i] = L.h2; st.frames[(3 * h + 2) * n + i] = L.r;
    st.head[i] = h + 1 == p.num_stack ? 0 : h + 1;
}

// Observation write-out; EVERY thread of the block calls it (`live` = this lane writes a row).
//   Integrator: [h1, h2, r, I] (:789-793), one float4 per lane.
//   Stacking:   np.array(frames).reshape(1,-1)[0], oldest first (:1162-1164) = 3S floats per lane.  Written lane by lane
//               that is a 12 S-byte stride between neighbouring lanes; instead the block's rows are assembled in LDS
//               ([blockDim][3S] + a live flag per row) and leave as one contiguous, coalesced run.
template <typename S, typename SI, typename OT>
__device__ __forceinline__ void wt_write_obs(const WtParams& p, const WtPtrs<S, SI>& st, int i, bool live, S h1, S h2, S r, S I,
                                             OT* __restrict__ obs, float* __restrict__ lds) {
    if (p.num_stack > 0) {
        const int D = p.obs_dim;
        int* flag = reinterpret_cast<int*>(lds + blockDim.x * D);
        flag[threadIdx.x] = live ? 1 : 0;
        if (live) {
            const size_t n = (size_t)p.n;
            int slot = st.head[i];
            float* row = lds + threadIdx.x * D;
            for (int j = 0; j < p.num_stack; ++j) {
                row[3 * j + 0] = (float)st.frames[(3 * slot + 0) * n + i];
                row[3 * j + 1] = (float)st.frames[(3 * slot + 1) * n + i];
                row[3 * j + 2] = (float)st.frames[(3 * slot + 2) * n + i];
                slot = slot + 1 == p.num_stack ? 0 : slot + 1;
            }
        }
        __syncthreads();
        const size_t base = (size_t)blockIdx.x * blockDim.x * D;
        for (int e = threadIdx.x; e < (int)blockDim.x * D; e += blockDim.x)
            if (flag[e / D]) obs[base + e] = (OT)lds[e];
    } else if (live) {
        OT* o = obs + 4 * (size_t)i;
        o[0] = (OT)(float)h1; o[1] = (OT)(float)h2; o[2] = (OT)(float)r; o[3] = (OT)(float)I;   // one 16-B (8-B) store per lane
    }
}

template <typename S, typename SI, typename OT>
__global__ void wt_reset_kernel(WtParams p, WtPtrs<S, SI> st, const uint8_t* __restrict__ mask,
                                const double* __restrict__ draws, OT* __restrict__ obs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < p.n && (!mask || mask[i]);
    WtLane<S> L{};
    if (live) {
        wt_lane_load<S>(p, st, i, L);
        wt_lane_reset<S>(p, p.env_offset + (uint32_t)i, draws ? draws + 6 * (size_t)i : nullptr, L);
        wt_lane_store<S>(p, st, i, L);
        if (p.num_stack > 0) wt_frames_fill<S>(p, st, i, L);
    }
    wt_write_obs<S>(p, st, i, live, L.h1, L.h2, L.r, L.I, obs, lds);
}

template <typename S, typename SI, typename ActT, typename OT, bool RESIDUAL>
__global__ void wt_step_kernel(WtParams p, WtPtrs<S, SI> st, const ActT* __restrict__ act,
                               const OT* __restrict__ obs_in, PriorK K, const double* __restrict__ noise,
                               const double* __restrict__ reset_draws, OT* __restrict__ obs,
                               OT* __restrict__ reward, uint8_t* __restrict__ done) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < p.n;
    WtLane<S> L{};
    if (live) {
        double a;
        if constexpr (RESIDUAL) {  // agent_residual.py:61
            double dot = 0.0;
            for (int j = 0; j < p.obs_dim; ++j) dot += (double)(float)obs_in[(size_t)i * p.obs_dim + j] * K.k[j];
            a = residual_tanh((float)act[i]) + dot;
        } else {
            a = (double)act[i];
        }
        const uint32_t gid = p.env_offset + (uint32_t)i;
        wt_lane_load<S>(p, st, i, L);
        double z1n, z2n;
        wt_lane_noise<S>(p, gid, L, noise ? noise + 2 * (size_t)i : nullptr, z1n, z2n);
        float rew;
        const bool d = wt_lane_step<S>(p, a, z1n, z2n, L, rew);
        reward[i] = (OT)rew;
        done[i] = (uint8_t)d;
        if (d && p.auto_reset) {
            wt_lane_reset<S>(p, gid, reset_draws ? reset_draws + 6 * (size_t)i : nullptr, L);
            wt_lane_store<S>(p, st, i, L);
            if (p.num_stack > 0) wt_frames_fill<S>(p, st, i, L);
        } else {
            wt_lane_store<S>(p, st, i, L);
            if (p.num_stack > 0) wt_frames_push<S>(p, st, i, L);
        }
    }
    wt_write_obs<S>(p, st, i, live, L.h1, L.h2, L.r, L.I, obs, lds);
}

template <typename S, typename SI, typename OT>
__global__ void wt_observe_kernel(WtParams p, WtPtrs<S, SI> st, OT* __restrict__ obs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < p.n;
    const int k = live ? i : 0;
    wt_write_obs<S>(p, st, i, live, st.h1[k], st.h2[k], st.r[k], p.num_stack > 0 ? S(0) : (S)st.I[k], obs, lds);
}

// ============================================================================================ launchers
// LDS of the Stacking observation staging: [block][3S] floats + one live flag per row
static inline size_t wt_obs_lds_bytes(const WtParams& p, int block) {
    return p.num_stack > 0 ? sizeof(float) * (size_t)block * (p.obs_dim + 1) : 0;
}

static inline dim3 lane_grid(int n, int& block) {
    // one wave per workgroup while the launch is small, so a 16 384-env launch still spreads over all 256 CUs
    block = n <= 65536 ? 64 : 256;
    return dim3((unsigned)((n + block - 1) / block));
}

template <typename S, typename SI, typename OT>
int launch_ph_reset(const PhParams& p, const PhPtrs<S, SI>& st, const uint8_t* mask, const double* draws, OT* obs,
                    hipStream_t s) {
    int block;
    const dim3 grid = lane_grid(p.n, block);
    hipLaunchKernelGGL((ph_reset_kernel<S, SI, OT>), grid, dim3(block), 0, s, p, st, mask, draws, obs);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

template <typename S, typename SI, typename OT>
int launch_ph_step(const PhParams& p, const PhPtrs<S, SI>& st, const void* act, int act_dtype, bool residual,
                   const OT* obs_in, const PriorK& K, const double* reset_draws, OT* obs, OT* reward,
                   uint8_t* done, hipStream_t s) {
    int block;
    const dim3 grid = lane_grid(p.n, block);
    if (residual)
        hipLaunchKernelGGL((ph_step_kernel<S, SI, float, OT, true>), grid, dim3(block), 0, s, p, st, (const float*)act, obs_in,
                           K, reset_draws, obs, reward, done);
    else if (act_dtype == PIME_F32)
        hipLaunchKernelGGL((ph_step_kernel<S, SI, float, OT, false>), grid, dim3(block), 0, s, p, st, (const float*)act,
                           obs_in, K, reset_draws, obs, reward, done);
    else
        hipLaunchKernelGGL((ph_step_kernel<S, SI, double, OT, false>), grid, dim3(block), 0, s, p, st, (const double*)act,
                           obs_in, K, reset_draws, obs, reward, done);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

template <typename S, typename SI, typename OT>
int launch_ph_observe(const PhParams& p, const PhPtrs<S, SI>& st, OT* obs, hipStream_t s) {
    int block;
    const dim3 grid = lane_grid(p.n, block);
    hipLaunchKernelGGL((ph_observe_kernel<S, SI, OT>), grid, dim3(block), 0, s, p, st, obs);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

template <typename S, typename SI, typename OT>
int launch_wt_reset(const WtParams& p, const WtPtrs<S, SI>& st, const uint8_t* mask, const double* draws, OT* obs,
                    hipStream_t s) {
    int block;
    const dim3 grid = lane_grid(p.n, block);
    hipLaunchKernelGGL((wt_reset_kernel<S, SI, OT>), grid, dim3(block), wt_obs_lds_bytes(p, block), s, p, st, mask, draws, obs);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

template <typename S, typename SI, typename OT>
int launch_wt_step(const WtParams& p, const WtPtrs<S, SI>& st, const void* act, int act_dtype, bool residual,
                   const OT* obs_in, const PriorK& K, const double* noise, const double* reset_draws, OT* obs,
                   OT* reward, uint8_t* done, hipStream_t s) {
    int block;
    const dim3 grid = lane_grid(p.n, block);
    if (residual)
        hipLaunchKernelGGL((wt_step_kernel<S, SI, float, OT, true>), grid, dim3(block), wt_obs_lds_bytes(p, block), s, p, st,
                           (const float*)act, obs_in, K, noise, reset_draws, obs, reward, done);
    else if (act_dtype == PIME_F32)
        hipLaunchKernelGGL((wt_step_kernel<S, SI, float, OT, false>), grid, dim3(block), wt_obs_lds_bytes(p, block), s, p, st,
                           (const float*)act, obs_in, K, noise, reset_draws, obs, reward, done);
    else
        hipLaunchKernelGGL((wt_step_kernel<S, SI, double, OT, false>), grid, dim3(block), wt_obs_lds_bytes(p, block), s, p, st,
                           (const double*)act, obs_in, K, noise, reset_draws, obs, reward, done);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

template <typename S, typename SI, typename OT>
int launch_wt_observe(const WtParams& p, const WtPtrs<S, SI>& st, OT* obs, hipStream_t s) {
    int block;
    const dim3 grid = lane_grid(p.n, block);
    hipLaunchKernelGGL((wt_observe_kernel<S, SI, OT>), grid, dim3(block), wt_obs_lds_bytes(p, block), s, p, st, obs);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

// explicit instantiations used by abi.hip: (S, SI, OT) = f64 state, f32 state, f32 state + binary16 I with float / binary16 outputs
#define PIME_INST(S, SI, OT)                                                                                           \
    template int launch_ph_reset<S, SI, OT>(const PhParams&, const PhPtrs<S, SI>&, const uint8_t*, const double*, OT*, \
                                            hipStream_t);                                                              \
    template int launch_ph_step<S, SI, OT>(const PhParams&, const PhPtrs<S, SI>&, const void*, int, bool, const OT*,   \
                                           const PriorK&, const double*, OT*, OT*, uint8_t*, hipStream_t);             \
    template int launch_ph_observe<S, SI, OT>(const PhParams&, const PhPtrs<S, SI>&, OT*, hipStream_t);                \
    template int launch_wt_reset<S, SI, OT>(const WtParams&, const WtPtrs<S, SI>&, const uint8_t*, const double*, OT*, \
                                            hipStream_t);                                                              \
    template int launch_wt_step<S, SI, OT>(const WtParams&, const WtPtrs<S, SI>&, const void*, int, bool, const OT*,   \
                                           const PriorK&, const double*, const double*, OT*, OT*, uint8_t*,            \
                                           hipStream_t);                                                               \
    template int launch_wt_observe<S, SI, OT>(const WtParams&, const WtPtrs<S, SI>&, OT*, hipStream_t);
PIME_INST(double, double, float)
PIME_INST(float, float, float)
PIME_INST(float, half_t, float)
PIME_INST(float, half_t, half_t)

}  // namespace pime
