// Register-resident form of one pH env lane: the step / reset arithmetic shared by the per-step kernels
// (env_kernels.hip) and the fused multi-step rollout kernel (rollout.hip), so both run literally the same code.
// Reference semantics: /root/reference/gym_control/envs/ph.py:320-348 (step), :448-478 (NoBound), :409-445 (reset),
// :114-121 (ZOH), :187-189 (LUT), gym TimeLimit.
#pragma once
// Every lane function is __host__ __device__: the kernels (env_kernels.hip, the fused rollouts) run them on the GPU, and
// cpu_twins.hip compiles the SAME source for the host into libpime_cpu.so (float64 state), the CPU twin of SURVEY.md section 8(b).
#include "env_state.hpp"

namespace pime {

template <typename T>
__host__ __device__ __forceinline__ T clip(T v, T lo, T hi) {  // np.clip = minimum(maximum(v, lo), hi)
    const T m = v > lo ? v : lo;
    return m < hi ? m : hi;
}

template <typename S>
__host__ __device__ __forceinline__ S reward_of(int reward_type, S achieved, S goal, S thr) {
    const S d = fabs(achieved - goal);
    if (reward_type == PIME_REWARD_DISTANCE) return -d;
    if (reward_type == PIME_REWARD_SQUARE) return -(d * d);
    return d > thr ? S(-1) : S(-0.0);
}

template <typename S>
__host__ __device__ __forceinline__ S ph_lookup(const PhParams& p, const S* __restrict__ table, double C, double x) {
    // observe_state (ph.py:187-189): first i with MHCl[i] >= around(C*x, 5)  ==  rint(C*x*1e5)  
    long long k = (long long)rint(C * x * p.table_scale);  // round-half-even like np.around (v_rndne_f64 on the device, rint on the host)
    k = k < 0 ? 0 : (k >= p.table_len ? p.table_len - 1 : k);  // reference: IndexError (unreachable in range)
    return table[k];
}

template <typename S>
struct PhLane {
    double x, A, B, C, qww, qc;
    S I, r, last_a;
    int t, episode;
    bool plant_changed;  // (A,B,C,qww,qc) were rewritten by a resampling reset and must be stored
};

template <typename S, typename SI>
__host__ __device__ __forceinline__ void ph_lane_load(const PhParams& p, const PhPtrs<S, SI>& st, int i, PhLane<S>& L) {
    L.x = st.x[i]; L.A = st.A[i]; L.B = st.B[i]; L.C = st.C[i];
    L.I = (S)st.I[i]; L.r = st.r[i];
    L.t = st.t[i]; L.episode = st.episode[i];
    L.last_a = p.has_punish ? st.last_a[i] : S(0);
    L.qww = L.qc = 0.0;
    L.plant_changed = false;
}

template <typename S, typename SI>
__host__ __device__ __forceinline__ void ph_lane_store(const PhParams& p, const PhPtrs<S, SI>& st, int i, const PhLane<S>& L) {
    st.x[i] = L.x; st.I[i] = (SI)L.I; st.r[i] = L.r; st.t[i] = L.t; st.episode[i] = L.episode;
    if (p.has_punish) st.last_a[i] = L.last_a;
    if (L.plant_changed) {
        st.A[i] = L.A; st.B[i] = L.B; st.C[i] = L.C; st.qww[i] = L.qww; st.qc[i] = L.qc;
    }
}

// reset_all / reset_r (ph.py:412-445).  gid = global lane id (Philox counter word); draws = this lane's 4 injected
// values (qww_V, qc_V, x0, r) or nullptr for in-kernel Philox.
template <typename S>
__host__ __device__ __forceinline__ void ph_lane_reset(const PhParams& p, const S* __restrict__ table, uint32_t gid,
                                              const double* __restrict__ draws, PhLane<S>& L, float (&obs)[3]) {
    const int ep = L.episode + 1;
    L.episode = ep;
    const bool resample = p.resample_every > 0 && (ep % p.resample_every) == 0;
    double qww, qc, x0, r;
    if (draws) {  // seed-for-seed replay of the reference's MT19937 draws (host generated)
        qww = draws[0]; qc = draws[1]; x0 = draws[2]; r = draws[3];
    } else {
        double u0, u1, u2, u3;
        philox_pair(p.seed, gid, (uint32_t)ep, 0, STREAM_RESET, u0, u1);
        philox_pair(p.seed, gid, (uint32_t)ep, 1, STREAM_RESET, u2, u3);
        qww = p.qww_lo + (p.qww_hi - p.qww_lo) * u0;  // np.random.uniform(lo, hi), ph.py:410
        qc = p.qc_lo + (p.qc_hi - p.qc_lo) * u1;
        x0 = p.x0_lo + (p.x0_hi - p.x0_lo) * u2;      // ph.py:420
        r = p.r_lo + (p.r_hi - p.r_lo) * u3;          // ph.py:424
    }
    if (resample) {  // update_system (ph.py:114-121): ZOH of qc_V/(s+qww_V) at T -> closed form
        const double e = -qww * p.sample_t;
        L.qww = qww; L.qc = qc;
        L.A = exp(e);
        L.B = -expm1(e) / qww;
        L.C = qc;
        L.plant_changed = true;
    }
    L.x = x0;
    const S y = ph_lookup<S>(p, table, L.C, x0);
    L.t = 0;
    L.r = (S)r;
    L.I = S(0);
    obs[0] = (float)y; obs[1] = (float)r; obs[2] = 0.0f;
}

// One env step for the (unclipped) env action `a`; no reset.  Returns the TimeLimit done flag.
template <typename S>
__host__ __device__ __forceinline__ bool ph_lane_step(const PhParams& p, const S* __restrict__ table, double a, PhLane<S>& L,
                                             float (&obs)[3], float& reward) {
    a = clip(a, -1.0, 1.0);                                   // ph.py:321
    S delta_u = S(0);
    if (p.has_punish) {
        delta_u = L.t != 0 ? (S)a - L.last_a : S(0);          // :322
        L.last_a = (S)a;
    }
    L.t += 1;                                                 // :325
    const double u = p.u_low + (p.u_high - p.u_low) * ((a - -1.0) / (1.0 - -1.0));  // action(): :155-159
    L.x = L.A * L.x + L.B * u;                                // :330
    const S y = ph_lookup<S>(p, table, L.C, L.x);             // :332
    S rew = reward_of<S>(p.reward_type, y, L.r, (S)p.thr);    // :334
    const S I_raw = L.I + (L.r - y);                          // :339-340
    L.I = p.integral_bound ? clip(I_raw, (S)-p.integral_max, (S)p.integral_max) : I_raw;  // :341 / :470
    if (p.has_punish) {
        rew -= (S)p.action_punish * fabs((S)u);               // :336
        rew -= (S)p.action_change_punish * fabs(delta_u);     // :337
        rew += -(S)p.integral_punish * fabs(p.integral_bound ? I_raw : L.I);  // :343 / :473
    }
    reward = (float)rew;
    obs[0] = (float)y; obs[1] = (float)L.r; obs[2] = (float)L.I;
    return L.t >= p.max_steps;  // gym TimeLimit; the env itself returns False (:348)
}

// np.tanh(action_f32) of the residual composition (agent_residual.py:61): the float64 tanh rounded ONCE to float32.  A float32
// tanh differs between implementations by an ulp (ocml tanhf vs glibc tanhf vs numpy's), which now and then moves C*x*1e5 across
// a rounding boundary of the titration table; the float64 functions agree to ~1e-16 relative, so their float32 roundings differ
// with probability ~1e-9 per call and the device, the oracle and numpy's correctly-rounded float32 tanh all read the same cell.
__host__ __device__ __forceinline__ double residual_tanh(float a_pre) { return (double)(float)tanh((double)a_pre); }

// env action of the residual policy: np.tanh(action_f32) + state_f32 @ priorK_f64 (agent_residual.py:61)
__host__ __device__ __forceinline__ double ph_residual_action(float a_pre, const float (&obs_in)[3], const PriorK& K) {
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) dot += (double)obs_in[j] * K.k[j];
    return residual_tanh(a_pre) + dot;
}

// ============================================================================================ water tank
// Register-resident lane of NonLinearWaterTankChangingParamUniformGoal{Integrator,Stacking}
// (/root/reference/gym_control/envs/nonlinear_watertank.py:800-826, :1118-1147 step; :890-939, :1166-1208 reset).
// The Stacking variant's frame ring stays in memory and is handled by the caller; everything else is here.
template <typename S>
struct WtLane {
    S h1, h2, r, I, a1, a2, kp;
    int t, episode;
    bool plant_changed;
};

template <typename S, typename SI>
__host__ __device__ __forceinline__ void wt_lane_load(const WtParams& p, const WtPtrs<S, SI>& st, int i, WtLane<S>& L) {
    L.h1 = st.h1[i]; L.h2 = st.h2[i]; L.r = st.r[i];
    L.I = p.num_stack == 0 ? (S)st.I[i] : S(0);
    L.a1 = st.a1[i]; L.a2 = st.a2[i]; L.kp = st.kp[i];
    L.t = st.t[i]; L.episode = st.episode[i];
    L.plant_changed = false;
}

template <typename S, typename SI>
__host__ __device__ __forceinline__ void wt_lane_store(const WtParams& p, const WtPtrs<S, SI>& st, int i, const WtLane<S>& L) {
    st.h1[i] = L.h1; st.h2[i] = L.h2; st.r[i] = L.r; st.t[i] = L.t; st.episode[i] = L.episode;
    if (p.num_stack == 0) st.I[i] = (SI)L.I;
    if (L.plant_changed) { st.a1[i] = L.a1; st.a2[i] = L.a2; st.kp[i] = L.kp; }
}

// draws = this lane's 6 injected values (a1, a2, Kp, h1, h2, r) or nullptr for in-kernel Philox
template <typename S>
__host__ __device__ __forceinline__ void wt_lane_reset(const WtParams& p, uint32_t gid, const double* __restrict__ draws,
                                              WtLane<S>& L) {
    const int ep = L.episode + 1;
    L.episode = ep;
    const bool resample = p.resample_every > 0 && (ep % p.resample_every) == 0;
    double v[6];
    if (draws) {
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = draws[j];
    } else {
        double u[6];
#pragma unroll
        for (uint32_t s = 0; s < 3; ++s) philox_pair(p.seed, gid, (uint32_t)ep, s, STREAM_RESET, u[2 * s], u[2 * s + 1]);
        v[0] = p.a1_lo + (p.a1_hi - p.a1_lo) * u[0];  // sample_parameters :890-894
        v[1] = p.a2_lo + (p.a2_hi - p.a2_lo) * u[1];
        v[2] = p.kp_lo + (p.kp_hi - p.kp_lo) * u[2];
        v[3] = p.h_lo + (p.h_hi - p.h_lo) * u[3];     // :912
        v[4] = p.h_lo + (p.h_hi - p.h_lo) * u[4];
        v[5] = p.r_lo + (p.r_hi - p.r_lo) * u[5];     // :913
    }
    if (resample) {
        L.a1 = (S)v[0]; L.a2 = (S)v[1]; L.kp = (S)v[2];
        L.plant_changed = true;
    }
    L.h1 = (S)v[3]; L.h2 = (S)v[4]; L.r = (S)v[5];
    L.t = 0;
    L.I = S(0);
}

// the two process-noise normals of step L.t+1 (already scaled): injected pair or Philox Box-Muller (:271-272,:810-811)
template <typename S>
__host__ __device__ __forceinline__ void wt_lane_noise(const WtParams& p, uint32_t gid, const WtLane<S>& L,
                                              const double* __restrict__ noise, double& z1n, double& z2n) {
    if (noise) {
        z1n = noise[0]; z2n = noise[1];
    } else {
        double ua, ub;
        philox_pair(p.seed, gid, (uint32_t)L.episode, (uint32_t)(L.t + 1), STREAM_NOISE, ua, ub);
        if constexpr (sizeof(S) == 4) {
            // float32 state: Box-Muller on the float32 pipes (v_log / v_sqrt / v_sin / v_cos; the last two take revolutions) --
            // the float64 log + sqrt + sincos of the exact form are several hundred float64 instructions per lane-step, more than
            // the 20 Euler sub-steps they perturb.  Same Philox uniforms; the normals differ from the float64 ones by ~1e-6
            // relative, i.e. ~1e-8 of a level after the 0.01 noise scale: far inside the mode's 2e-4.
            const float u1 = (float)(1.0 - ua);                      // in [2^-53, 1]: log finite
            const float rad = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1)), rev = (float)ub;
            z1n = (double)((float)p.noise_scale * (rad * __builtin_amdgcn_cosf(rev)));
            z2n = (double)((float)p.noise_scale * (rad * __builtin_amdgcn_sinf(rev)));
        } else {
            const double rad = sqrt(-2.0 * log(1.0 - ua)), ang = 6.283185307179586476925286766559 * ub;
            double sn, cs;
            sincos(ang, &sn, &cs);
            z1n = p.noise_scale * (rad * cs);
            z2n = p.noise_scale * (rad * sn);
        }
    }
}

// sqrt of the Euler sub-steps (:805-809).  float64 state: IEEE sqrt (reference precision).  float32 state (PIME_STATE_MIXED*): the
// hardware v_sqrt_f32 (<= 1 ulp) instead of the correctly rounded sqrtf, which hipcc expands into the same instruction plus a
// ~10-instruction Newton / scale fix-up -- 40 of them per env step made the kernel VALU-bound at 0.32 of HBM (profiles/
// r02_m_env_pmc.json).  One ulp of a root is a 6e-8 relative perturbation of a smooth map, inside the mode's stated 2e-4.
template <typename S>
__host__ __device__ __forceinline__ S tank_sqrt(S v) {
    if constexpr (sizeof(S) == 4) return __builtin_amdgcn_sqrtf(v);
    else return sqrt(v);
}

// One env step for env action `a` (NOT clipped, :258-260); returns done (:816-821)
template <typename S>
__host__ __device__ __forceinline__ bool wt_lane_step(const WtParams& p, double a, double z1n, double z2n, WtLane<S>& L,
                                             float& reward) {
    L.t += 1;                                                           // :801
    const S u = (S)(a * p.pmax / 2. + p.pmax / 2.);                     // action_P
    S h1 = L.h1, h2 = L.h2;
    const S A1 = (S)p.A1, A2 = (S)p.A2, G = (S)p.G, dt = (S)p.dt;
    const S lo = S(-0.0), hi = (S)INFINITY;                            // Box(low=-0, high=inf) :252-257
    for (int s = 0; s < p.n_discrete; ++s) {                            // :805-809, both roots from the OLD h1,h2
        const S s1 = tank_sqrt<S>(2 * G * h1), s2 = tank_sqrt<S>(2 * G * h2);
        const S n1 = h1 + (-L.a1 / A1 * s1 + L.kp / A1 * u) * dt;
        const S n2 = h2 + (L.a1 / A2 * s1 - L.a2 / A2 * s2) * dt;
        h1 = clip(n1, lo, hi);
        h2 = clip(n2, lo, hi);
    }
    h1 = clip(h1 + (S)z1n, lo, hi);                                     // :810-813
    h2 = clip(h2 + (S)z2n, lo, hi);
    L.h1 = h1; L.h2 = h2;
    S rew = reward_of<S>(p.reward_type, h2, L.r, (S)p.thr);
    if (p.reward_type != PIME_REWARD_SPARSE) rew = rew * (S)p.z1;       // :506,508
    if (p.num_stack == 0) {
        const S I_raw = L.I + (L.r - h2);                               // :822-823
        rew += -(S)p.integral_punish * fabs(I_raw);                     // :824
        L.I = clip(I_raw, (S)-p.integral_max, (S)p.integral_max);       // :825
    }
    reward = (float)rew;
    return !(L.t < p.max_steps);
}

}  // namespace pime
