// Host-side pieces of an env handle shared by the GPU library (abi.hip) and its CPU twin (cpu_twins.hip): the configuration check,
// the SoA carving of the state slab (identical layout in HBM and in host memory) and the per-launch parameter blocks.
#pragma once
#include "env_state.hpp"

namespace pime {

struct Carver {
    size_t off = 0;
    char* base = nullptr;
    template <typename T>
    T* take(size_t count) {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

template <typename S, typename SI>
inline void carve_ph(Carver& c, PhPtrs<S, SI>& p, int n, int table_len, S** table_out) {
    p.x = c.take<double>(n); p.A = c.take<double>(n); p.B = c.take<double>(n); p.C = c.take<double>(n);
    p.qww = c.take<double>(n); p.qc = c.take<double>(n);
    p.I = c.take<SI>(n); p.r = c.take<S>(n); p.last_a = c.take<S>(n);
    p.t = c.take<int32_t>(n); p.episode = c.take<int32_t>(n);
    *table_out = c.take<S>(table_len);
    p.table = *table_out;
}

template <typename S, typename SI>
inline void carve_wt(Carver& c, WtPtrs<S, SI>& p, int n, int obs_dim, int num_stack) {
    p.h1 = c.take<S>(n); p.h2 = c.take<S>(n); p.r = c.take<S>(n); p.I = c.take<SI>(n);
    p.a1 = c.take<S>(n); p.a2 = c.take<S>(n); p.kp = c.take<S>(n);
    p.frames = num_stack > 0 ? c.take<S>((size_t)n * obs_dim) : nullptr;
    p.head = num_stack > 0 ? c.take<int32_t>(n) : nullptr;
    p.t = c.take<int32_t>(n); p.episode = c.take<int32_t>(n);
}

inline int check_cfg(const pime_env_cfg* c) {
    PIME_REQUIRE(c != nullptr, "cfg is NULL");
    PIME_REQUIRE(c->kind == PIME_ENV_PH || c->kind == PIME_ENV_WT, "unknown env kind %d", c->kind);
    PIME_REQUIRE(c->n_envs >= 1, "n_envs = %d", c->n_envs);
    PIME_REQUIRE(c->state_mode == PIME_STATE_F64 || c->state_mode == PIME_STATE_MIXED || c->state_mode == PIME_STATE_MIXED16,
                 "unknown state_mode %d", c->state_mode);
    PIME_REQUIRE(c->reward_type >= PIME_REWARD_DISTANCE && c->reward_type <= PIME_REWARD_SPARSE, "unknown reward_type %d",
                 c->reward_type);
    PIME_REQUIRE(c->max_steps >= 1, "max_steps = %d", c->max_steps);
    PIME_REQUIRE(c->resample_every >= 0, "resample_every = %d", c->resample_every);
    if (c->kind == PIME_ENV_PH) {
        PIME_REQUIRE(c->ph_table != nullptr && c->ph_table_len >= 2, "pH env needs the titration table (pime_ph_table_build)");
        PIME_REQUIRE(c->num_stack == 0, "num_stack applies to the water-tank env only");
    } else {
        PIME_REQUIRE(c->num_stack >= 0 && 3 * c->num_stack <= kMaxObsDim, "num_stack = %d out of range", c->num_stack);
        PIME_REQUIRE(c->wt_n_discrete >= 1, "wt_n_discrete = %d", c->wt_n_discrete);
    }
    return PIME_OK;
}

inline void fill_params(const pime_env_cfg& c, int obs_dim, PhParams& ph, WtParams& wt) {
    if (c.kind == PIME_ENV_PH) {
        PhParams& p = ph;
        p.n = c.n_envs; p.max_steps = c.max_steps; p.reward_type = c.reward_type; p.integral_bound = c.integral_bound;
        p.resample_every = c.resample_every; p.table_len = c.ph_table_len; p.auto_reset = 0;
        p.has_punish = (c.integral_punish != 0.0 || c.action_punish != 0.0 || c.action_change_punish != 0.0);
        p.env_offset = c.env_offset; p.seed = c.seed;
        p.integral_max = c.integral_max; p.integral_punish = c.integral_punish; p.action_punish = c.action_punish;
        p.action_change_punish = c.action_change_punish; p.thr = c.distance_threshold;
        p.sample_t = c.ph_sample_t; p.u_low = c.ph_u_low; p.u_high = c.ph_u_high; p.table_scale = c.ph_table_scale;
        p.qww_lo = c.range_lo[0]; p.qww_hi = c.range_hi[0]; p.qc_lo = c.range_lo[1]; p.qc_hi = c.range_hi[1];
        p.x0_lo = c.init_lo[0]; p.x0_hi = c.init_hi[0]; p.r_lo = c.init_lo[1]; p.r_hi = c.init_hi[1];
    } else {
        WtParams& p = wt;
        p.n = c.n_envs; p.max_steps = c.max_steps; p.reward_type = c.reward_type; p.num_stack = c.num_stack;
        p.resample_every = c.resample_every; p.n_discrete = c.wt_n_discrete; p.auto_reset = 0; p.obs_dim = obs_dim;
        p.env_offset = c.env_offset; p.seed = c.seed;
        p.integral_max = c.integral_max; p.integral_punish = c.integral_punish; p.thr = c.distance_threshold;
        p.A1 = c.wt_A1; p.A2 = c.wt_A2; p.G = c.wt_G; p.dt = c.wt_dt; p.noise_scale = c.wt_noise_scale; p.z1 = c.wt_z1;
        p.pmax = c.wt_pmax;
        p.a1_lo = c.range_lo[0]; p.a1_hi = c.range_hi[0]; p.a2_lo = c.range_lo[1]; p.a2_hi = c.range_hi[1];
        p.kp_lo = c.range_lo[2]; p.kp_hi = c.range_hi[2];
        p.h_lo = c.init_lo[0]; p.h_hi = c.init_hi[0]; p.r_lo = c.init_lo[1]; p.r_hi = c.init_hi[1];
    }
}


}  // namespace pime
