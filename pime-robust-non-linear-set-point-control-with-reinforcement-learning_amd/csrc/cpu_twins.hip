// libpime_cpu.so: the CPU twins of the env step / reset entry points (include/pime_cpu.h; SURVEY.md section 8(b): "`*_cpu` twins of
// step / reset operating on host pointers (the CPU baseline)").
//
// NOT part of the product path: libpime_hip.so has no CPU fallback and the pime_amd package never loads this library.  It exists
// so that the CPU baseline beside the GPU number (bench.py: cpu_baseline) and the parity tests can run the PRODUCT'S OWN lane
// arithmetic -- env_device.hpp, the very functions the kernels call, compiled here for the host -- on the box's cores, in the
// reference's precision (float64 state, PIME_STATE_F64 semantics).  replaces, per lane: PH1D...Integrator.step / reset
// (/root/reference/gym_control/envs/ph.py:320-348,409-445) and NonLinearWaterTank...Integrator.step / reset
// (nonlinear_watertank.py:800-826,890-939), as pime_env_step / pime_env_reset do on the GPU.
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "env_device.hpp"
#include "env_handle.hpp"
#include "pime_cpu.h"

namespace pime {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pime

using namespace pime;

struct pime_env_cpu {
    pime_env_cfg cfg{};
    int obs_dim = 0;
    bool was_reset = false;
    std::vector<char> slab;
    std::vector<double> table;
    PhParams ph{};
    WtParams wt{};
    PhPtrs<double> ph64{};
    WtPtrs<double> wt64{};
};

namespace {
// lanes [lo, hi) per thread, contiguous chunks, on a persistent pool (a fork-join of std::threads per call cost more than a
// 16 384-lane step); results do not depend on the thread count (lanes are independent)
class Pool {
  public:
    void run(int threads, const std::function<void(int)>& job) {   // job(t) for t in [0, threads); returns when all are done
        std::unique_lock<std::mutex> lk(mu_);
        while ((int)workers_.size() < threads - 1) {
            const int id = (int)workers_.size();
            workers_.emplace_back([this, id] { loop(id); });
        }
        job_ = &job; active_ = threads - 1; pending_ = threads - 1; ++gen_;
        lk.unlock();
        cv_.notify_all();
        job(threads - 1);   // the caller takes the last chunk
        lk.lock();
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }

  private:
    void loop(int id) {
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            if (id >= active_) continue;
            const std::function<void(int)>* job = job_;
            lk.unlock();
            (*job)(id);
            lk.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* job_ = nullptr;
    unsigned long long gen_ = 0;
    int active_ = 0, pending_ = 0;
    bool stop_ = false;
};
Pool g_pool;
std::mutex g_pool_user;   // one parallel region at a time

template <class F>
void for_lanes(int n, int threads, F&& body) {
    if (threads > 64) threads = 64;
    if (threads <= 1 || n < 64 * threads) { body(0, n); return; }
    const int per = (n + threads - 1) / threads;
    std::lock_guard<std::mutex> user(g_pool_user);
    g_pool.run(threads, [&](int t) {
        const int lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo < hi) body(lo, hi);
    });
}
}  // namespace

extern "C" {

const char* pime_cpu_last_error(void) { return g_err; }

pime_env_cpu* pime_env_create_cpu(const pime_env_cfg* cfg) {
    if (check_cfg(cfg) != PIME_OK) return nullptr;
    if (cfg->state_mode != PIME_STATE_F64) { set_error("the CPU twin keeps float64 state: state_mode must be PIME_STATE_F64"); return nullptr; }
    if (cfg->kind == PIME_ENV_WT && cfg->num_stack != 0) { set_error("the CPU twin serves the Integrator observation (num_stack 0)"); return nullptr; }
    pime_env_cpu* e = new (std::nothrow) pime_env_cpu();
    if (!e) { set_error("out of host memory"); return nullptr; }
    e->cfg = *cfg;
    e->cfg.ph_table = nullptr;
    const int n = cfg->n_envs;
    e->obs_dim = cfg->kind == PIME_ENV_PH ? 3 : 4;
    double* table = nullptr;
    for (int pass = 0; pass < 2; ++pass) {   // the GPU handle's slab layout, in host memory
        Carver c;
        c.base = pass ? e->slab.data() : nullptr;
        if (cfg->kind == PIME_ENV_PH) carve_ph(c, e->ph64, n, cfg->ph_table_len, &table);
        else carve_wt(c, e->wt64, n, e->obs_dim, 0);
        if (pass == 0) e->slab.assign(((c.off + 255) & ~size_t(255)) + 256, 0);
    }
    int32_t* ep = cfg->kind == PIME_ENV_PH ? e->ph64.episode : e->wt64.episode;
    for (int i = 0; i < n; ++i) ep[i] = -1;   // the first reset is episode 0 (and resamples), as on the GPU
    if (cfg->kind == PIME_ENV_PH) std::memcpy(table, cfg->ph_table, sizeof(double) * cfg->ph_table_len);
    fill_params(e->cfg, e->obs_dim, e->ph, e->wt);
    return e;
}

void pime_env_destroy_cpu(pime_env_cpu* e) { delete e; }

int pime_env_reset_cpu(pime_env_cpu* e, const uint8_t* mask, const double* draws, float* obs, int32_t threads) {
    PIME_REQUIRE(e != nullptr && obs != nullptr, "pime_env_reset_cpu: NULL handle or obs");
    e->was_reset = true;
    const int n = e->cfg.n_envs;
    if (e->cfg.kind == PIME_ENV_PH) {
        for_lanes(n, threads, [&](int lo, int hi) {
            for (int i = lo; i < hi; ++i) {
                if (mask && !mask[i]) continue;
                PhLane<double> L;
                ph_lane_load<double>(e->ph, e->ph64, i, L);
                float o[3];
                ph_lane_reset<double>(e->ph, e->ph64.table, e->ph.env_offset + (uint32_t)i, draws ? draws + 4 * (size_t)i : nullptr, L, o);
                ph_lane_store<double>(e->ph, e->ph64, i, L);
                for (int c = 0; c < 3; ++c) obs[3 * (size_t)i + c] = o[c];
            }
        });
    } else {
        for_lanes(n, threads, [&](int lo, int hi) {
            for (int i = lo; i < hi; ++i) {
                if (mask && !mask[i]) continue;
                WtLane<double> L;
                wt_lane_load<double>(e->wt, e->wt64, i, L);
                wt_lane_reset<double>(e->wt, e->wt.env_offset + (uint32_t)i, draws ? draws + 6 * (size_t)i : nullptr, L);
                wt_lane_store<double>(e->wt, e->wt64, i, L);
                float* o = obs + 4 * (size_t)i;
                o[0] = (float)L.h1; o[1] = (float)L.h2; o[2] = (float)L.r; o[3] = (float)L.I;
            }
        });
    }
    return PIME_OK;
}

// action: float64 env actions [N] (residual == 0), or with obs_in / priorK the pre-tanh residual actions as float32 (the composition
// of agent_residual.py:61 is then applied per lane, as pime_env_step_residual does)
static int step_cpu(pime_env_cpu* e, const double* action, const float* a_pre, const float* obs_in, const double* priorK,
                    const double* noise, int32_t auto_reset, const double* reset_draws, float* obs, float* reward, uint8_t* done,
                    int32_t threads) {
    PIME_REQUIRE(e != nullptr && obs && reward && done && (action || (a_pre && obs_in && priorK)), "pime_env_step_cpu: NULL argument");
    if (!e->was_reset) { set_error("pime_env_step_cpu before pime_env_reset_cpu"); return PIME_ERR_STATE; }
    const int n = e->cfg.n_envs;
    PriorK K{};
    if (priorK)
        for (int j = 0; j < e->obs_dim; ++j) K.k[j] = priorK[j];
    if (e->cfg.kind == PIME_ENV_PH) {
        for_lanes(n, threads, [&](int lo, int hi) {
            for (int i = lo; i < hi; ++i) {
                double a;
                if (action) a = action[i];
                else {
                    const float o_in[3] = {obs_in[3 * (size_t)i], obs_in[3 * (size_t)i + 1], obs_in[3 * (size_t)i + 2]};
                    a = ph_residual_action(a_pre[i], o_in, K);
                }
                PhLane<double> L;
                ph_lane_load<double>(e->ph, e->ph64, i, L);
                float o[3], rew;
                const bool d = ph_lane_step<double>(e->ph, e->ph64.table, a, L, o, rew);
                reward[i] = rew;
                done[i] = (uint8_t)d;
                if (d && auto_reset)
                    ph_lane_reset<double>(e->ph, e->ph64.table, e->ph.env_offset + (uint32_t)i,
                                          reset_draws ? reset_draws + 4 * (size_t)i : nullptr, L, o);
                ph_lane_store<double>(e->ph, e->ph64, i, L);
                for (int c = 0; c < 3; ++c) obs[3 * (size_t)i + c] = o[c];
            }
        });
    } else {
        for_lanes(n, threads, [&](int lo, int hi) {
            for (int i = lo; i < hi; ++i) {
                double a;
                if (action) a = action[i];
                else {
                    a = residual_tanh(a_pre[i]);
                    for (int j = 0; j < 4; ++j) a += (double)obs_in[4 * (size_t)i + j] * K.k[j];
                }
                WtLane<double> L;
                wt_lane_load<double>(e->wt, e->wt64, i, L);
                double z1n, z2n;
                wt_lane_noise<double>(e->wt, e->wt.env_offset + (uint32_t)i, L, noise ? noise + 2 * (size_t)i : nullptr, z1n, z2n);
                float rew;
                const bool d = wt_lane_step<double>(e->wt, a, z1n, z2n, L, rew);
                reward[i] = rew;
                done[i] = (uint8_t)d;
                if (d && auto_reset)
                    wt_lane_reset<double>(e->wt, e->wt.env_offset + (uint32_t)i, reset_draws ? reset_draws + 6 * (size_t)i : nullptr, L);
                wt_lane_store<double>(e->wt, e->wt64, i, L);
                float* o = obs + 4 * (size_t)i;
                o[0] = (float)L.h1; o[1] = (float)L.h2; o[2] = (float)L.r; o[3] = (float)L.I;
            }
        });
    }
    return PIME_OK;
}

int pime_env_step_cpu(pime_env_cpu* e, const double* action, const double* noise, int32_t auto_reset, const double* reset_draws,
                      float* obs, float* reward, uint8_t* done, int32_t threads) {
    return step_cpu(e, action, nullptr, nullptr, nullptr, noise, auto_reset, reset_draws, obs, reward, done, threads);
}

int pime_env_step_residual_cpu(pime_env_cpu* e, const float* a_pre, const float* obs_in, const double* priorK, const double* noise,
                               int32_t auto_reset, const double* reset_draws, float* obs, float* reward, uint8_t* done,
                               int32_t threads) {
    return step_cpu(e, nullptr, a_pre, obs_in, priorK, noise, auto_reset, reset_draws, obs, reward, done, threads);
}

int pime_env_read_field_cpu(pime_env_cpu* e, int32_t field, double* out) {
    PIME_REQUIRE(e != nullptr && out != nullptr, "pime_env_read_field_cpu: NULL argument");
    const int n = e->cfg.n_envs;
    const double* src = nullptr;
    const int32_t* isrc = nullptr;
    if (e->cfg.kind == PIME_ENV_PH) {
        switch (field) {
            case PIME_PH_X: src = e->ph64.x; break;
            case PIME_PH_I: src = e->ph64.I; break;
            case PIME_PH_R: src = e->ph64.r; break;
            case PIME_PH_A: src = e->ph64.A; break;
            case PIME_PH_B: src = e->ph64.B; break;
            case PIME_PH_C: src = e->ph64.C; break;
            case PIME_PH_QWW_V: src = e->ph64.qww; break;
            case PIME_PH_QC_V: src = e->ph64.qc; break;
            case PIME_PH_T: isrc = e->ph64.t; break;
            case PIME_PH_EPISODE: isrc = e->ph64.episode; break;
            default: set_error("field %d is not served by the pH CPU twin", field); return PIME_ERR_ARG;
        }
    } else {
        switch (field) {
            case PIME_WT_H1: src = e->wt64.h1; break;
            case PIME_WT_H2: src = e->wt64.h2; break;
            case PIME_WT_R: src = e->wt64.r; break;
            case PIME_WT_I: src = e->wt64.I; break;
            case PIME_WT_A1: src = e->wt64.a1; break;
            case PIME_WT_A2: src = e->wt64.a2; break;
            case PIME_WT_KP: src = e->wt64.kp; break;
            case PIME_WT_T: isrc = e->wt64.t; break;
            case PIME_WT_EPISODE: isrc = e->wt64.episode; break;
            default: set_error("field %d is not served by the water-tank CPU twin", field); return PIME_ERR_ARG;
        }
    }
    for (int i = 0; i < n; ++i) out[i] = src ? src[i] : (double)isrc[i];
    return PIME_OK;
}

}  // extern "C"
