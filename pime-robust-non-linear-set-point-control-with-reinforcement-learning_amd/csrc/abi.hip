// C ABI of libpime_hip.so (include/pime_hip.h): handle management, argument checks, field I/O, launch dispatch.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "env_handle.hpp"
#include "ppo_train.hpp"
#include "rollout.hpp"
#include "rollout_eval.hpp"
#include "rollout_offpolicy.hpp"
#include "td3.hpp"

namespace pime {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- deferred device frees (pime_common.hpp) ---------------------------------------------------------------------------------
namespace {
struct Release { void* p; int device; ReleaseKind kind; };
std::mutex g_release_mu;
std::vector<Release> g_release_queue;
std::atomic<int> g_capture_depth{0};
void release_now(const Release& r, bool synchronize_first) {
    (void)hipSetDevice(r.device);
    if (synchronize_first) (void)hipDeviceSynchronize();
    if (r.kind == RELEASE_IPC_CLOSE) (void)hipIpcCloseMemHandle(r.p);
    else (void)hipFree(r.p);
}
}  // namespace

bool capture_active() { return g_capture_depth.load(std::memory_order_acquire) > 0; }
void release_device(void* p, int device, ReleaseKind kind, bool synchronize_first) {
    if (!p) return;
    if (capture_active()) {
        std::lock_guard<std::mutex> lk(g_release_mu);
        g_release_queue.push_back(Release{p, device, kind});
        return;
    }
    release_now(Release{p, device, kind}, synchronize_first);
}
int drain_releases() {
    if (capture_active()) return 0;
    std::vector<Release> q;
    {
        std::lock_guard<std::mutex> lk(g_release_mu);
        q.swap(g_release_queue);
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < q.size(); ++i) release_now(q[i], i == 0 || q[i].device != q[i - 1].device);   // one synchronise per device run
    if (!q.empty() && prev >= 0) (void)hipSetDevice(prev);
    return (int)q.size();
}
int queued_releases() {
    std::lock_guard<std::mutex> lk(g_release_mu);
    return (int)g_release_queue.size();
}

// launchers defined in env_kernels.hip / gae_scan.hip / mlp_mfma.hip
// (S, SI, OT): arithmetic / storage type of the slow state words, storage type of the integrated error, output buffer type
template <typename S, typename SI, typename OT>
int launch_ph_reset(const PhParams&, const PhPtrs<S, SI>&, const uint8_t*, const double*, OT*, hipStream_t);
template <typename S, typename SI, typename OT>
int launch_ph_step(const PhParams&, const PhPtrs<S, SI>&, const void*, int, bool, const OT*, const PriorK&, const double*, OT*, OT*,
                   uint8_t*, hipStream_t);
template <typename S, typename SI, typename OT>
int launch_ph_observe(const PhParams&, const PhPtrs<S, SI>&, OT*, hipStream_t);
template <typename S, typename SI, typename OT>
int launch_wt_reset(const WtParams&, const WtPtrs<S, SI>&, const uint8_t*, const double*, OT*, hipStream_t);
template <typename S, typename SI, typename OT>
int launch_wt_step(const WtParams&, const WtPtrs<S, SI>&, const void*, int, bool, const OT*, const PriorK&, const double*,
                   const double*, OT*, OT*, uint8_t*, hipStream_t);
template <typename S, typename SI, typename OT>
int launch_wt_observe(const WtParams&, const WtPtrs<S, SI>&, OT*, hipStream_t);
int launch_gae_scan(const float*, const float*, const float*, int, int, float, int, float*, float*, hipStream_t);
int64_t mlp_packed_floats(int, int, int, int);
int mlp_check(int, int, int, int);
int launch_mlp_pack(int, int, int, int, const float* const*, float*, hipStream_t);
int launch_mlp_forward(int, const float*, int, int, int, int, const float*, float*, hipStream_t);

// ppo_train.hip
int64_t ppo_bwd_image_floats(int, int, int, int);
int64_t ppo_workspace_floats(int, int, int);
inline int stash_tiles_of(int kind, int md) { return (kind == 2 ? 6 : 5) * (md / 32); }
int launch_pack_bwd(int, int, int, int, const float* const*, float*, hipStream_t);
int launch_ppo_net(int, int, const PpoArgs&, hipStream_t);
int launch_ppo_fused(int, int, const PpoArgs&, hipStream_t);
int launch_ppo_fused_dual(int, int, const PpoArgs&, const PpoArgs&, hipStream_t);
int64_t fused_stash_floats(int, int, int);
bool fused_fits(int, int, int, int);
int launch_repack(const PackArgs&, const PackArgs&, float*, float*, float*, float*, hipStream_t);
int launch_grad_reduce(const PpoArgs&, const PpoArgs&, int, int, int, int, bool, bool, bool, bool, int, int, float* const*,
                       float* const*, float*, float*, double*, float*, int, int64_t*, const ReduceAdam*, float*, hipStream_t);
int fused_grid(int);
// mlp16.hip: the streamed 16x16x4 family (width 256; widths 64 / 128 under PIME_MLP16=1)
bool family16(int, int);
bool family16_grad(int, int, int, int);
int64_t ppo_fwd_image_floats(int, int, int, int);
int grid16(int, int, int, int, int);
int launch_pack16(const PackArgs&, float*, float*, hipStream_t);
int launch_pack16_b3(const PackArgs&, float*, hipStream_t);
bool b3_grad(int, int, int, int);
int64_t ppo_bwd_image_f32_floats(int, int, int, int);
int launch_ppo16(int, int, const PpoArgs&, hipStream_t);
int build_dw_jobs(int, int, const PpoArgs&, const float* const*, float* const*, DwJob*);
int launch_dw(const DwArgs&, int, hipStream_t);
int launch_critic_scale(int, int, float* const*, const double*, int, float*, float*, int64_t*, hipStream_t);
int launch_adam(float*, float*, float*, float*, long long, float, float, float, float, float*, const int32_t*, float* const (*)[2],
                const float*, long long, int, hipStream_t);
int launch_rollout(int, int, const RolloutArgs&, hipStream_t);
// td3_fused.hip
int td3_grid(int);
int64_t td3_workspace_floats(int, int, int);
bool td3_supported(int, int, int);
int launch_td3_grad(bool, int, const Td3GradArgs&, int, hipStream_t);
int launch_td3_apply(const Td3ApplyArgs&, hipStream_t);

}  // namespace pime

using namespace pime;

// One slab of HBM per handle; SoA arrays are carved from it at 256-B boundaries.
struct pime_env {
    pime_env_cfg cfg;
    int obs_dim = 0;
    bool was_reset = false;
    void* slab = nullptr;
    size_t slab_bytes = 0;
    std::vector<double> table_host;  // pH: for PIME_PH_Y reads
    PhParams ph{};
    WtParams wt{};
    PhPtrs<double> ph64{};
    PhPtrs<float> ph32{};
    WtPtrs<double> wt64{};
    WtPtrs<float> wt32{};
    PhPtrs<float, half_t> ph16{};   // PIME_STATE_MIXED16: as ph32 / wt32 with the integrated error stored as binary16
    WtPtrs<float, half_t> wt16{};
};

namespace {

int use_device(const pime_env* e) {
    PIME_HIP_TRY(hipSetDevice(e->cfg.device_id));
    return PIME_OK;
}

// device array <-> host double conversions for field I/O
struct FieldRef {
    void* ptr = nullptr;
    int type = 0;  // 0 double, 1 float, 2 int32, 3 binary16
    bool ph_params = false, read_only = false, is_y = false;
};

int resolve_field(pime_env* e, int field, FieldRef* out) {
    const bool f64 = e->cfg.state_mode == PIME_STATE_F64;
    FieldRef r;
#define SPTR(member) (f64 ? (void*)e->ph64.member : (void*)e->ph32.member)
#define WPTR(member) (f64 ? (void*)e->wt64.member : (void*)e->wt32.member)
    const int st = f64 ? 0 : 1;
    const int sti = e->cfg.state_mode == PIME_STATE_MIXED16 ? 3 : st;   // the integrated error: binary16 in MIXED16 (ph32 / wt32
                                                                        // alias every other array of ph16 / wt16)
    if (e->cfg.kind == PIME_ENV_PH) {
        switch (field) {
            case PIME_PH_X: r = {e->ph64.x, 0}; break;
            case PIME_PH_I: r = {sti == 3 ? (void*)e->ph16.I : SPTR(I), sti}; break;
            case PIME_PH_R: r = {SPTR(r), st}; break;
            case PIME_PH_Y: r = {nullptr, 0, false, true, true}; break;
            case PIME_PH_A: r = {e->ph64.A, 0}; break;
            case PIME_PH_B: r = {e->ph64.B, 0}; break;
            case PIME_PH_C: r = {e->ph64.C, 0}; break;
            case PIME_PH_QWW_V: r = {e->ph64.qww, 0, true}; break;
            case PIME_PH_QC_V: r = {e->ph64.qc, 0, true}; break;
            case PIME_PH_T: r = {e->ph64.t, 2}; break;
            case PIME_PH_EPISODE: r = {e->ph64.episode, 2}; break;
            default: set_error("field %d is not a pH field", field); return PIME_ERR_ARG;
        }
    } else {
        switch (field) {
            case PIME_WT_H1: r = {WPTR(h1), st}; break;
            case PIME_WT_H2: r = {WPTR(h2), st}; break;
            case PIME_WT_R: r = {WPTR(r), st}; break;
            case PIME_WT_I:
                if (e->cfg.num_stack > 0) { set_error("the Stacking variant has no integrator"); return PIME_ERR_ARG; }
                r = {sti == 3 ? (void*)e->wt16.I : WPTR(I), sti}; break;
            case PIME_WT_A1: r = {WPTR(a1), st}; break;
            case PIME_WT_A2: r = {WPTR(a2), st}; break;
            case PIME_WT_KP: r = {WPTR(kp), st}; break;
            case PIME_WT_T: r = {e->wt64.t, 2}; break;
            case PIME_WT_EPISODE: r = {e->wt64.episode, 2}; break;
            default: set_error("field %d is not a water-tank field", field); return PIME_ERR_ARG;
        }
    }
#undef SPTR
#undef WPTR
    *out = r;
    return PIME_OK;
}

int d2h_as_double(const FieldRef& f, int n, double* out, hipStream_t s) {
    if (f.type == 0) {
        PIME_HIP_TRY(hipMemcpyAsync(out, f.ptr, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        PIME_HIP_TRY(hipStreamSynchronize(s));
    } else if (f.type == 1) {
        std::vector<float> tmp(n);
        PIME_HIP_TRY(hipMemcpyAsync(tmp.data(), f.ptr, sizeof(float) * n, hipMemcpyDeviceToHost, s));
        PIME_HIP_TRY(hipStreamSynchronize(s));
        for (int i = 0; i < n; ++i) out[i] = tmp[i];
    } else if (f.type == 3) {
        std::vector<half_t> tmp(n);
        PIME_HIP_TRY(hipMemcpyAsync(tmp.data(), f.ptr, sizeof(half_t) * n, hipMemcpyDeviceToHost, s));
        PIME_HIP_TRY(hipStreamSynchronize(s));
        for (int i = 0; i < n; ++i) out[i] = (double)(float)tmp[i];
    } else {
        std::vector<int32_t> tmp(n);
        PIME_HIP_TRY(hipMemcpyAsync(tmp.data(), f.ptr, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
        PIME_HIP_TRY(hipStreamSynchronize(s));
        for (int i = 0; i < n; ++i) out[i] = tmp[i];
    }
    return PIME_OK;
}

int h2d_from_double(const FieldRef& f, int n, const double* in, hipStream_t s) {
    if (f.type == 0) {
        PIME_HIP_TRY(hipMemcpyAsync(f.ptr, in, sizeof(double) * n, hipMemcpyHostToDevice, s));
        PIME_HIP_TRY(hipStreamSynchronize(s));
    } else if (f.type == 1) {
        std::vector<float> tmp(n);
        for (int i = 0; i < n; ++i) tmp[i] = (float)in[i];
        PIME_HIP_TRY(hipMemcpyAsync(f.ptr, tmp.data(), sizeof(float) * n, hipMemcpyHostToDevice, s));
        PIME_HIP_TRY(hipStreamSynchronize(s));
    } else if (f.type == 3) {
        std::vector<half_t> tmp(n);
        for (int i = 0; i < n; ++i) tmp[i] = (half_t)(float)in[i];
        PIME_HIP_TRY(hipMemcpyAsync(f.ptr, tmp.data(), sizeof(half_t) * n, hipMemcpyHostToDevice, s));
        PIME_HIP_TRY(hipStreamSynchronize(s));
    } else {
        std::vector<int32_t> tmp(n);
        for (int i = 0; i < n; ++i) tmp[i] = (int32_t)in[i];
        PIME_HIP_TRY(hipMemcpyAsync(f.ptr, tmp.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
        PIME_HIP_TRY(hipStreamSynchronize(s));
    }
    return PIME_OK;
}

}  // namespace

template <typename S>
static int rollout_eval_t(pime_env* e, const PhPtrs<S>& ph, const WtPtrs<S>& wt, int32_t kind, int32_t md, const float* packed_actor,
                          const double* priorK, int32_t n_steps, int32_t seg_len, const double* setpoints, int32_t n_setpoints,
                          double* ret, double* trace, pime_stream stream) {
    EvalArgs<S> a{};
    a.env = e->cfg.kind == PIME_ENV_PH ? 0 : 1;
    a.n = e->cfg.n_envs;
    a.env_offset = e->cfg.env_offset;
    if (a.env == 0) { a.p = e->ph; a.p.auto_reset = 0; a.st = ph; }
    else { a.wp = e->wt; a.wp.auto_reset = 0; a.wst = wt; }
    a.img = packed_actor;
    for (int j = 0; j < e->obs_dim; ++j) a.K.k[j] = priorK[j];
    a.n_steps = n_steps; a.seg_len = seg_len;
    if (seg_len > 0)   // (validated by the caller: setpoints != NULL, 1 <= n_setpoints <= kMaxSetpoints)
        for (int j = 0; j < n_setpoints && j < kMaxSetpoints; ++j) a.setpoint[j] = setpoints[j];
    a.ret = ret; a.trace = trace;
    return launch_rollout_eval<S>(kind, md, a, static_cast<hipStream_t>(stream));
}


extern "C" {

int pime_abi_version(void) { return PIME_ABI_VERSION; }
const char* pime_last_error(void) { return g_err; }

int pime_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int pime_env_cfg_default(int32_t kind, pime_env_cfg* c) {
    PIME_REQUIRE(c != nullptr, "cfg is NULL");
    std::memset(c, 0, sizeof(*c));
    c->kind = kind;
    c->n_envs = 1;
    c->state_mode = PIME_STATE_F64;
    c->integral_bound = 1;
    c->resample_every = 1;       // the reference resamples on every reset (ph.py:413, nonlinear_watertank.py:903)
    c->integral_max = 25.0;      // ph.py:299, nonlinear_watertank.py:732
    c->distance_threshold = 0.05;
    if (kind == PIME_ENV_PH) {   // gym_control/__init__.py:3-14
        c->max_steps = 50;
        c->reward_type = PIME_REWARD_SQUARE;
        c->range_lo[0] = 0.005; c->range_hi[0] = 0.015;    // qww_V, ph.py:357
        c->range_lo[1] = 0.0015; c->range_hi[1] = 0.0025;  // qc_V, ph.py:358
        c->init_lo[0] = 0.0; c->init_hi[0] = 50.0;         // x0, ph.py:420
        c->init_lo[1] = 3.0; c->init_hi[1] = 11.0;         // r, ph.py:424
        c->ph_sample_t = 20.0; c->ph_u_low = 0.0; c->ph_u_high = 1.5; c->ph_table_scale = 1e5;
    } else if (kind == PIME_ENV_WT) {  // gym_control/__init__.py:50-69
        c->max_steps = 200;
        c->reward_type = PIME_REWARD_SQUARE;
        c->range_lo[0] = 0.0015; c->range_hi[0] = 0.0024;
        c->range_lo[1] = 0.0015; c->range_hi[1] = 0.0024;
        c->range_lo[2] = 0.07; c->range_hi[2] = 0.17;
        c->init_lo[0] = 0.0; c->init_hi[0] = 10.0;         // h1,h2: nonlinear_watertank.py:912
        c->init_lo[1] = 0.0; c->init_hi[1] = 10.0;         // r: :913
        c->wt_A1 = 1; c->wt_A2 = 1; c->wt_G = 980; c->wt_n_discrete = 20; c->wt_dt = 2.0 / 20;
        c->wt_noise_scale = 0.01; c->wt_z1 = 1; c->wt_pmax = 10.0;
    } else {
        set_error("unknown env kind %d", kind);
        return PIME_ERR_ARG;
    }
    return PIME_OK;
}

pime_env* pime_env_create(const pime_env_cfg* cfg) {
    (void)drain_releases();   // slabs parked by handles destroyed under a stream capture
    if (check_cfg(cfg) != PIME_OK) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible: libpime_hip has no CPU fallback");
        return nullptr;
    }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) {
        set_error("device_id %d out of range [0,%d)", cfg->device_id, ndev);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is '%s'; libpime_hip is built for gfx950 only", cfg->device_id, prop.gcnArchName);
        return nullptr;
    }
    pime_env* e = new (std::nothrow) pime_env();
    if (!e) { set_error("out of host memory"); return nullptr; }
    e->cfg = *cfg;
    e->cfg.ph_table = nullptr;  // never keep the caller's pointer
    const int n = cfg->n_envs;
    const bool f64 = cfg->state_mode == PIME_STATE_F64;
    e->obs_dim = cfg->kind == PIME_ENV_PH ? 3 : (cfg->num_stack > 0 ? 3 * cfg->num_stack : 4);
    void* table_dev = nullptr;
    for (int pass = 0; pass < 2; ++pass) {  // pass 0 sizes the slab, pass 1 carves it
        Carver c;
        c.base = pass ? static_cast<char*>(e->slab) : nullptr;
        if (cfg->kind == PIME_ENV_PH) {
            if (f64) { double* t; carve_ph(c, e->ph64, n, cfg->ph_table_len, &t); table_dev = t; }
            else {
                float* t;
                if (cfg->state_mode == PIME_STATE_MIXED16) {   // ph32 aliases every array but I (which it must not touch)
                    carve_ph(c, e->ph16, n, cfg->ph_table_len, &t);
                    const PhPtrs<float, half_t>& h = e->ph16;
                    e->ph32.x = h.x; e->ph32.A = h.A; e->ph32.B = h.B; e->ph32.C = h.C; e->ph32.qww = h.qww; e->ph32.qc = h.qc;
                    e->ph32.r = h.r; e->ph32.last_a = h.last_a; e->ph32.I = nullptr; e->ph32.t = h.t; e->ph32.episode = h.episode;
                    e->ph32.table = h.table;
                } else {
                    carve_ph(c, e->ph32, n, cfg->ph_table_len, &t);
                }
                table_dev = t;
                // shared float64 arrays are addressed through ph64 by the field accessors
                e->ph64.x = e->ph32.x; e->ph64.A = e->ph32.A; e->ph64.B = e->ph32.B; e->ph64.C = e->ph32.C;
                e->ph64.qww = e->ph32.qww; e->ph64.qc = e->ph32.qc; e->ph64.t = e->ph32.t; e->ph64.episode = e->ph32.episode;
            }
        } else {
            if (f64) carve_wt(c, e->wt64, n, e->obs_dim, cfg->num_stack);
            else if (cfg->state_mode == PIME_STATE_MIXED16) {
                carve_wt(c, e->wt16, n, e->obs_dim, cfg->num_stack);
                const WtPtrs<float, half_t>& h = e->wt16;
                e->wt32.h1 = h.h1; e->wt32.h2 = h.h2; e->wt32.r = h.r; e->wt32.a1 = h.a1; e->wt32.a2 = h.a2; e->wt32.kp = h.kp;
                e->wt32.I = nullptr; e->wt32.frames = h.frames; e->wt32.head = h.head; e->wt32.t = h.t; e->wt32.episode = h.episode;
                e->wt64.t = h.t; e->wt64.episode = h.episode;
            }
            else { carve_wt(c, e->wt32, n, e->obs_dim, cfg->num_stack); e->wt64.t = e->wt32.t; e->wt64.episode = e->wt32.episode; }
        }
        if (pass == 0) {
            e->slab_bytes = (c.off + 255) & ~size_t(255);
            if (hipSetDevice(cfg->device_id) != hipSuccess || hipMalloc(&e->slab, e->slab_bytes) != hipSuccess) {
                set_error("hipMalloc of %zu B env state failed: %s", e->slab_bytes, hipGetErrorString(hipGetLastError()));
                delete e;
                return nullptr;
            }
        }
    }
    bool ok = hipMemset(e->slab, 0, e->slab_bytes) == hipSuccess;
    // episode counters start at -1 so that the first reset is episode 0 (and resamples)
    std::vector<int32_t> minus1(n, -1);
    int32_t* ep = cfg->kind == PIME_ENV_PH ? e->ph64.episode : e->wt64.episode;
    ok = ok && hipMemcpy(ep, minus1.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice) == hipSuccess;
    if (cfg->kind == PIME_ENV_PH) {
        e->table_host.assign(cfg->ph_table, cfg->ph_table + cfg->ph_table_len);
        if (f64) {
            ok = ok && hipMemcpy(table_dev, e->table_host.data(), sizeof(double) * cfg->ph_table_len, hipMemcpyHostToDevice) == hipSuccess;
        } else {
            std::vector<float> t32(e->table_host.begin(), e->table_host.end());
            ok = ok && hipMemcpy(table_dev, t32.data(), sizeof(float) * cfg->ph_table_len, hipMemcpyHostToDevice) == hipSuccess;
        }
    }
    if (!ok) {
        set_error("initialising env state failed: %s", hipGetErrorString(hipGetLastError()));
        (void)hipFree(e->slab);
        delete e;
        return nullptr;
    }
    fill_params(e->cfg, e->obs_dim, e->ph, e->wt);
    return e;
}

void pime_env_destroy(pime_env* e) {
    if (!e) return;
    // under a stream capture (pime_capture_begin, or the runtime says so) the slab is parked and freed by the next entry point
    // outside the capture; otherwise: wait for the launches that use it, free it, and whatever an earlier capture left queued
    release_device(e->slab, e->cfg.device_id, RELEASE_FREE, /*synchronize_first=*/true);
    delete e;
    (void)drain_releases();
}

void pime_capture_begin(void) { g_capture_depth.fetch_add(1, std::memory_order_acq_rel); }
void pime_capture_end(void) {
    if (g_capture_depth.fetch_sub(1, std::memory_order_acq_rel) <= 1) {
        g_capture_depth.store(0, std::memory_order_release);
        (void)drain_releases();
    }
}
void pime_capture_leave(void) {   // as pime_capture_end without the drain: for finalisers that run inside a capture nobody announced
    if (g_capture_depth.fetch_sub(1, std::memory_order_acq_rel) <= 1) g_capture_depth.store(0, std::memory_order_release);
}
int pime_deferred_releases(void) { return queued_releases(); }

int32_t pime_env_obs_dim(const pime_env* e) { return e ? e->obs_dim : 0; }
int32_t pime_env_num_envs(const pime_env* e) { return e ? e->cfg.n_envs : 0; }
int32_t pime_env_reset_draw_width(const pime_env* e) { return e ? (e->cfg.kind == PIME_ENV_PH ? 4 : 6) : 0; }

// OT = float: the regular entry points (every state mode); OT = binary16: the *_h entry points (PIME_STATE_MIXED16 only)
extern "C++" {
template <typename OT>
static int reset_common(pime_env* e, const uint8_t* mask, const double* draws, OT* obs, pime_stream stream) {
    PIME_REQUIRE(e != nullptr && obs != nullptr, "pime_env_reset: NULL handle or obs");
    if (int rc = use_device(e)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    e->was_reset = true;
    const int mode = e->cfg.state_mode;
    if constexpr (std::is_same<OT, half_t>::value) {
        PIME_REQUIRE(mode == PIME_STATE_MIXED16, "binary16 observations need an env handle in PIME_STATE_MIXED16 mode");
        return e->cfg.kind == PIME_ENV_PH ? launch_ph_reset(e->ph, e->ph16, mask, draws, obs, s)
                                          : launch_wt_reset(e->wt, e->wt16, mask, draws, obs, s);
    } else {
        if (e->cfg.kind == PIME_ENV_PH)
            return mode == PIME_STATE_F64 ? launch_ph_reset(e->ph, e->ph64, mask, draws, obs, s)
                 : mode == PIME_STATE_MIXED ? launch_ph_reset(e->ph, e->ph32, mask, draws, obs, s)
                                            : launch_ph_reset(e->ph, e->ph16, mask, draws, obs, s);
        return mode == PIME_STATE_F64 ? launch_wt_reset(e->wt, e->wt64, mask, draws, obs, s)
             : mode == PIME_STATE_MIXED ? launch_wt_reset(e->wt, e->wt32, mask, draws, obs, s)
                                        : launch_wt_reset(e->wt, e->wt16, mask, draws, obs, s);
    }
}

}  // extern "C++"

int pime_env_reset(pime_env* e, const uint8_t* mask, const double* draws, float* obs, pime_stream stream) {
    return reset_common<float>(e, mask, draws, obs, stream);
}
int pime_env_reset_h(pime_env* e, const uint8_t* mask, const double* draws, uint16_t* obs, pime_stream stream) {
    return reset_common<half_t>(e, mask, draws, reinterpret_cast<half_t*>(obs), stream);
}

extern "C++" {
template <typename OT>
static int step_common(pime_env* e, const void* act, int act_dtype, bool residual, const OT* obs_in, const double* priorK,
                       const double* noise, int auto_reset, const double* reset_draws, OT* obs, OT* reward, uint8_t* done,
                       pime_stream stream) {
    PIME_REQUIRE(e != nullptr, "NULL env handle");
    PIME_REQUIRE(act && obs && reward && done, "pime_env_step: NULL action/obs/reward/done");
    PIME_REQUIRE(act_dtype == PIME_F32 || act_dtype == PIME_F64, "action_dtype %d", act_dtype);
    if (!e->was_reset) { set_error("pime_env_step before pime_env_reset"); return PIME_ERR_STATE; }
    PriorK K{};
    if (residual) {
        PIME_REQUIRE(obs_in != nullptr && priorK != nullptr, "pime_env_step_residual: NULL obs_in/priorK");
        for (int j = 0; j < e->obs_dim; ++j) K.k[j] = priorK[j];
    }
    if (int rc = use_device(e)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int mode = e->cfg.state_mode;
    constexpr bool kHalf = std::is_same<OT, half_t>::value;
    if (kHalf) PIME_REQUIRE(mode == PIME_STATE_MIXED16, "binary16 observations need an env handle in PIME_STATE_MIXED16 mode");
    if (e->cfg.kind == PIME_ENV_PH) {
        PIME_REQUIRE(noise == nullptr, "the pH env has no process noise input");
        PhParams p = e->ph;
        p.auto_reset = auto_reset ? 1 : 0;
        if constexpr (kHalf) return launch_ph_step(p, e->ph16, act, act_dtype, residual, obs_in, K, reset_draws, obs, reward, done, s);
        else
            return mode == PIME_STATE_F64 ? launch_ph_step(p, e->ph64, act, act_dtype, residual, obs_in, K, reset_draws, obs, reward, done, s)
                 : mode == PIME_STATE_MIXED ? launch_ph_step(p, e->ph32, act, act_dtype, residual, obs_in, K, reset_draws, obs, reward, done, s)
                                            : launch_ph_step(p, e->ph16, act, act_dtype, residual, obs_in, K, reset_draws, obs, reward, done, s);
    }
    WtParams p = e->wt;
    p.auto_reset = auto_reset ? 1 : 0;
    if constexpr (kHalf) return launch_wt_step(p, e->wt16, act, act_dtype, residual, obs_in, K, noise, reset_draws, obs, reward, done, s);
    else
        return mode == PIME_STATE_F64 ? launch_wt_step(p, e->wt64, act, act_dtype, residual, obs_in, K, noise, reset_draws, obs, reward, done, s)
             : mode == PIME_STATE_MIXED ? launch_wt_step(p, e->wt32, act, act_dtype, residual, obs_in, K, noise, reset_draws, obs, reward, done, s)
                                        : launch_wt_step(p, e->wt16, act, act_dtype, residual, obs_in, K, noise, reset_draws, obs, reward, done, s);
}

}  // extern "C++"

int pime_env_step(pime_env* e, const void* action, int32_t action_dtype, const double* noise, int32_t auto_reset,
                  const double* reset_draws, float* obs, float* reward, uint8_t* done, pime_stream stream) {
    return step_common<float>(e, action, action_dtype, false, nullptr, nullptr, noise, auto_reset, reset_draws, obs, reward, done, stream);
}

int pime_env_step_residual(pime_env* e, const float* a_pre, const float* obs_in, const double* priorK, const double* noise,
                           int32_t auto_reset, const double* reset_draws, float* obs, float* reward, uint8_t* done,
                           pime_stream stream) {
    return step_common<float>(e, a_pre, PIME_F32, true, obs_in, priorK, noise, auto_reset, reset_draws, obs, reward, done, stream);
}

int pime_env_step_h(pime_env* e, const float* action, const double* noise, int32_t auto_reset, const double* reset_draws,
                    uint16_t* obs, uint16_t* reward, uint8_t* done, pime_stream stream) {
    return step_common<half_t>(e, action, PIME_F32, false, nullptr, nullptr, noise, auto_reset, reset_draws,
                               reinterpret_cast<half_t*>(obs), reinterpret_cast<half_t*>(reward), done, stream);
}

int pime_env_step_residual_h(pime_env* e, const float* a_pre, const uint16_t* obs_in, const double* priorK, const double* noise,
                             int32_t auto_reset, const double* reset_draws, uint16_t* obs, uint16_t* reward, uint8_t* done,
                             pime_stream stream) {
    return step_common<half_t>(e, a_pre, PIME_F32, true, reinterpret_cast<const half_t*>(obs_in), priorK, noise, auto_reset,
                               reset_draws, reinterpret_cast<half_t*>(obs), reinterpret_cast<half_t*>(reward), done, stream);
}

int pime_env_observe(pime_env* e, float* obs, pime_stream stream) {
    PIME_REQUIRE(e != nullptr && obs != nullptr, "pime_env_observe: NULL handle or obs");
    if (int rc = use_device(e)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int mode = e->cfg.state_mode;
    if (e->cfg.kind == PIME_ENV_PH)
        return mode == PIME_STATE_F64 ? launch_ph_observe(e->ph, e->ph64, obs, s)
             : mode == PIME_STATE_MIXED ? launch_ph_observe(e->ph, e->ph32, obs, s) : launch_ph_observe(e->ph, e->ph16, obs, s);
    return mode == PIME_STATE_F64 ? launch_wt_observe(e->wt, e->wt64, obs, s)
         : mode == PIME_STATE_MIXED ? launch_wt_observe(e->wt, e->wt32, obs, s) : launch_wt_observe(e->wt, e->wt16, obs, s);
}

int pime_env_read_field(pime_env* e, int32_t field, double* out, pime_stream stream) {
    PIME_REQUIRE(e != nullptr && out != nullptr, "pime_env_read_field: NULL handle or out");
    if (int rc = use_device(e)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    FieldRef f;
    if (int rc = resolve_field(e, field, &f)) return rc;
    const int n = e->cfg.n_envs;
    if (f.is_y) {  // y = pH_table[rint(C*x*1e5)], ph.py:187-189, from the float64 host copy of the table
        std::vector<double> x(n), C(n);
        FieldRef fx{e->ph64.x, 0}, fc{e->ph64.C, 0};
        if (int rc = d2h_as_double(fx, n, x.data(), s)) return rc;
        if (int rc = d2h_as_double(fc, n, C.data(), s)) return rc;
        for (int i = 0; i < n; ++i) {
            long long k = std::llrint(C[i] * x[i] * e->cfg.ph_table_scale);
            k = k < 0 ? 0 : (k >= e->cfg.ph_table_len ? e->cfg.ph_table_len - 1 : k);
            out[i] = e->table_host[(size_t)k];
        }
        return PIME_OK;
    }
    return d2h_as_double(f, n, out, s);
}

int pime_env_write_field(pime_env* e, int32_t field, const double* in, const uint8_t* mask, pime_stream stream) {
    PIME_REQUIRE(e != nullptr && in != nullptr, "pime_env_write_field: NULL handle or in");
    if (int rc = use_device(e)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    FieldRef f;
    if (int rc = resolve_field(e, field, &f)) return rc;
    PIME_REQUIRE(!f.read_only, "field %d is read-only", field);
    const int n = e->cfg.n_envs;
    std::vector<double> v(n);
    if (mask) {
        if (int rc = d2h_as_double(f, n, v.data(), s)) return rc;
        for (int i = 0; i < n; ++i) if (mask[i]) v[i] = in[i];
    } else {
        std::memcpy(v.data(), in, sizeof(double) * n);
    }
    if (int rc = h2d_from_double(f, n, v.data(), s)) return rc;
    if (f.ph_params) {  // update_system (ph.py:114-121): rebuild the ZOH plant of every lane from (qww_V, qc_V)
        std::vector<double> qww(n), qc(n), A(n), B(n);
        FieldRef fq{e->ph64.qww, 0}, fc{e->ph64.qc, 0}, fA{e->ph64.A, 0}, fB{e->ph64.B, 0}, fC{e->ph64.C, 0};
        if (int rc = d2h_as_double(fq, n, qww.data(), s)) return rc;
        if (int rc = d2h_as_double(fc, n, qc.data(), s)) return rc;
        for (int i = 0; i < n; ++i) {
            const double ex = -qww[i] * e->cfg.ph_sample_t;
            A[i] = std::exp(ex);
            B[i] = qww[i] != 0.0 ? -std::expm1(ex) / qww[i] : e->cfg.ph_sample_t;
        }
        if (int rc = h2d_from_double(fA, n, A.data(), s)) return rc;
        if (int rc = h2d_from_double(fB, n, B.data(), s)) return rc;
        if (int rc = h2d_from_double(fC, n, qc.data(), s)) return rc;
    }
    return PIME_OK;
}

int pime_env_set_punish(pime_env* e, double integral_punish, double action_punish, double action_change_punish) {
    PIME_REQUIRE(e != nullptr, "NULL env handle");
    e->cfg.integral_punish = integral_punish;
    e->cfg.action_punish = action_punish;
    e->cfg.action_change_punish = action_change_punish;
    fill_params(e->cfg, e->obs_dim, e->ph, e->wt);
    return PIME_OK;
}

int pime_env_set_max_steps(pime_env* e, int32_t max_steps) {
    PIME_REQUIRE(e != nullptr && max_steps >= 1, "pime_env_set_max_steps: bad arguments");
    e->cfg.max_steps = max_steps;
    fill_params(e->cfg, e->obs_dim, e->ph, e->wt);
    return PIME_OK;
}

int pime_env_set_resample_every(pime_env* e, int32_t n) {
    PIME_REQUIRE(e != nullptr && n >= 0, "pime_env_set_resample_every: bad arguments");
    e->cfg.resample_every = n;
    fill_params(e->cfg, e->obs_dim, e->ph, e->wt);
    return PIME_OK;
}

int pime_gae_scan(const float* reward, const float* mask, const float* value, int32_t T, int32_t N, float lambda,
                  int32_t use_gae, float* r_sum, float* adv, pime_stream stream) {
    PIME_REQUIRE(reward && mask && value && r_sum && adv, "pime_gae_scan: NULL buffer");
    PIME_REQUIRE(T >= 1 && N >= 1, "pime_gae_scan: T=%d N=%d", T, N);
    return launch_gae_scan(reward, mask, value, T, N, lambda, use_gae, r_sum, adv, static_cast<hipStream_t>(stream));
}

int64_t pime_mlp_packed_floats(int32_t kind, int32_t D, int32_t Di, int32_t md) {
    if (mlp_check(kind, D, Di, md) != PIME_OK) return 0;
    return mlp_packed_floats(kind, D, Di, md);
}

int pime_mlp_pack(int32_t kind, int32_t D, int32_t Di, int32_t md, const float* const* params, float* packed,
                  pime_stream stream) {
    PIME_REQUIRE(params != nullptr && packed != nullptr, "pime_mlp_pack: NULL params/packed");
    return launch_mlp_pack(kind, D, Di, md, params, packed, static_cast<hipStream_t>(stream));
}

int pime_mlp_forward(int32_t kind, const float* x, int32_t M, int32_t D, int32_t Di, int32_t md, const float* packed,
                     float* out, pime_stream stream) {
    PIME_REQUIRE(x != nullptr && packed != nullptr && out != nullptr, "pime_mlp_forward: NULL x/packed/out");
    return launch_mlp_forward(kind, x, M, D, Di, md, packed, out, static_cast<hipStream_t>(stream));
}

int64_t pime_ppo_bwd_image_floats(int32_t kind, int32_t D, int32_t Di, int32_t md) {
    if (mlp_check(kind, D, Di, md) != PIME_OK) return 0;
    return ppo_bwd_image_floats(kind, D, Di, md);
}
int64_t pime_ppo_bwd_image_f32_floats(int32_t kind, int32_t D, int32_t Di, int32_t md) {
    if (mlp_check(kind, D, Di, md) != PIME_OK) return 0;
    return ppo_bwd_image_f32_floats(kind, D, Di, md);
}

int64_t pime_ppo_fwd_image_floats(int32_t kind, int32_t D, int32_t Di, int32_t md) {
    if (mlp_check(kind, D, Di, md) != PIME_OK) return 0;
    return ppo_fwd_image_floats(kind, D, Di, md);
}

int64_t pime_ppo_workspace_floats(int32_t kind, int32_t B, int32_t md) {
    if (B < 1 || (md != 64 && md != 128 && md != 256)) {
        set_error("pime_ppo_workspace_floats: B=%d md=%d kind=%d", B, md, kind);
        return 0;
    }
    return ppo_workspace_floats(kind, B, md);
}

int pime_ppo_pack_bwd(int32_t kind, int32_t D, int32_t Di, int32_t md, const float* const* params, float* image,
                      pime_stream stream) {
    PIME_REQUIRE(params != nullptr && image != nullptr, "pime_ppo_pack_bwd: NULL params/image");
    return launch_pack_bwd(kind, D, Di, md, params, image, static_cast<hipStream_t>(stream));
}

int pime_rollout_supported(const pime_env* e, int32_t kind, int32_t md) {
    if (e == nullptr) return 0;
    if (e->cfg.state_mode != PIME_STATE_MIXED && e->cfg.state_mode != PIME_STATE_MIXED16) return 0;
    if (e->cfg.kind != PIME_ENV_PH && e->cfg.num_stack != 0) {   // Stacking1/4/10 under a plain actor (no integrator column)
        const int S = e->cfg.num_stack;
        if (kind != PIME_MLP_PLAIN_ACTOR || !(S == 1 || S == 4 || S == 10)) return 0;
        if (e->cfg.state_mode == PIME_STATE_MIXED16) return 0;   // binary16 rows: pH and the Integrator tank (config 5)
    }
    if (kind != PIME_MLP_PLAIN_ACTOR && kind != PIME_MLP_MODULAR_ACTOR) return 0;
    if (md == 256) return e->cfg.state_mode == PIME_STATE_MIXED ? 1 : 0;   // the streamed 16-tile rollout (mlp16.hip); float32 rows only
    return (md == 64 || md == 128) && !family16(kind, md) ? 1 : 0;
}

static int rollout_common(pime_env* e, int32_t kind, int32_t md, const float* packed_actor, const float* a_std_log,
                          const double* priorK, int32_t n_steps, uint64_t noise_seed, uint32_t noise_epoch, void* state,
                          float* action, float* noise, void* reward, uint8_t* done, bool half, pime_stream stream) {
    PIME_REQUIRE(e != nullptr, "NULL env handle");
    PIME_REQUIRE(e->cfg.state_mode == (half ? PIME_STATE_MIXED16 : PIME_STATE_MIXED),
                 "pime_rollout needs an env handle in PIME_STATE_MIXED mode, pime_rollout_h one in PIME_STATE_MIXED16 mode");
    PIME_REQUIRE(packed_actor && a_std_log && priorK && state && action && noise && reward && done && n_steps >= 1,
                 "pime_rollout: bad arguments");
    PIME_REQUIRE(pime_rollout_supported(e, kind, md), "pime_rollout: no fused rollout for actor kind %d width %d "
                 "(pime_rollout_supported)", kind, md);
    if (!e->was_reset) { set_error("pime_rollout before pime_env_reset"); return PIME_ERR_STATE; }
    if (int rc = use_device(e)) return rc;
    RolloutArgs a{};
    a.env = e->cfg.kind == PIME_ENV_PH ? 0 : (e->cfg.num_stack == 0 ? 1 : 2);
    a.n = e->cfg.n_envs;
    a.env_offset = e->cfg.env_offset;
    if (a.env == 0) { a.p = e->ph; a.p.auto_reset = 1; a.st = e->ph32; }
    else { a.wp = e->wt; a.wp.auto_reset = 1; a.wst = e->wt32; }
    a.img = packed_actor; a.a_std_log = a_std_log;
    for (int j = 0; j < e->obs_dim; ++j) a.K.k[j] = priorK[j];
    a.n_steps = n_steps; a.noise_seed = noise_seed; a.noise_epoch = noise_epoch;
    a.action = action; a.noise = noise; a.done = done;
    if (half) {
        a.I16 = a.env == 0 ? e->ph16.I : e->wt16.I;
        a.state_h = static_cast<half_t*>(state); a.reward_h = static_cast<half_t*>(reward);
    } else {
        a.state = static_cast<float*>(state); a.reward = static_cast<float*>(reward);
    }
    return launch_rollout(kind, md, a, static_cast<hipStream_t>(stream));
}

int pime_rollout(pime_env* e, int32_t kind, int32_t md, const float* packed_actor, const float* a_std_log,
                 const double* priorK, int32_t n_steps, uint64_t noise_seed, uint32_t noise_epoch, float* state,
                 float* action, float* noise, float* reward, uint8_t* done, pime_stream stream) {
    return rollout_common(e, kind, md, packed_actor, a_std_log, priorK, n_steps, noise_seed, noise_epoch, state, action, noise,
                          reward, done, false, stream);
}

int pime_rollout_h(pime_env* e, int32_t kind, int32_t md, const float* packed_actor, const float* a_std_log,
                   const double* priorK, int32_t n_steps, uint64_t noise_seed, uint32_t noise_epoch, uint16_t* state,
                   float* action, float* noise, uint16_t* reward, uint8_t* done, pime_stream stream) {
    return rollout_common(e, kind, md, packed_actor, a_std_log, priorK, n_steps, noise_seed, noise_epoch, state, action, noise,
                          reward, done, true, stream);
}

int pime_rollout_eval_supported(const pime_env* e, int32_t kind, int32_t md) {
    if (e == nullptr) return 0;
    if (e->cfg.state_mode != PIME_STATE_MIXED && e->cfg.state_mode != PIME_STATE_F64) return 0;
    if (e->cfg.kind != PIME_ENV_PH && e->cfg.num_stack != 0) {   // Stacking: the width-256 kernel only (plain actor, returns only)
        const int S = e->cfg.num_stack;
        return (md == 256 && kind == PIME_MLP_PLAIN_ACTOR && e->cfg.state_mode == PIME_STATE_MIXED && (S == 1 || S == 4 || S == 10)) ? 2 : 0;
    }   // (2: returns and trace, no set-point schedule -- the protocols of utils/test.py are written for the Integrator observation)
    if (kind == -1) return 1;                                            // the prior controller alone
    if (kind != PIME_MLP_PLAIN_ACTOR && kind != PIME_MLP_MODULAR_ACTOR) return 0;
    if (md == 256) return e->cfg.state_mode == PIME_STATE_MIXED ? 1 : 0;   // the streamed kernel's evaluation mode (float32 state)
    return (md == 64 || md == 128) && !family16(kind, md) ? 1 : 0;
}

int pime_rollout_eval(pime_env* e, int32_t kind, int32_t md, const float* packed_actor, const double* priorK, int32_t n_steps,
                      int32_t seg_len, const double* setpoints, int32_t n_setpoints, double* ret, double* trace,
                      pime_stream stream) {
    PIME_REQUIRE(e != nullptr, "NULL env handle");
    PIME_REQUIRE(pime_rollout_eval_supported(e, kind, md), "pime_rollout_eval: not served for this handle / actor kind %d width %d "
                 "(pime_rollout_eval_supported)", kind, md);
    PIME_REQUIRE(priorK && n_steps >= 1 && (kind == -1 || packed_actor) && (ret || trace), "pime_rollout_eval: bad arguments");
    PIME_REQUIRE(n_setpoints >= 0 && n_setpoints <= kMaxSetpoints, "pime_rollout_eval: n_setpoints %d (0 .. %d)", n_setpoints, kMaxSetpoints);
    PIME_REQUIRE(seg_len >= 0 && (seg_len == 0 || (setpoints && n_setpoints >= 1 && n_setpoints <= kMaxSetpoints &&
                                                   (n_steps + seg_len - 1) / seg_len <= n_setpoints)),
                 "pime_rollout_eval: the set-point schedule does not cover n_steps (at most %d segments)", kMaxSetpoints);
    if (!e->was_reset) { set_error("pime_rollout_eval before pime_env_reset"); return PIME_ERR_STATE; }
    if (int rc = use_device(e)) return rc;
    if (md == 256 && kind != -1) {   // the streamed 16-tile rollout kernel in evaluation mode (mlp16.hip)
        PIME_REQUIRE(seg_len == 0 || pime_rollout_eval_supported(e, kind, md) == 1, "pime_rollout_eval: no set-point schedule on a "
                     "Stacking observation (the protocols are defined on the Integrator observation)");
        RolloutArgs a{};
        a.env = e->cfg.kind == PIME_ENV_PH ? 0 : (e->cfg.num_stack == 0 ? 1 : 2);
        a.n = e->cfg.n_envs;
        a.env_offset = e->cfg.env_offset;
        if (a.env == 0) { a.p = e->ph; a.p.auto_reset = 0; a.st = e->ph32; }
        else { a.wp = e->wt; a.wp.auto_reset = 0; a.wst = e->wt32; }
        a.img = packed_actor;
        a.a_std_log = nullptr;                  // not read in evaluation mode (no exploration noise)
        for (int j = 0; j < e->obs_dim; ++j) a.K.k[j] = priorK[j];
        a.n_steps = n_steps; a.eval_mode = 1; a.ret = ret; a.trace = trace; a.seg_len = seg_len;
        if (seg_len > 0)
            for (int j = 0; j < n_setpoints && j < 16; ++j) a.setpoint[j] = setpoints[j];
        return launch_rollout(kind, md, a, static_cast<hipStream_t>(stream));
    }
    if (e->cfg.state_mode == PIME_STATE_F64)
        return rollout_eval_t<double>(e, e->ph64, e->wt64, kind, md, packed_actor, priorK, n_steps, seg_len, setpoints, n_setpoints,
                                      ret, trace, stream);
    return rollout_eval_t<float>(e, e->ph32, e->wt32, kind, md, packed_actor, priorK, n_steps, seg_len, setpoints, n_setpoints, ret,
                                 trace, stream);
}

int pime_rollout_offpolicy_supported(const pime_env* e, int32_t md) {
    if (e == nullptr || e->cfg.state_mode != PIME_STATE_MIXED) return 0;
    if (e->cfg.kind != PIME_ENV_PH && e->cfg.num_stack != 0) return 0;
    return (md == 64 || md == 128) && !family16(PIME_MLP_CRITIC, md) ? 1 : 0;
}

int pime_rollout_offpolicy(pime_env* e, int32_t md, const float* packed_actor, const double* priorK, float explore_noise,
                           float gamma, float reward_scale, int32_t n_steps, uint64_t noise_seed, uint32_t noise_epoch,
                           float* obs, float* ring_state, float* ring_other, int32_t slot0, int32_t slots, pime_stream stream) {
    PIME_REQUIRE(e != nullptr, "NULL env handle");
    PIME_REQUIRE(pime_rollout_offpolicy_supported(e, md), "pime_rollout_offpolicy: not served for this handle / width %d", md);
    PIME_REQUIRE(packed_actor && priorK && obs && ring_state && ring_other && n_steps >= 1 && slots >= 2 && slot0 >= 0 &&
                 slot0 < slots && n_steps <= slots, "pime_rollout_offpolicy: bad arguments");
    if (!e->was_reset) { set_error("pime_rollout_offpolicy before pime_env_reset"); return PIME_ERR_STATE; }
    if (int rc = use_device(e)) return rc;
    OffPolicyArgs a{};
    a.env = e->cfg.kind == PIME_ENV_PH ? 0 : 1;
    a.n = e->cfg.n_envs;
    a.env_offset = e->cfg.env_offset;
    if (a.env == 0) { a.p = e->ph; a.p.auto_reset = 1; a.st = e->ph32; }
    else { a.wp = e->wt; a.wp.auto_reset = 1; a.wst = e->wt32; }
    a.img = packed_actor;
    for (int j = 0; j < e->obs_dim; ++j) a.K.k[j] = priorK[j];
    a.explore_noise = explore_noise; a.gamma = gamma; a.reward_scale = reward_scale;
    a.n_steps = n_steps; a.noise_seed = noise_seed; a.noise_epoch = noise_epoch;
    a.obs = obs; a.ring_state = ring_state; a.ring_other = ring_other; a.slot0 = slot0; a.slots = slots;
    return launch_rollout_offpolicy(md, a, static_cast<hipStream_t>(stream));
}

int pime_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, float* step, pime_stream stream) {
    PIME_REQUIRE(param && grad && exp_avg && exp_avg_sq && step && n >= 1, "pime_adam_step: bad arguments");
    return launch_adam(param, const_cast<float*>(grad), exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, nullptr, nullptr, nullptr, 0, 1,
                       static_cast<hipStream_t>(stream));
}

static int check_net(const pime_ppo_net* n, bool actor) {
    PIME_REQUIRE(n != nullptr, "pime_ppo_minibatch_grad: NULL net");
    if (int rc = mlp_check(n->kind, n->D, n->Di, n->md)) return rc;
    PIME_REQUIRE(n->params && n->grads && n->img_fwd && n->img_bwd && n->workspace, "pime_ppo_net: NULL member");
    PIME_REQUIRE(actor == (n->kind != PIME_MLP_CRITIC), "pime_ppo_minibatch_grad: actor/critic kinds swapped");
    if (actor) PIME_REQUIRE(n->a_std_log && n->g_a_std_log, "pime_ppo_net (actor): NULL a_std_log / gradient");
    const int np = n->kind == PIME_MLP_MODULAR_ACTOR ? 12 : 8;
    for (int i = 0; i < np; ++i) PIME_REQUIRE(n->params[i] && n->grads[i], "pime_ppo_net: params/grads[%d] NULL", i);
    return PIME_OK;
}

// PIME_GRAD_BF16X3=1: the bf16x3 planes of the nets that use them, re-split from the parameters (after an optimizer step whose image
// map kept the f32 images current; pime_ppo_repack does it as part of launch_pack16).
static int refresh_b3(const pime_ppo_net* actor, const pime_ppo_net* critic, hipStream_t s) {
    const pime_ppo_net* nets[2] = {critic, actor};
    for (int k = 0; k < 2; ++k) {
        const pime_ppo_net* n = nets[k];
        if (!b3_grad(n->kind, n->md, n->D, n->Di)) continue;
        PackArgs pa{};
        const int np = n->kind == PIME_MLP_MODULAR_ACTOR ? 12 : 8;
        for (int i = 0; i < np; ++i) pa.p[i] = n->params[i];
        pa.kind = n->kind; pa.D = n->D; pa.Di = n->Di; pa.md = n->md;
        if (int rc = launch_pack16_b3(pa, const_cast<float*>(n->img_bwd), s)) return rc;
    }
    return PIME_OK;
}

int pime_ppo_repack(const pime_ppo_net* actor, const pime_ppo_net* critic, pime_stream stream) {
    if (int rc = check_net(actor, true)) return rc;
    if (int rc = check_net(critic, false)) return rc;
    PackArgs pa[2];
    const pime_ppo_net* nets[2] = {critic, actor};
    bool f16[2];
    for (int k = 0; k < 2; ++k) {
        const pime_ppo_net* n = nets[k];
        pa[k] = PackArgs{};
        const int np = n->kind == PIME_MLP_MODULAR_ACTOR ? 12 : 8;
        for (int i = 0; i < np; ++i) pa[k].p[i] = n->params[i];
        pa[k].kind = n->kind; pa[k].D = n->D; pa[k].Di = n->Di; pa[k].md = n->md;
        f16[k] = family16_grad(n->kind, n->md, n->D, n->Di);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!f16[0] && !f16[1])
        return launch_repack(pa[0], pa[1], const_cast<float*>(critic->img_fwd), const_cast<float*>(critic->img_bwd),
                             const_cast<float*>(actor->img_fwd), const_cast<float*>(actor->img_bwd), s);
    for (int k = 0; k < 2; ++k) {   // a net of the 16-tile family: one launch packs its forward and transposed images
        const pime_ppo_net* n = nets[k];
        if (f16[k]) {
            if (int rc = launch_pack16(pa[k], const_cast<float*>(n->img_fwd), const_cast<float*>(n->img_bwd), s)) return rc;
        } else {
            if (int rc = launch_mlp_pack(n->kind, n->D, n->Di, n->md, n->params, const_cast<float*>(n->img_fwd), s)) return rc;
            if (int rc = launch_pack_bwd(n->kind, n->D, n->Di, n->md, n->params, const_cast<float*>(n->img_bwd), s)) return rc;
        }
    }
    return PIME_OK;
}

// ---- image map: where every flat parameter element sits in the packed images ----------------------------------------------
__global__ void code_fill_kernel(float* codes, long long n, long long total) {
    for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < total; j += (long long)gridDim.x * blockDim.x)
        codes[j] = j < n ? (float)(j + 1) : 0.f;   // element j is coded j + 1 (exact in float32 below 2^24); 0 = padding / frozen
}
__global__ void image_invert_kernel(const float* img, long long floats, int32_t* map, int which, int net, int* dup) {
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < floats; e += (long long)gridDim.x * blockDim.x) {
        const float c = img[e];
        if (c >= 1.f) {
            const long long j = (long long)c - 1;
            if (atomicExch(&map[2 * j + which], (int32_t)e | (net << 28)) != -1) atomicAdd(dup, 1);   // an element packed twice: not a permutation
        }
    }
}

int pime_ppo_image_map(const pime_ppo_net* actor, const pime_ppo_net* critic, const float* flat_param, int64_t n,
                       int32_t* image_map, pime_stream stream) {
    if (int rc = check_net(actor, true)) return rc;
    if (int rc = check_net(critic, false)) return rc;
    PIME_REQUIRE(flat_param && image_map && n >= 1 && n < (1 << 24), "pime_ppo_image_map: bad flat tensor (1 <= n < 2^24)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const pime_ppo_net* nets[2] = {critic, actor};
    // index-coded stand-ins of the parameters: the same offsets inside `codes` as inside the flat tensor; parameters outside
    // the flat tensor (frozen ones) read the zero tail
    int64_t tail = 1;
    const float* fake[2][12];
    for (int k = 0; k < 2; ++k) {
        const pime_ppo_net* nk = nets[k];
        const int np = nk->kind == PIME_MLP_MODULAR_ACTOR ? 12 : 8;
        int poff[13], psize[12];
        slab_layout(nk->kind, nk->D, nk->Di, nk->md, poff, psize);   // psize: the parameters' element counts
        for (int i = 0; i < np; ++i) tail = psize[i] > tail ? psize[i] : tail;
    }
    float* codes = nullptr;
    PIME_HIP_TRY(hipMalloc(&codes, sizeof(float) * (size_t)(n + tail)));
    auto fail = [&](int rc) { (void)hipFree(codes); return rc; };
    for (int k = 0; k < 2; ++k) {
        const pime_ppo_net* nk = nets[k];
        const int np = nk->kind == PIME_MLP_MODULAR_ACTOR ? 12 : 8;
        int poff[13], psize[12];
        slab_layout(nk->kind, nk->D, nk->Di, nk->md, poff, psize);
        for (int i = 0; i < np; ++i) {
            const float* p = nk->params[i];
            const long long off = p - flat_param;
            if (p >= flat_param && p < flat_param + n) {
                if (off + psize[i] > n) {
                    set_error("pime_ppo_image_map: parameter %d of the %s straddles the end of the flat tensor", i, k ? "actor" : "critic");
                    return fail(PIME_ERR_ARG);
                }
                fake[k][i] = codes + off;
            } else {
                fake[k][i] = codes + n;
            }
        }
    }
    hipLaunchKernelGGL(code_fill_kernel, dim3(64), dim3(256), 0, s, codes, (long long)n, (long long)(n + tail));
    if (hipMemsetAsync(image_map, 0xff, sizeof(int32_t) * 2 * (size_t)n, s) != hipSuccess) return fail(PIME_ERR_DEVICE);
    int* dup = nullptr;
    if (hipMalloc(&dup, sizeof(int)) != hipSuccess) return fail(PIME_ERR_DEVICE);
    (void)hipMemsetAsync(dup, 0, sizeof(int), s);
    float* scratch[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int64_t floats[2][2];
    auto cleanup = [&](int rc) {
        for (int k = 0; k < 2; ++k)
            for (int w = 0; w < 2; ++w) (void)hipFree(scratch[k][w]);
        (void)hipFree(dup);
        return fail(rc);
    };
    pime_ppo_net coded[2];
    for (int k = 0; k < 2; ++k) {
        const pime_ppo_net* nk = nets[k];
        floats[k][0] = ppo_fwd_image_floats(nk->kind, nk->D, nk->Di, nk->md);
        floats[k][1] = ppo_bwd_image_f32_floats(nk->kind, nk->D, nk->Di, nk->md);   // the map covers the f32 images (bf16x3 planes behind
        const int64_t alloc[2] = {floats[k][0], pime_ppo_bwd_image_floats(nk->kind, nk->D, nk->Di, nk->md)};   // them are re-split per step)
        for (int w = 0; w < 2; ++w)
            if (floats[k][w] <= 0 || hipMalloc(&scratch[k][w], sizeof(float) * (size_t)alloc[w]) != hipSuccess ||
                hipMemsetAsync(scratch[k][w], 0, sizeof(float) * (size_t)alloc[w], s) != hipSuccess)   // padding the pack kernels
                return cleanup(PIME_ERR_DEVICE);                                                    // never write must read 0
        coded[k] = *nk;
        coded[k].params = fake[k];
        coded[k].img_fwd = scratch[k][0];
        coded[k].img_bwd = scratch[k][1];
    }
    if (int rc = pime_ppo_repack(&coded[1], &coded[0], stream)) return cleanup(rc);   // the library's own pack kernels, per family
    for (int k = 0; k < 2; ++k)
        for (int w = 0; w < 2; ++w)
            hipLaunchKernelGGL(image_invert_kernel, dim3(128), dim3(256), 0, s, scratch[k][w], (long long)floats[k][w], image_map, w, k, dup);
    int ndup = 0;
    if (hipMemcpyAsync(&ndup, dup, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return cleanup(PIME_ERR_DEVICE);
    if (ndup) {
        set_error("pime_ppo_image_map: %d parameter elements are packed more than once: the images are not permutations", ndup);
        return cleanup(PIME_ERR_ARG);
    }
    return cleanup(PIME_OK);
}

int pime_adam_step_images(const pime_adam* opt, const pime_ppo_net* actor, const pime_ppo_net* critic, pime_stream stream) {
    if (int rc = check_net(actor, true)) return rc;
    if (int rc = check_net(critic, false)) return rc;
    PIME_REQUIRE(opt && opt->param && opt->grad && opt->exp_avg && opt->exp_avg_sq && opt->step && opt->n >= 1 && opt->image_map,
                 "pime_adam_step_images: bad pime_adam (image_map must be set: pime_ppo_image_map)");
    float* const img[2][2] = {{const_cast<float*>(critic->img_fwd), const_cast<float*>(critic->img_bwd)},
                              {const_cast<float*>(actor->img_fwd), const_cast<float*>(actor->img_bwd)}};
    if (int rc = launch_adam(opt->param, opt->grad, opt->exp_avg, opt->exp_avg_sq, opt->n, opt->lr, opt->beta1, opt->beta2, opt->eps,
                             opt->step, opt->image_map, img, nullptr, 0, 1, static_cast<hipStream_t>(stream)))
        return rc;
    return refresh_b3(actor, critic, static_cast<hipStream_t>(stream));
}

int pime_adam_step_dp(const pime_adam* opt, const pime_ppo_net* actor, const pime_ppo_net* critic, pime_stream stream) {
    PIME_REQUIRE(opt && opt->param && opt->grad && opt->exp_avg && opt->exp_avg_sq && opt->step && opt->n >= 1, "pime_adam_step_dp: bad pime_adam");
    PIME_REQUIRE(opt->dp_moments && opt->dp_world >= 1 && opt->critic_offset >= 0 && opt->critic_offset <= opt->n,
                 "pime_adam_step_dp: dp_moments / dp_world / critic_offset");
    float* img[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    if (opt->image_map) {
        if (int rc = check_net(actor, true)) return rc;
        if (int rc = check_net(critic, false)) return rc;
        img[0][0] = const_cast<float*>(critic->img_fwd); img[0][1] = const_cast<float*>(critic->img_bwd);
        img[1][0] = const_cast<float*>(actor->img_fwd); img[1][1] = const_cast<float*>(actor->img_bwd);
    }
    if (int rc = launch_adam(opt->param, opt->grad, opt->exp_avg, opt->exp_avg_sq, opt->n, opt->lr, opt->beta1, opt->beta2, opt->eps,
                             opt->step, opt->image_map, img, opt->dp_moments, (long long)opt->critic_offset, opt->dp_world,
                             static_cast<hipStream_t>(stream)))
        return rc;
    return opt->image_map ? refresh_b3(actor, critic, static_cast<hipStream_t>(stream)) : PIME_OK;
}

static int minibatch_impl(const pime_ppo_net* actor, const pime_ppo_net* critic, const pime_ppo_batch* b, float ratio_clip,
                          float lambda_entropy, float* critic_scale, double* moments, float* loss_sums, const pime_adam* opt,
                          pime_stream stream) {
    if (int rc = check_net(actor, true)) return rc;
    if (int rc = check_net(critic, false)) return rc;
    PIME_REQUIRE(b && b->state && b->action && b->logprob && b->adv && b->r_sum && b->indices && b->B >= 1,
                 "pime_ppo_minibatch_grad: bad batch");
    PIME_REQUIRE(critic_scale && moments && loss_sums, "pime_ppo_minibatch_grad: NULL critic_scale / moments / loss_sums");
    PIME_REQUIRE(actor->D == critic->D, "actor and critic state_dim differ");
    hipStream_t s = static_cast<hipStream_t>(stream);
    DwArgs dw{};
    static const bool force_split = std::getenv("PIME_PPO_SPLIT") != nullptr;  // A/B knob: the net + dW kernel pipeline
    // Which kernel serves which net (index 0 = critic, 1 = actor):
    //   F16   the streamed 16-tile family (mlp16.hip): width 256, and 64 / 128 on observations too wide for FUSED; slabs
    //   FUSED the LDS-resident kernel (ppo_fused.hip): 64 / 128 while its LDS map fits; slabs
    //   SPLIT the net + dW pipeline (ppo_train.hip, float atomics): a modular actor on a wide observation, or PIME_PPO_SPLIT
    // Slab nets are finished by ppo_grad_reduce_kernel (which also derives the critic scale); a split critic by critic_scale_kernel.
    enum { F16, FUSED, SPLIT };
    const pime_ppo_net* nets[2] = {critic, actor};
    int mode[2];
    for (int k = 0; k < 2; ++k) {
        const pime_ppo_net* n = nets[k];
        if (family16_grad(n->kind, n->md, n->D, n->Di)) mode[k] = F16;
        else mode[k] = (force_split || !fused_fits(n->kind, n->D, n->Di, n->md)) ? SPLIT : FUSED;
    }
    const bool any_split = mode[0] == SPLIT || mode[1] == SPLIT, any_slab = mode[0] != SPLIT || mode[1] != SPLIT;
    PIME_REQUIRE(!(b->dp_moments && mode[0] == SPLIT), "pime_ppo_minibatch_grad: dp_moments needs the critic on a slab kernel (it takes the split pipeline here)");
    PIME_REQUIRE(!(b->dp_moments && opt), "pime_ppo_minibatch_step: dp_moments with a fused optimizer step (the all-reduce has to come first)");
    ReduceAdam adam{};
    if (opt) {
        PIME_REQUIRE(!any_split, "pime_ppo_minibatch_step: the optimizer step is fused into the slab reduction, which these nets "
                     "do not use (split pipeline): call pime_ppo_minibatch_grad + pime_adam_step");
        PIME_REQUIRE(opt->param && opt->grad && opt->exp_avg && opt->exp_avg_sq && opt->step && opt->n >= 1,
                     "pime_ppo_minibatch_step: bad pime_adam");
        adam = ReduceAdam{opt->grad, opt->param, opt->exp_avg, opt->exp_avg_sq, opt->step, (long long)opt->n, opt->lr, opt->beta1,
                          opt->beta2, opt->eps, opt->image_map,
                          {{const_cast<float*>(critic->img_fwd), const_cast<float*>(critic->img_bwd)},
                           {const_cast<float*>(actor->img_fwd), const_cast<float*>(actor->img_bwd)}}};
    }
    static const bool tracing = std::getenv("PIME_FUSED_TRACE") != nullptr;  // tuning aid: phase marks of one workgroup
    static long long* trace_dev = nullptr;
    if (tracing && !trace_dev) {
        PIME_HIP_TRY(hipMalloc(&trace_dev, (2 * 64 + 2 * 1024) * sizeof(long long)));   // marks of one workgroup + [net][workgroup][start, end]
    }
    if (tracing) PIME_HIP_TRY(hipMemsetAsync(trace_dev, 0, (2 * 64 + 2 * 1024) * sizeof(long long), s));
    if (mode[0] == SPLIT) PIME_HIP_TRY(hipMemsetAsync(moments, 0, 2 * sizeof(double), s));  // atomics accumulate into it
    if (b->flags & PIME_PPO_OVERWRITE_GRADS) {   // the atomics of the split pipeline need zeroed targets
        for (int k = 0; k < 2; ++k) {
            if (mode[k] != SPLIT) continue;
            const pime_ppo_net* n = nets[k];
            int poff[13], psize[12];
            slab_layout(n->kind, n->D, n->Di, n->md, poff, psize);
            const int np = n->kind == PIME_MLP_MODULAR_ACTOR ? 12 : 8;
            for (int i = 0; i < np; ++i) PIME_HIP_TRY(hipMemsetAsync(n->grads[i], 0, sizeof(float) * psize[i], s));
            if (k == 1) PIME_HIP_TRY(hipMemsetAsync(actor->g_a_std_log, 0, sizeof(float), s));
        }
    }
    PpoArgs slab_args[2];
    // both nets on the LDS-resident fused kernel and of one width: ONE launch serves them (ppo_fused_dual_kernel); PIME_PPO_DUAL=0
    // keeps one launch per net (A/B, and the per-net phase trace)
    static const bool dual_off = std::getenv("PIME_PPO_DUAL") != nullptr && std::atoi(std::getenv("PIME_PPO_DUAL")) == 0;
    const bool dual = mode[0] == FUSED && mode[1] == FUSED && critic->md == actor->md && !dual_off && !tracing;
    for (int k = 0; k < 2; ++k) {
        const pime_ppo_net* n = nets[k];
        PpoArgs a{};
        a.state = b->state; a.action = b->action; a.logprob = b->logprob; a.adv = b->adv; a.r_sum = b->r_sum;
        a.indices = b->indices; a.index_row = b->index_row; a.B = b->B; a.D = n->D; a.Di = n->Di;
        a.a_std_log = n->a_std_log; a.moments = moments; a.ratio_clip = ratio_clip; a.lambda_entropy = lambda_entropy;
        a.img_fwd = n->img_fwd; a.img_bwd = n->img_bwd;
        const int64_t ntiles = (b->B + 31) / 32;
        a.stash = n->workspace;
        a.dout = n->workspace + ntiles * (int64_t)stash_tiles_of(n->kind, n->md) * 1024;
        a.xg = a.dout + ntiles * 32;
        a.loss_sums = loss_sums; a.g_std = n->g_a_std_log;
        a.stagger = 3;  // measured best of 0..4 on MI355X (532 -> 522 us per minibatch gradient)
        if (const char* e = std::getenv("PIME_STAGGER")) a.stagger = std::atoi(e);  // tuning knob
        const int np = n->kind == PIME_MLP_MODULAR_ACTOR ? 12 : 8;
        for (int i = 0; i < np; ++i) a.grad[i] = n->grads[i];
        a.trace = (tracing && mode[k] != SPLIT) ? trace_dev + 64 * k : nullptr;
        a.trace_wg = tracing ? std::atoi(std::getenv("PIME_FUSED_TRACE")) : 0;
        a.trace_span = a.trace ? trace_dev + 128 + 1024 * k : nullptr;
        int psize[12];
        if (mode[k] == F16) {
            a.slab = n->workspace;
            a.slab_stride = n->kind == PIME_MLP_MODULAR_ACTOR ? slab_layout16m(n->md, a.poff, psize) : slab_layout16(n->D, n->md, a.poff, psize);
            slab_args[k] = a;
            if (int rc = launch_ppo16(n->kind, n->md, a, s)) return rc;
        } else if (mode[k] == FUSED) {
            a.slab = n->workspace + fused_stash_floats(n->kind, b->B, n->md);
            a.slab_stride = slab_layout(n->kind, n->D, n->Di, n->md, a.poff, psize);
            slab_args[k] = a;
            if (dual) {
                if (k == 1)
                    if (int rc = launch_ppo_fused_dual(actor->kind, actor->md, slab_args[1], slab_args[0], s)) return rc;
            } else if (int rc = launch_ppo_fused(n->kind, n->md, a, s)) return rc;
        } else {
            slab_args[k] = a;
            if (int rc = launch_ppo_net(n->kind, n->md, a, s)) return rc;
            dw.njobs += build_dw_jobs(n->kind, n->md, a, n->params, n->grads, dw.job + dw.njobs);
        }
    }
    if (any_split) {
        dw.tiles_per_wg = 16;
        if (const char* e = std::getenv("PIME_DW_DEBUG")) dw.debug_skip = std::atoi(e);  // timing ablations only
        if (int rc = launch_dw(dw, b->B, s)) return rc;
    }
    if (tracing && any_slab) {
        static long long t[128 + 2048];
        PIME_HIP_TRY(hipStreamSynchronize(s));
        PIME_HIP_TRY(hipMemcpy(t, trace_dev, sizeof(t), hipMemcpyDeviceToHost));
        for (int k = 0; k < 2; ++k) {
            std::fprintf(stderr, "[pime trace] %s:", k ? "actor " : "critic");
            for (int i = 1; i < 32; ++i)
                if (t[64 * k + i]) std::fprintf(stderr, " m%d=%.1f", i, (double)(t[64 * k + i] - t[64 * k]) * 0.01);
            for (int i = 32; i < 40; ++i)
                if (t[64 * k + i]) std::fprintf(stderr, " c%d=%lld", i - 32, t[64 * k + i]);
            for (int i = 40; i < 56; ++i)   // sub-marks of one job (16-tile family: the passes of the first sliced weight gradient)
                if (t[64 * k + i]) std::fprintf(stderr, " p%d=%.1f", i - 40, (double)(t[64 * k + i] - t[64 * k]) * 0.01);
            std::fprintf(stderr, "\n");
            // every workgroup's start / end (100 MHz wall clock): launch skew, the slowest workgroup, the whole span
            const long long* sp = t + 128 + 1024 * k;
            long long s0 = 0, s1 = 0, e0 = 0, e1 = 0, dmin = 0, dmax = 0;
            int n = 0;
            for (int w = 0; w < 512; ++w) {
                if (!sp[2 * w] || !sp[2 * w + 1]) continue;
                const long long st = sp[2 * w], en = sp[2 * w + 1], d = en - st;
                if (!n) { s0 = s1 = st; e0 = e1 = en; dmin = dmax = d; }
                s0 = st < s0 ? st : s0; s1 = st > s1 ? st : s1; e0 = en < e0 ? en : e0; e1 = en > e1 ? en : e1;
                dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax;
                ++n;
            }
            if (n)
                std::fprintf(stderr, "[pime trace] %s: %d workgroups: starts spread %.1f us, per-workgroup time %.1f .. %.1f us, "
                             "first start -> last end %.1f us\n", k ? "actor " : "critic", n, (s1 - s0) * 0.01, dmin * 0.01, dmax * 0.01,
                             (e1 - s0) * 0.01);
        }
    }
    if (any_slab) {
        const int nslabs[2] = {mode[0] == F16 ? grid16(critic->kind, b->B, critic->md, critic->D, critic->Di) : fused_grid(b->B),
                               mode[1] == F16 ? grid16(actor->kind, b->B, actor->md, actor->D, actor->Di) : fused_grid(b->B)};
        if (int rc = launch_grad_reduce(slab_args[0], slab_args[1], critic->kind, critic->md, actor->kind, actor->md,
                                        mode[0] == F16, mode[1] == F16, mode[0] != SPLIT, mode[1] != SPLIT, nslabs[0], nslabs[1],
                                        critic->grads, actor->grads, actor->g_a_std_log, critic_scale, moments, loss_sums + 3,
                                        b->flags & PIME_PPO_OVERWRITE_GRADS, mode[0] != SPLIT ? b->index_row : nullptr,
                                        opt ? &adam : nullptr, b->dp_moments, s))
            return rc;
        if (opt && opt->image_map)   // (without a map the caller re-packs: pime_ppo_repack splits the planes too)
            if (int rc = refresh_b3(actor, critic, s)) return rc;
    }
    if (mode[0] == SPLIT)   // scales the split critic's gradients, advances the index-table row
        return launch_critic_scale(critic->D, critic->md, critic->grads, moments, b->B, critic_scale, loss_sums + 3,
                                   b->index_row, s);
    return PIME_OK;
}

int pime_ppo_minibatch_grad(const pime_ppo_net* actor, const pime_ppo_net* critic, const pime_ppo_batch* b,
                            float ratio_clip, float lambda_entropy, float* critic_scale, double* moments,
                            float* loss_sums, pime_stream stream) {
    return minibatch_impl(actor, critic, b, ratio_clip, lambda_entropy, critic_scale, moments, loss_sums, nullptr, stream);
}

int pime_ppo_minibatch_step(const pime_ppo_net* actor, const pime_ppo_net* critic, const pime_ppo_batch* b,
                            float ratio_clip, float lambda_entropy, float* critic_scale, double* moments,
                            float* loss_sums, const pime_adam* opt, pime_stream stream) {
    PIME_REQUIRE(opt != nullptr, "pime_ppo_minibatch_step: NULL pime_adam");
    return minibatch_impl(actor, critic, b, ratio_clip, lambda_entropy, critic_scale, moments, loss_sums, opt, stream);
}

// -- fused TD3 optimizer step (csrc/td3_fused.hip) -----------------------------------------------------------------------
int pime_td3_supported(int32_t D, int32_t action_dim, int32_t md) { return td3_supported(D, action_dim, md) ? 1 : 0; }

int64_t pime_td3_param_floats(int32_t which, int32_t D, int32_t md) {
    if (!td3_supported(D, 1, md) || which < 0 || which > 1) {
        set_error("pime_td3_param_floats: unsupported net (which %d, D %d, width %d)", which, D, md);
        return -1;
    }
    return which == 0 ? td3_actor_off(D, md).total : td3_critic_off(D, md).total;
}

int pime_td3_param_offsets(int32_t which, int32_t D, int32_t md, int32_t* offsets) {
    PIME_REQUIRE(offsets && td3_supported(D, 1, md) && (which == 0 || which == 1), "pime_td3_param_offsets: bad arguments");
    if (which == 0) {
        const Td3ActorOff o = td3_actor_off(D, md);
        const int v[8] = {o.W1, o.b1, o.W2, o.b2, o.W3, o.b3, o.w4, o.b4};
        for (int i = 0; i < 8; ++i) offsets[i] = v[i];
    } else {
        const Td3CriticOff o = td3_critic_off(D, md);
        const int v[8] = {o.W1, o.b1, o.W2, o.b2, o.q1w, o.q1b, o.q2w, o.q2b};
        for (int i = 0; i < 8; ++i) offsets[i] = v[i];
    }
    return PIME_OK;
}

int64_t pime_td3_workspace_floats(int32_t D, int32_t md, int32_t B) {
    if (!td3_supported(D, 1, md) || B < 1) {
        set_error("pime_td3_workspace_floats: unsupported shape (D %d, width %d, batch %d)", D, md, B);
        return -1;
    }
    return td3_workspace_floats(D, md, B);
}

static int check_td3_net(const pime_td3_net* n, const char* what) {
    PIME_REQUIRE(n && n->param && n->target && n->grad && n->exp_avg && n->exp_avg_sq && n->step, "pime_td3_step: NULL pointer in the %s's pime_td3_net", what);
    PIME_REQUIRE(n->lr > 0.f && n->beta1 >= 0.f && n->beta1 < 1.f && n->beta2 >= 0.f && n->beta2 < 1.f && n->eps > 0.f,
                 "pime_td3_step: bad Adam constants for the %s", what);
    return PIME_OK;
}

int pime_td3_step(int32_t D, int32_t md, const pime_td3_net* actor, const pime_td3_net* critic, const pime_td3_batch* b, float tau,
                  int32_t update_freq, int32_t soft_mode, int32_t phases, float* workspace, float* loss, pime_stream stream) {
    PIME_REQUIRE(td3_supported(D, 1, md), "pime_td3_step: no kernel for state_dim %d width %d (D <= %d, width 64 | 128)", D, md, kTd3MaxD);
    if (int rc = check_td3_net(actor, "actor")) return rc;
    if (int rc = check_td3_net(critic, "critic")) return rc;
    PIME_REQUIRE(b && b->state && b->other && b->idx && b->nxt && b->B >= 1, "pime_td3_step: bad pime_td3_batch");
    PIME_REQUIRE(workspace != nullptr, "pime_td3_step: NULL workspace");
    PIME_REQUIRE(soft_mode >= 0 && soft_mode <= 2 && (soft_mode != 2 || update_freq >= 1), "pime_td3_step: soft_mode %d / update_freq %d", soft_mode, update_freq);
    PIME_REQUIRE(b->row >= 0, "pime_td3_step: table row %lld", (long long)b->row);
    const int soft = soft_mode == 1 || (soft_mode == 2 && b->row % update_freq == 0);
    PIME_REQUIRE(phases >= 1 && phases <= 255 && !((phases & 2) && (phases & 16)) && !((phases & 8) && (phases & 64)),
                 "pime_td3_step: phases %d", phases);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = td3_grid(b->B), ngroups = (b->B + 15) / 16;
    const Td3SlabLayout LA = td3_actor_slab(D, md), LC = td3_critic_slab(D, md);
    float* const slab_c = workspace;
    float* const slab_a = workspace + (size_t)grid * LC.stride;
    // [2][B][8] gathered rows, by row parity: written by the critic launch of a row, read by its actor launch -- which may still run
    // while the critic launch of the NEXT row (other parity) gathers (the caller's two-stream schedule, see include/pime_hip.h)
    float* const xg = slab_a + (size_t)grid * LA.stride + (size_t)(b->row & 1) * b->B * 8;
    Td3Batch tb{b->state, b->other, b->idx, b->nxt, b->noise, (long long)b->row, b->epoch, b->B, b->noise_seed, b->noise_epoch, b->policy_noise, b->noise_clip};
    auto apply = [&](const pime_td3_net* n, const Td3SlabLayout& L, const float* slab, int slot, int mode) {
        Td3ApplyArgs a{};
        a.L = L; a.slab = slab; a.nslabs = grid; a.mode = mode;
        a.param = n->param; a.target = n->target; a.grad = n->grad; a.exp_avg = n->exp_avg; a.exp_avg_sq = n->exp_avg_sq; a.step = n->step;
        a.lr = n->lr; a.b1 = n->beta1; a.b2 = n->beta2; a.eps = n->eps; a.tau = tau;
        a.row = (long long)b->row; a.soft = soft;
        a.loss = loss; a.loss_slot = slot; a.inv_B = 1.0f / (float)b->B;
        return launch_td3_apply(a, s);
    };
    static const bool tracing = std::getenv("PIME_TD3_TRACE") != nullptr;   // tuning aid: phase marks of workgroup 0 (synchronises)
    static long long* trace_dev = nullptr;
    if (tracing && !trace_dev) PIME_HIP_TRY(hipMalloc(&trace_dev, 64 * sizeof(long long)));
    if (tracing) PIME_HIP_TRY(hipMemsetAsync(trace_dev, 0, 64 * sizeof(long long), s));
    if (phases & 1) {
        Td3GradArgs g{tb, D, actor->target, critic->param, critic->target, slab_c, xg, LC.stride, ngroups, tracing ? trace_dev : nullptr};
        if (int rc = launch_td3_grad(true, md, g, grid, s)) return rc;
    }
    if (phases & (2 | 16))
        if (int rc = apply(critic, LC, slab_c, 1, (phases & 16) ? 1 : 0)) return rc;
    if (phases & 32)
        if (int rc = apply(critic, LC, slab_c, 1, 2)) return rc;
    if (phases & 4) {
        Td3GradArgs g{tb, D, actor->param, critic->target, nullptr, slab_a, xg, LA.stride, ngroups, tracing ? trace_dev + 32 : nullptr};
        if (int rc = launch_td3_grad(false, md, g, grid, s)) return rc;
    }
    if (phases & (8 | 64))
        if (int rc = apply(actor, LA, slab_a, 0, (phases & 64) ? 1 : 0)) return rc;
    if (phases & 128)
        if (int rc = apply(actor, LA, slab_a, 0, 2)) return rc;
    if (tracing) {
        long long t[64];
        PIME_HIP_TRY(hipStreamSynchronize(s));
        PIME_HIP_TRY(hipMemcpy(t, trace_dev, sizeof(t), hipMemcpyDeviceToHost));
        for (int k = 0; k < 2; ++k) {
            std::fprintf(stderr, "[pime td3 trace] %s:", k ? "actor " : "critic");
            for (int i = 1; i < 30; ++i)
                if (t[32 * k + i]) std::fprintf(stderr, " m%d=%.2f", i, (double)(t[32 * k + i] - t[32 * k]) * 0.01);
            if (t[32 * k + 30]) std::fprintf(stderr, " shader_cycles=%lld", t[32 * k + 30]);   // s_memtime ticks first mark -> last mark
            std::fprintf(stderr, "\n");
        }
    }
    return PIME_OK;
}

}  // extern "C"
