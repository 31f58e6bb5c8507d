// Fused small-MLP forwards on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
// replaces (reference, /root/reference):
//   CriticAdv value pass over the whole buffer      elegantrl/agent.py:619-620 with elegantrl/net.py:274-277
//   ActorResidualIntegratorModularPPO mean          elegantrl/net_residual.py:153-160,172-176
//   ActorResidualPPO / ActorPPO mean                elegantrl/net_residual.py:19-22,45-48
//
// Design (DESIGN.md "mlp_forward"):
//  * One wave owns a tile of 32 samples for the WHOLE network.  Activations never leave registers: with the
//    weights as the MFMA A operand (rows = output features) and the activations as B (columns = samples), the
//    32x32 accumulator of layer l has its sample on the lane and its features in the 16 registers, which is
//    exactly the B-operand shape of layer l+1's k-steps (k = lane>>5 picks the feature pair {f, f+4}).  So
//    layer chaining needs no LDS round trip and no cross-lane traffic; only the head's 2-way lane-half sum
//    uses one permute.
//  * All weights of the net (<= 132 KB at width 128) sit in LDS for the lifetime of a persistent workgroup,
//    pre-permuted by mlp_pack_kernel into the order the k-steps consume them, so the per-k-step A operands of
//    all output tiles are ONE conflict-free ds_read_b128 (or b64) per lane.
//  * First layer (K = state_dim <= 32) and head (N = 1) are VALU work in the same register layout.
#include "mlp_device.hpp"

namespace pime {

// ---- pack: nn.Linear layout -> packed image -------------------------------------------------------------------
__global__ void mlp_pack_kernel(PackArgs a, float* __restrict__ out) {
    pack_forward_image(a, out, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

constexpr int kMlpThreads = 512;  // 8 waves: two per SIMD so one wave's tanh/VALU overlaps the other's MFMAs

template <int T, int KIND>
__global__ __launch_bounds__(kMlpThreads) void mlp_forward_kernel(const float* __restrict__ x, int M, int D, int Di,
                                                                  const float* __restrict__ packed,
                                                                  float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MlpLayout L = mlp_layout(KIND, D, Di, T * 32);
    stage_image(lds, packed, L.total / 4);  // straight 16-B copies, coalesced in HBM and conflict-free in LDS
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6, h = lane >> 5;
    const int ntiles = (M + 31) / 32;
    for (int tile = blockIdx.x * waves + wave; tile < ntiles; tile += gridDim.x * waves) {
        PIME_NO_HOIST();
        const int m = tile * 32 + (lane & 31);
        const float* xrow = x + (size_t)(m < M ? m : M - 1) * D;
        float y;
        if constexpr (KIND == MLP_MODULAR_ACTOR) {
            constexpr int H = T / 2;
            const int Do = D - Di;
            f32x16 cat[T];  // torch.cat([other_net(..), integrator_net(..)], -1): tiles [0,H) and [H,T)
            {
                f32x16 a0[T];
                layer_first<T, 2>(lds + L.off[0], xrow, Do, h, a0);   // activations are applied by the consuming layer
                PIME_NO_HOIST();
                layer_mfma_in<T, H, 2, 1>(lds + L.off[1], lds + L.off[2], lane, a0, *reinterpret_cast<f32x16(*)[H]>(&cat[0]));
            }
            {
                f32x16 a0[T];
                PIME_NO_HOIST();
                layer_first<T, 2>(lds + L.off[3], xrow + Do, Di, h, a0);
                PIME_NO_HOIST();
                layer_mfma_in<T, H, 2, 1>(lds + L.off[4], lds + L.off[5], lane, a0, *reinterpret_cast<f32x16(*)[H]>(&cat[H]));
            }
            f32x16 n0[T];
            PIME_NO_HOIST();
            layer_mfma_in<T, T, 1, 1>(lds + L.off[6], lds + L.off[7], lane, cat, n0);
            PIME_NO_HOIST();
            y = layer_head<T>(lds + L.off[8], lds[L.off[9]], lane, n0);
        } else {
            constexpr int ACT = KIND == MLP_CRITIC ? 0 : 1;
            f32x16 a0[T], a1[T];
            layer_first<T, 2>(lds + L.off[0], xrow, D, h, a0);
            PIME_NO_HOIST();
            layer_mfma_in<T, T, 2, ACT>(lds + L.off[1], lds + L.off[2], lane, a0, a1);
            PIME_NO_HOIST();
            layer_mfma_in<T, T, ACT, ACT>(lds + L.off[3], lds + L.off[4], lane, a1, a0);
            PIME_NO_HOIST();
            y = layer_head<T>(lds + L.off[5], lds[L.off[6]], lane, a0);
        }
        if (h == 0 && m < M) out[m] = y;
    }
}

// ---- host launchers ---------------------------------------------------------------------------------------------
bool family16(int kind, int md);
int64_t packed16_floats(int kind, int D, int Di, int md);
int launch_pack16(const PackArgs& a, float* fwd, float* bwd, hipStream_t s);
int launch_forward16(int kind, const float* x, int M, int D, int Di, int md, const float* img, float* out, hipStream_t s);

int64_t mlp_packed_floats(int kind, int D, int Di, int md) {
    return family16(kind, md) ? packed16_floats(kind, D, Di, md) : (int64_t)mlp_layout(kind, D, Di, md).total;
}

int mlp_check(int kind, int D, int Di, int md) {
    PIME_REQUIRE(kind >= MLP_CRITIC && kind <= MLP_MODULAR_ACTOR, "mlp kind %d unknown", kind);
    PIME_REQUIRE(D >= 1 && D <= kMaxObsDim, "state_dim %d out of range [1,%d]", D, kMaxObsDim);
    if (kind == MLP_MODULAR_ACTOR) {
        PIME_REQUIRE(Di >= 1 && Di < D, "integrator_dim %d must be in [1, state_dim)", Di);
        // 64 / 128: LDS-resident (this file, ppo_fused.hip); 256: the streamed 16x16x4 family (mlp16.hip: mlp16m_forward_kernel, ppo16m_kernel)
        PIME_REQUIRE(md == 64 || md == 128 || md == 256, "fused modular-actor kernels support width 64, 128 or 256, got %d", md);
    } else {
        // 64 / 128: the LDS-resident 32x32x2 family (this file, ppo_fused.hip); 256: the streamed 16x16x4 family (mlp16.hip)
        PIME_REQUIRE(md == 64 || md == 128 || md == 256, "fused MLP kernels support width 64, 128 or 256, got %d", md);
    }
    return PIME_OK;
}

int launch_mlp_pack(int kind, int D, int Di, int md, const float* const* params, float* packed, hipStream_t s) {
    if (int rc = mlp_check(kind, D, Di, md)) return rc;
    PackArgs a{};
    const int np = kind == MLP_MODULAR_ACTOR ? 12 : 8;
    for (int i = 0; i < np; ++i) {
        PIME_REQUIRE(params[i] != nullptr, "mlp params[%d] is NULL", i);
        a.p[i] = params[i];
    }
    a.kind = kind; a.D = D; a.Di = Di; a.md = md;
    if (family16(kind, md)) return launch_pack16(a, packed, nullptr, s);
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(64), dim3(256), 0, s, a, packed);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

template <int T, int KIND>
static int launch_fwd(const float* x, int M, int D, int Di, const float* packed, float* out, hipStream_t s) {
    const MlpLayout L = mlp_layout(KIND, D, Di, T * 32);
    const size_t lds_bytes = (size_t)L.total * sizeof(float);
    static LdsLimit lds_limit;  // per instantiation
    PIME_RAISE_LDS(lds_limit, (mlp_forward_kernel<T, KIND>), 160 * 1024);
    PIME_REQUIRE(lds_bytes <= 160 * 1024, "packed MLP image (%zu B) exceeds the 160 KB LDS", lds_bytes);
    const int ntiles = (M + 31) / 32, waves = kMlpThreads / 64;
    int grid = (ntiles + waves - 1) / waves;
    // small launches: spread the tiles over more CUs (one tile per wave costs ~3 x 16k cycles of MFMA issue)
    if (ntiles <= 256 * 2) grid = ntiles < 256 ? ntiles : 256;
    if (grid > 256) grid = 256;  // persistent: one workgroup per CU, weights stay in its LDS
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((mlp_forward_kernel<T, KIND>), dim3(grid), dim3(kMlpThreads), lds_bytes, s, x, M, D, Di, packed,
                       out);
    PIME_HIP_TRY(hipGetLastError());
    return PIME_OK;
}

int launch_mlp_forward(int kind, const float* x, int M, int D, int Di, int md, const float* packed, float* out,
                       hipStream_t s) {
    if (int rc = mlp_check(kind, D, Di, md)) return rc;
    PIME_REQUIRE(M >= 1, "M = %d rows", M);
    if (family16(kind, md)) return launch_forward16(kind, x, M, D, Di, md, packed, out, s);
    const int T = md / 32;
#define PIME_FWD(TT, KK) \
    if (T == TT && kind == KK) return launch_fwd<TT, KK>(x, M, D, Di, packed, out, s);
    PIME_FWD(2, MLP_CRITIC) PIME_FWD(4, MLP_CRITIC)
    PIME_FWD(2, MLP_PLAIN_ACTOR) PIME_FWD(4, MLP_PLAIN_ACTOR)
    PIME_FWD(2, MLP_MODULAR_ACTOR) PIME_FWD(4, MLP_MODULAR_ACTOR)
#undef PIME_FWD
    set_error("no fused MLP instantiation for kind %d width %d", kind, md);
    return PIME_ERR_ARG;
}

}  // namespace pime
