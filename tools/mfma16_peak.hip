// What v_mfma_f32_16x16x4_f32 sustains per wave and per SIMD, in the issue patterns of csrc/mlp16.hip:
//   chain  : NACC independent 4-register accumulators, register operands (NACC = 8: one k-step of a width-128 layer)
//   lds    : the same with the A operands read from LDS by ds_read_b128 one group (16 MFMAs) ahead
// hipcc --offload-arch=gfx950 -O3 tools/mfma16_peak.hip -o /tmp/mfma16_peak && /tmp/mfma16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC>
__global__ void chain16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void lds16(float* out, int iters, float b) {
    __shared__ __attribute__((aligned(16))) float w[16 * 2 * 64 * 4];   // 16 k-steps x 2 quads x 64 lanes x float4 = 32 KB
    for (int i = threadIdx.x; i < 16 * 2 * 64 * 4; i += blockDim.x) w[i] = 1e-30f * i;
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = (float)(threadIdx.x + i);
    const float4* wl = reinterpret_cast<const float4*>(w) + (threadIdx.x & 63);
    float4 wf[2][4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) wf[0][m] = wl[m * 64];
#pragma unroll
        for (int g = 0; g < 8; ++g) {   // 8 groups of 2 k-steps = 16 MFMAs
            const int cur = g & 1;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int m = half * 2; m < half * 2 + 2; ++m) {
                    const int q = m & 1;
                    const float4 x = wf[cur][m];
                    acc[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, b, acc[4 * q + 0], 0, 0, 0);
                    acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, b, acc[4 * q + 1], 0, 0, 0);
                    acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, b, acc[4 * q + 2], 0, 0, 0);
                    acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, b, acc[4 * q + 3], 0, 0, 0);
                }
                if (half == 0 && g + 1 < 8) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) wf[cur ^ 1][m] = wl[((g + 1) * 4 + m) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static void run(const char* name, F launch, double mfma_per_wave_iter, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4, grid = 256 * waves_per_simd;   // 256-thread workgroups, 1 / 2 / 4 per CU
        launch(grid, threads, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch(grid, threads, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * (threads / 64) * iters * mfma_per_wave_iter * 2048.0;
        const double cyc = ms * 1e-3 * 2.4e9 / (iters * mfma_per_wave_iter * waves_per_simd);
        printf("%-10s %d wave(s)/SIMD: %.3f ms, %.1f TFLOP/s, %.1f cycles@2.4GHz per MFMA per SIMD\n", name, waves_per_simd, ms,
               flops / ms / 1e9, cyc);
    }
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 1024 * 1024);
    const int iters = 4000;
    run("chain<8>", [&](int g, int t, int it) { chain16<8><<<g, t>>>(out, it, 1e-30f, 1e-30f); }, 64, iters);
    run("chain<16>", [&](int g, int t, int it) { chain16<16><<<g, t>>>(out, it, 1e-30f, 1e-30f); }, 128, iters);
    run("chain<2>", [&](int g, int t, int it) { chain16<2><<<g, t>>>(out, it, 1e-30f, 1e-30f); }, 16, iters);
    run("lds<8>", [&](int g, int t, int it) { lds16<<<g, t>>>(out, it, 1e-30f); }, 128, iters);
    return 0;
}
