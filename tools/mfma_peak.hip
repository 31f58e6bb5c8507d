// What the f32 matrix pipe sustains on this chip: waves that do nothing but dependent chains of
// v_mfma_f32_32x32x2_f32 on 4 independent accumulators (the issue pattern of the MLP layers), at 1, 2 and 4 waves per
// SIMD.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void mfma_chain(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd, grid = 256;
        mfma_chain<<<grid, threads>>>(out, 10, 1e-30f, 1e-30f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        mfma_chain<<<grid, threads>>>(out, iters, 1e-30f, 1e-30f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * (threads / 64) * iters * 64.0 * 4096.0;
        printf("%d wave(s)/SIMD: %.3f ms, %.1f TFLOP/s\n", waves_per_simd, ms, flops / ms / 1e9);
    }
    return 0;
}
