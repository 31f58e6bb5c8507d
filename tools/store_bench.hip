// How fast can a CU issue 1 KB stores (global_store_dwordx4, one wave writes 1 KB contiguous)?  4 waves per workgroup, one per
// SIMD, as the width-256 weight-gradient passes of csrc/mlp16.hip: bursts of 16 stores per wave, then `gap` iterations of VALU work.
// hipcc --offload-arch=gfx950 -O3 tools/store_bench.hip -o /tmp/store_bench && /tmp/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int BURST>
__global__ __launch_bounds__(256) void stores(float* out, int iters, int gap, size_t wg_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = out + blockIdx.x * wg_stride + wave * (BURST * 256);
    f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < BURST; ++k) *reinterpret_cast<f32x4*>(base + (size_t)(it & 15) * 4 * BURST * 256 + k * 256 + lane * 4) = v;
        for (int g = 0; g < gap; ++g) { acc = fmaf(acc, 1.0001f, v.x); asm volatile("" : "+v"(acc)); }
        v.w += 1.f;
    }
    if (acc == 12345.f) out[0] = acc;
}

int main() {
    const size_t wg_stride = (size_t)16 * 4 * 16 * 256;   // floats per workgroup: 16 slots x 4 waves x 16 stores x 1 KB
    float* out;
    hipMalloc(&out, sizeof(float) * wg_stride * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 32})
        for (int gap : {0, 2000, 8000}) {
            const int iters = 400;
            stores<16><<<grid, 256>>>(out, 10, gap, wg_stride);
            hipEventRecord(e0);
            stores<16><<<grid, 256>>>(out, iters, gap, wg_stride);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double us_it = ms * 1e3 / iters, bytes = 4.0 * 16 * 1024;   // per CU and iteration
            printf("grid %3d gap %5d VALU: %.3f us per burst of 16 x 1 KB per wave (4 waves): %.1f B/clk/CU at 2.4 GHz, %.2f TB/s over %d CUs\n", grid, gap,
                   us_it, bytes / (us_it * 2400.0), bytes * grid / us_it * 1e-6, grid);
        }
    return 0;
}
