// Issue cost of vector instructions on gfx950, one wave per SIMD: chains of 16 independent accumulators of
//   v_fma_f32 (1 flop-pair per lane), v_pk_fma_f32 (2), v_exp_f32 / v_rcp_f32 (transcendental), v_add_f32 with a DPP operand,
// and an f32 MFMA stream with and without interleaved v_fma_f32 of the SAME wave.  Cycles per instruction from the wall clock at
// the measured shader clock.  hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o valu_rate_bench && ./valu_rate_bench
// (why: DESIGN.md section 4 argues from "every vector instruction is time on the SIMD"; this prints the per-instruction prices)
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>
__global__ void chain(float* out, int iters, float a, float b) {
    float v[16];
    f32x2 p[16];
    f32x16 acc[4];
    for (int i = 0; i < 16; ++i) { v[i] = (float)(threadIdx.x + i) * 1e-3f; p[i] = f32x2{v[i], v[i] + 1.f}; }
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], a, b);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = __builtin_elementwise_fma(p[i], f32x2{a, a}, f32x2{b, b});
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_rcpf(v[i]);
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                v[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0x128, 0xf, 0xf, true));
        } else if constexpr (MODE == 5) {   // 16 MFMAs
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        } else if constexpr (MODE == 6) {   // 16 MFMAs + 16 independent v_fma_f32 between them
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    v[4 * k + i] = __builtin_fmaf(v[4 * k + i], a, b);
                }
        } else {                            // 16 MFMAs + 16 independent v_pk_fma_f32
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    p[4 * k + i] = __builtin_elementwise_fma(p[4 * k + i], f32x2{a, a}, f32x2{b, b});
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i] + p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void clock_probe(long long* out) {   // shader clock: s_memtime counts at 100 MHz, s_memrealtime... use wall_clock64 vs clock64
    const long long c0 = clock64(), w0 = wall_clock64();
    while (wall_clock64() - w0 < 100000) {}   // 1 ms at 100 MHz
    out[0] = clock64() - c0; out[1] = wall_clock64() - w0;
}

template <int MODE>
static void run(const char* name, float* out, double ghz, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, threads = 256 * waves_per_simd, grid = 256;
    chain<MODE><<<grid, threads>>>(out, 10, 0.999f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<MODE><<<grid, threads>>>(out, iters, 0.999f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)iters * 16 * waves_per_simd);
    printf("%-46s %d wave(s)/SIMD: %7.3f ms  %6.2f SIMD cycles per group-of-one\n", name, waves_per_simd, ms, cyc);
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    long long *clk, h[2];
    hipMalloc(&clk, 16);
    clock_probe<<<1, 64>>>(clk);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    printf("shader clock %.3f GHz\n", ghz);
    for (int w : {1, 2}) {
        run<0>("v_fma_f32", out, ghz, w);
        run<1>("v_pk_fma_f32", out, ghz, w);
        run<2>("v_exp_f32", out, ghz, w);
        run<3>("v_rcp_f32", out, ghz, w);
        run<4>("v_add_f32 dpp row_ror:8", out, ghz, w);
        run<5>("v_mfma_f32_32x32x2_f32", out, ghz, w);
        run<6>("v_mfma_f32_32x32x2_f32 + v_fma_f32 (same wave)", out, ghz, w);
        run<7>("v_mfma_f32_32x32x2_f32 + v_pk_fma_f32", out, ghz, w);
    }
    return 0;
}
