// A/B of ONE streamed chain layer of the 16-tile family (csrc/mlp16.hip) on the product's launch shape (512 workgroups of four waves,
// two per compute unit, weight slices by LDS-DMA): layer16r (v_mfma_f32_16x16x4_f32) against layer16r_b3 (three bf16 pieces per f32
// operand, six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block).  Both against a float64 reference: max and rms error relative to
// max |y|; then the layer in a loop (x <- 0.05 y between layers), f32-equivalent TFLOP/s.
//   bash tools/build_b3_benches.sh && ./tools/bin/layer16_b3_bench
// (the tool includes mlp16.hip itself -- the layer routines are templates of that translation unit -- and links the other objects)
#include "mlp16.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace pime;

template <int T, bool B3>
__global__ __launch_bounds__(k16Threads, 2) void bench_kernel(const float* __restrict__ img, const float* __restrict__ x,
                                                               float* __restrict__ y, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t tile = (size_t)blockIdx.x * k16Waves + wave;
    f32x4 in[T], out[T];
    const f32x4* xp = reinterpret_cast<const f32x4*>(x) + (tile * 64 + lane) * T;
#pragma unroll
    for (int t = 0; t < T; ++t) in[t] = xp[t];
    for (int rep = 0; rep < reps; ++rep) {
        if constexpr (B3) layer16r_b3<T, T, 2, false>(img, nullptr, lds, lane, tid, in, out);
        else layer16r<T, T, 2, false>(img, nullptr, lds, lane, tid, in, out);
        if (rep + 1 < reps)
#pragma unroll
            for (int t = 0; t < T; ++t) in[t] = out[t] * 0.05f;
    }
    f32x4* yp = reinterpret_cast<f32x4*>(y) + (tile * 64 + lane) * T;
#pragma unroll
    for (int t = 0; t < T; ++t) yp[t] = out[t];
}

// The same layer with ONE workgroup per compute unit (the width-256 gradient kernels' situation: a wave alone on its SIMD), for
// slice shapes (OTX output tiles per slice) and buffer counts NB of layer16r_b3; OTX < 0: the f32 layer.
template <int T, int OTX, int NB>
__global__ __launch_bounds__(k16Threads, 1) void bench1_kernel(const float* __restrict__ img, const float* __restrict__ x,
                                                                float* __restrict__ y, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t tile = (size_t)blockIdx.x * k16Waves + wave;
    f32x4 in[T], out[T];
    const f32x4* xp = reinterpret_cast<const f32x4*>(x) + (tile * 64 + lane) * T;
#pragma unroll
    for (int t = 0; t < T; ++t) in[t] = xp[t];
    for (int rep = 0; rep < reps; ++rep) {
        if constexpr (OTX >= 0) layer16r_b3<T, T, 2, false, OTX, NB>(img, nullptr, lds, lane, tid, in, out);
        else layer16r<T, T, 2, false>(img, nullptr, lds, lane, tid, in, out);
        if (rep + 1 < reps)
#pragma unroll
            for (int t = 0; t < T; ++t) in[t] = out[t] * 0.05f;
    }
    f32x4* yp = reinterpret_cast<f32x4*>(y) + (tile * 64 + lane) * T;
#pragma unroll
    for (int t = 0; t < T; ++t) yp[t] = out[t];
}

template <int T, int OTX, int NB>
static void run1(const char* name, const float* img, const float* dx, float* dy, int N) {
    const int grid = 256, reps = 400, MD = T * 16;
    const size_t lds = 120 * 1024;   // the gradient kernels' region at width 256; > 80 KB: one workgroup per compute unit
    hipFuncSetAttribute((const void*)bench1_kernel<T, OTX, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((bench1_kernel<T, OTX, NB>), dim3(grid), dim3(k16Threads), lds, 0, img, dx, dy, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    (void)N;
    printf("width %3d one workgroup per CU  %-22s %6.2f us per layer (64 samples): %.1f TFLOP/s f32-equivalent\n", MD, name,
           best * 1e3 / reps, 2.0 * MD * MD * 64.0 * grid * reps / (best * 1e-3) / 1e12);
}

template <int T>
static int run() {
    constexpr int MD = T * 16;
    const int grid = 512, N = grid * k16Waves * 16;   // samples
    std::vector<float> W((size_t)MD * MD), X((size_t)MD * N);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& w : W) w = rnd() * 0.15f;   // ~ the scale of a trained layer
    for (auto& v : X) v = rnd();           // activations in (-1, 1)
    // f32 chain image (pack16_layer, natural = false)
    const int Q = T / 4, KS = MD / 4;
    std::vector<float> img((size_t)MD * MD);
    for (int idx = 0; idx < MD * MD; ++idx) {
        const int e = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) % Q, ks = idx / (Q * 256);
        const int i = lane & 15, g = lane >> 4;
        img[idx] = W[(size_t)(16 * (4 * q + e) + i) * MD + 16 * (ks >> 2) + 4 * g + (ks & 3)];
    }
    (void)KS;
    // bf16x3 planes (pack16_b3_layer)
    using G = Layer16B3Geom<T, T>;
    std::vector<unsigned short> pl((size_t)G::IMAGE * 2);
    for (int idx = 0; idx < (T / 2) * T * 64 * 8; ++idx) {
        const int e = idx & 7, lane = (idx >> 3) & 63, tt = (idx >> 9) % G::OT, sl = idx / (512 * G::OT);
        const int ks = sl / G::SPK, to = (sl % G::SPK) * G::OT + tt, i = lane & 15, g = lane >> 4;
        float p3[3];
        split_bf16x3(W[(size_t)(16 * to + i) * MD + 16 * (2 * ks + (e >> 2)) + 4 * g + (e & 3)], p3[0], p3[1], p3[2]);
        for (int p = 0; p < 3; ++p) {
            const unsigned u = __builtin_bit_cast(unsigned, p3[p]);
            pl[((size_t)((sl * G::OT + tt) * 3 + p) * 64 + lane) * 8 + e] = (unsigned short)(u >> 16);
        }
    }
    // activations in accumulator layout: tile, lane (sample i, group g), tile t, register r = feature 16 t + 4 g + r
    std::vector<float> Xd((size_t)MD * N);
    for (int tile = 0; tile < N / 16; ++tile)
        for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < T; ++t)
                for (int r = 0; r < 4; ++r)
                    Xd[(((size_t)tile * 64 + lane) * T + t) * 4 + r] = X[(size_t)(16 * t + 4 * (lane >> 4) + r) * N + tile * 16 + (lane & 15)];
    float *dimg, *dx, *dy;
    void* dpl;
    hipMalloc(&dimg, img.size() * 4); hipMalloc(&dx, Xd.size() * 4); hipMalloc(&dy, Xd.size() * 4); hipMalloc(&dpl, pl.size() * 2);
    hipMemcpy(dimg, img.data(), img.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, Xd.data(), Xd.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpl, pl.data(), pl.size() * 2, hipMemcpyHostToDevice);
    const size_t lds_f32 = layer16_lds_floats<T>() * 4, lds_b3 = layer16_b3_lds_floats<T>() * 4;
    hipFuncSetAttribute((const void*)bench_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f32);
    hipFuncSetAttribute((const void*)bench_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b3);
    const int NSAMP = 2048;
    std::vector<double> ref((size_t)MD * NSAMP);
    double ymax = 0;
    for (int f = 0; f < MD; ++f)
        for (int c = 0; c < NSAMP; ++c) {
            double a = 0;
            for (int k = 0; k < MD; ++k) a += (double)W[(size_t)f * MD + k] * (double)X[(size_t)k * N + c];
            ref[(size_t)f * NSAMP + c] = a;
            ymax = fmax(ymax, fabs(a));
        }
    std::vector<float> Y(Xd.size());
    auto err = [&](const char* name) {
        hipMemcpy(Y.data(), dy, Y.size() * 4, hipMemcpyDeviceToHost);
        double mx = 0, ss = 0, sb = 0;   // sb: the SIGNED error along the result's sign (a rounding that truncates shows up here)
        for (int f = 0; f < MD; ++f)
            for (int c = 0; c < NSAMP; ++c) {
                const int t = f >> 4, g = (f & 15) >> 2, r = f & 3, tile = c >> 4, lane = (c & 15) + 16 * g;
                const double rv = ref[(size_t)f * NSAMP + c];
                const double e = (double)Y[(((size_t)tile * 64 + lane) * T + t) * 4 + r] - rv, d = fabs(e);
                mx = fmax(mx, d);
                ss += d * d;
                sb += rv >= 0 ? e : -e;
            }
        printf("width %3d %-7s max |err| / max |y| = %.3e   rms err / max |y| = %.3e   mean signed err (towards larger |y|) / max |y| = %+.3e\n", MD,
               name, mx / ymax, sqrt(ss / ((double)MD * NSAMP)) / ymax, sb / ((double)MD * NSAMP) / ymax);
        return mx / ymax;
    };
    int bad = 0;
    hipLaunchKernelGGL((bench_kernel<T, false>), dim3(grid), dim3(k16Threads), lds_f32, 0, dimg, dx, dy, 1);
    hipDeviceSynchronize();
    if (err("f32") > 1e-5) bad = 1;
    hipLaunchKernelGGL((bench_kernel<T, true>), dim3(grid), dim3(k16Threads), lds_b3, 0, (const float*)dpl, dx, dy, 1);
    hipDeviceSynchronize();
    if (err("bf16x3") > 1e-5) bad = 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 400;
    double tf[2];
    for (int which = 0; which < 2; ++which) {
        float best = 1e30f;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL((bench_kernel<T, false>), dim3(grid), dim3(k16Threads), lds_f32, 0, dimg, dx, dy, reps);
            else hipLaunchKernelGGL((bench_kernel<T, true>), dim3(grid), dim3(k16Threads), lds_b3, 0, (const float*)dpl, dx, dy, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        tf[which] = 2.0 * MD * MD * (double)N * reps / (best * 1e-3) / 1e12;
        printf("width %3d %-7s %8.3f ms for %d layers x %d samples: %.1f TFLOP/s (f32-equivalent), LDS %zu B per workgroup\n", MD,
               which ? "bf16x3" : "f32", best, reps, N, tf[which], which ? lds_b3 : lds_f32);
    }
    printf("width %3d bf16x3 / f32 = %.2fx\n", MD, tf[1] / tf[0]);
    run1<T, -1, 3>("f32", dimg, dx, dy, N);
    run1<T, 0, 3>("bf16x3 8 tiles x 3", (const float*)dpl, dx, dy, N);
    run1<T, 0, 4>("bf16x3 8 tiles x 4", (const float*)dpl, dx, dy, N);
    run1<T, 0, 5>("bf16x3 8 tiles x 5", (const float*)dpl, dx, dy, N);
    if constexpr (T == 16) {
        run1<T, 16, 2>("bf16x3 16 tiles x 2", (const float*)dpl, dx, dy, N);
        run1<T, 4, 5>("bf16x3 4 tiles x 5", (const float*)dpl, dx, dy, N);
    }
    hipFree(dimg); hipFree(dx); hipFree(dy); hipFree(dpl);
    return bad;
}

int main() {
    int bad = run<8>();
    bad |= run<16>();
    return bad;
}
