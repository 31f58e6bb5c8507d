// A/B of ONE width-256 weight-gradient job of the 16-tile family (csrc/mlp16.hip) on the product's launch shape (256 workgroups of four
// waves, one per compute unit, 64 samples per job): dw16_sliced (v_mfma_f32_16x16x4_f32, f32 sample-major LDS image) against
// dw16_sliced_b3 (three bf16 planes per operand in LDS, ds_read_b64_tr_b16 transposing reads, six v_mfma_f32_16x16x32_bf16 per
// block and k-step).  Both against a float64 reference (workgroup 0): max error relative to max |dW|; then the job in a loop
// (accumulating into the workgroup's slab, as the second and later groups of a workgroup do).
//   bash tools/build_b3_benches.sh && ./tools/bin/dw16_b3_bench
#include "mlp16.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace pime;

constexpr int T = 16, MD = 256;

template <bool B3>
__global__ __launch_bounds__(k16Threads, 1) void bench_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               float* __restrict__ slab, float* __restrict__ gbias, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t tile = (size_t)blockIdx.x * k16Waves + wave;
    f32x4 va[T], vb[T];
    const f32x4* ap = reinterpret_cast<const f32x4*>(a) + (tile * 64 + lane) * T;
    const f32x4* bp = reinterpret_cast<const f32x4*>(b) + (tile * 64 + lane) * T;
#pragma unroll
    for (int t = 0; t < T; ++t) { va[t] = ap[t]; vb[t] = bp[t]; }
    float* const gW = slab + (size_t)blockIdx.x * MD * MD;
    float* const gb = gbias + (size_t)blockIdx.x * MD;
    for (int rep = 0; rep < reps; ++rep) {
        if constexpr (B3) dw16_sliced_b3<T>(lds, lane, wave, va, vb, gW, gb, rep > 0);
        else dw16_sliced<T>(lds, lane, wave, va, vb, gW, gb, rep > 0);
    }
}

int main() {
    const int grid = 256, N = grid * 64;
    std::vector<float> A((size_t)N * MD), B((size_t)N * MD);   // [sample][feature]
    srand(2);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : A) v = rnd() * 0.01f;   // dZ
    for (auto& v : B) v = rnd();           // activations
    auto to_acc = [&](const std::vector<float>& X) {   // accumulator layout: tile, lane (sample i, group g), tile t, register r
        std::vector<float> d((size_t)N * MD);
        for (int tile = 0; tile < N / 16; ++tile)
            for (int lane = 0; lane < 64; ++lane)
                for (int t = 0; t < T; ++t)
                    for (int r = 0; r < 4; ++r)
                        d[(((size_t)tile * 64 + lane) * T + t) * 4 + r] = X[(size_t)(tile * 16 + (lane & 15)) * MD + 16 * t + 4 * (lane >> 4) + r];
        return d;
    };
    const std::vector<float> Ad = to_acc(A), Bd = to_acc(B);
    float *da, *db, *dslab, *dgb;
    hipMalloc(&da, Ad.size() * 4); hipMalloc(&db, Bd.size() * 4);
    hipMalloc(&dslab, (size_t)grid * MD * MD * 4); hipMalloc(&dgb, (size_t)grid * MD * 4);
    hipMemcpy(da, Ad.data(), Ad.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, Bd.data(), Bd.size() * 4, hipMemcpyHostToDevice);
    const size_t lds_f32 = (size_t)Dw16Sliced<T>::FLOATS * 4, lds_b3 = (size_t)Dw16SlicedB3<T>::FLOATS * 4;
    hipFuncSetAttribute((const void*)bench_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f32);
    hipFuncSetAttribute((const void*)bench_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b3);
    // float64 reference of workgroup 0 (samples 0..63)
    std::vector<double> ref((size_t)MD * MD), refb(MD, 0.0);
    double wmax = 0, bmax = 0;
    for (int x = 0; x < MD; ++x) {
        for (int y = 0; y < MD; ++y) {
            double acc = 0;
            for (int s = 0; s < 64; ++s) acc += (double)A[(size_t)s * MD + x] * (double)B[(size_t)s * MD + y];
            ref[(size_t)x * MD + y] = acc;
            wmax = fmax(wmax, fabs(acc));
        }
        for (int s = 0; s < 64; ++s) refb[x] += (double)A[(size_t)s * MD + x];
        bmax = fmax(bmax, fabs(refb[x]));
    }
    std::vector<float> W((size_t)MD * MD), G(MD);
    int bad = 0;
    auto err = [&](const char* name) {
        hipMemcpy(W.data(), dslab, W.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(G.data(), dgb, G.size() * 4, hipMemcpyDeviceToHost);
        double mx = 0, mb = 0;
        for (int blk = 0; blk < T * T; ++blk)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {   // slab_layout16: block (a, b), lane, register -> dW[16a + 4(lane >> 4) + r][16b + (lane & 15)]
                    const int x = 16 * (blk / T) + 4 * (lane >> 4) + r, y = 16 * (blk % T) + (lane & 15);
                    mx = fmax(mx, fabs((double)W[((size_t)blk * 64 + lane) * 4 + r] - ref[(size_t)x * MD + y]));
                }
        for (int x = 0; x < MD; ++x) mb = fmax(mb, fabs((double)G[x] - refb[x]));
        printf("%-7s dW: max |err| / max |dW| = %.3e    bias: max |err| / max |db| = %.3e\n", name, mx / wmax, mb / bmax);
        if (mx / wmax > 1e-5 || mb / bmax > 1e-5) bad = 1;
    };
    hipLaunchKernelGGL((bench_kernel<false>), dim3(grid), dim3(k16Threads), lds_f32, 0, da, db, dslab, dgb, 1);
    hipDeviceSynchronize();
    err("f32");
    hipMemset(dslab, 0, (size_t)grid * MD * MD * 4);
    hipMemset(dgb, 0, (size_t)grid * MD * 4);
    hipLaunchKernelGGL((bench_kernel<true>), dim3(grid), dim3(k16Threads), lds_b3, 0, da, db, dslab, dgb, 1);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("bf16x3 launch failed: %s\n", hipGetErrorString(e)); return 2; }
    err("bf16x3");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    double us[2];
    for (int which = 0; which < 2; ++which) {
        float best = 1e30f;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL((bench_kernel<false>), dim3(grid), dim3(k16Threads), lds_f32, 0, da, db, dslab, dgb, reps);
            else hipLaunchKernelGGL((bench_kernel<true>), dim3(grid), dim3(k16Threads), lds_b3, 0, da, db, dslab, dgb, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        us[which] = best * 1e3 / reps;
        printf("%-7s %.2f us per job (64 samples x 256 x 256 per workgroup, %d workgroups): %.1f TFLOP/s f32-equivalent, LDS %zu B\n",
               which ? "bf16x3" : "f32", us[which], grid, 2.0 * 64 * MD * MD * grid / (us[which] * 1e-6) / 1e12, which ? lds_b3 : lds_f32);
    }
    printf("bf16x3 / f32 = %.2fx\n", us[0] / us[1]);
    return bad;
}
