// What would a "three bf16 pieces per f32 operand" layer buy, and what would it cost in accuracy?  (open item 1b)
// One 128 x 128 layer Y = W X for 32-sample tiles, two ways, on the product's launch shape (256 workgroups x 8 waves, weights in LDS):
//   f32   : v_mfma_f32_32x32x2_f32, the product's image and chain (csrc/mlp_device.hpp: layer_mfma)
//   bf16x3: W and X split as hi + mid + lo in bf16 (W on the host, X on the fly from registers), six v_mfma_f32_32x32x16_bf16 per
//           (k-block, output tile): hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid -- the products below 2^-24 relative are dropped
// Both against a float64 reference: max and rms error relative to max |Y|.  Timing: the layer in a loop, x <- 0.05 y between layers.
// hipcc --offload-arch=gfx950 -O3 tools/bf16x3_layer_bench.hip -o bf16x3_layer_bench && ./bf16x3_layer_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int MD = 128, T = 4, THREADS = 512;

__host__ __device__ inline int feat32(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- f32: image [ks = kt*16 + s][lane][ot]  (pack_mfma)
__global__ __launch_bounds__(THREADS) void layer_f32(const float* __restrict__ img, const float* __restrict__ x, float* __restrict__ y, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < MD * MD / 4; i += THREADS) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(img)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * (THREADS / 64) + wave;
    f32x16 in[T], out[T];
    {   // the tile's activations in accumulator layout, 64 floats per lane, contiguous (x[(tile 64 + lane) 64 + t 16 + r])
        const float4* xp = reinterpret_cast<const float4*>(x + ((size_t)tile * 64 + lane) * 64);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = xp[t * 4 + q];
                in[t][4 * q] = v.x; in[t][4 * q + 1] = v.y; in[t][4 * q + 2] = v.z; in[t][4 * q + 3] = v.w;
            }
    }
    const float4* wl = reinterpret_cast<const float4*>(lds) + lane;
    for (int rep = 0; rep < reps; ++rep) {
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < 16; ++r) out[t][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < T; ++kt)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if ((s & 3) == 0) asm volatile("" ::: "memory");   // bound the fragment prefetch depth (as the product's layers do)
                const float4 w = wl[(kt * 16 + s) * 64];
                const float b = in[kt][s];
                out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, b, out[0], 0, 0, 0);
                out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, b, out[1], 0, 0, 0);
                out[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, b, out[2], 0, 0, 0);
                out[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, b, out[3], 0, 0, 0);
            }
        if (rep + 1 < reps)
            for (int t = 0; t < T; ++t)
                for (int r = 0; r < 16; ++r) in[t][r] = out[t][r] * 0.05f;
    }
    {
        float4* yp = reinterpret_cast<float4*>(y + ((size_t)tile * 64 + lane) * 64);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) yp[t * 4 + q] = make_float4(out[t][4 * q], out[t][4 * q + 1], out[t][4 * q + 2], out[t][4 * q + 3]);
    }
}

// ---- bf16x3: planes [p][kt][u][ot][lane] of 8 bf16: A operand of the MFMA (kt, u, ot): row = lane & 31, k = 8 (lane >> 5) + e  <->
//      W[ot*32 + row][kt*32 + feat32(8 u + e, lane >> 5)]
__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 a = (__bf16)v[e];
        const float r1 = v[e] - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        hi[e] = a; mid[e] = b; lo[e] = (__bf16)r2;
    }
}
__global__ __launch_bounds__(THREADS) void layer_bf16x3(const bf16x8* __restrict__ planes, const float* __restrict__ x, float* __restrict__ y, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PLANE = T * 2 * T * 64;   // bf16x8 words per plane
    bf16x8* wl = reinterpret_cast<bf16x8*>(lds);
    for (int i = threadIdx.x; i < 3 * PLANE; i += THREADS) wl[i] = planes[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * (THREADS / 64) + wave;
    f32x16 in[T], out[T];
    {   // the tile's activations in accumulator layout, 64 floats per lane, contiguous (x[(tile 64 + lane) 64 + t 16 + r])
        const float4* xp = reinterpret_cast<const float4*>(x + ((size_t)tile * 64 + lane) * 64);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = xp[t * 4 + q];
                in[t][4 * q] = v.x; in[t][4 * q + 1] = v.y; in[t][4 * q + 2] = v.z; in[t][4 * q + 3] = v.w;
            }
    }
    for (int rep = 0; rep < reps; ++rep) {
        for (int t = 0; t < T; ++t)
            for (int r = 0; r < 16; ++r) out[t][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < T; ++kt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                asm volatile("" ::: "memory");   // bound the fragment prefetch depth
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = in[kt][8 * u + e];
                bf16x8 bh, bm, bl;
                split3(v, bh, bm, bl);
#pragma unroll
                for (int ot = 0; ot < T; ++ot) {
                    if (ot == 2) asm volatile("" ::: "memory");
                    const int w = ((kt * 2 + u) * T + ot) * 64 + lane;
                    const bf16x8 ah = wl[w], am = wl[PLANE + w], al = wl[2 * PLANE + w];
                    f32x16 c = out[ot];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);   // the small terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                    out[ot] = c;
                }
            }
        if (rep + 1 < reps)
            for (int t = 0; t < T; ++t)
                for (int r = 0; r < 16; ++r) in[t][r] = out[t][r] * 0.05f;
    }
    {
        float4* yp = reinterpret_cast<float4*>(y + ((size_t)tile * 64 + lane) * 64);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) yp[t * 4 + q] = make_float4(out[t][4 * q], out[t][4 * q + 1], out[t][4 * q + 2], out[t][4 * q + 3]);
    }
}

static unsigned short bf16_bits(float f) {   // round to nearest even
    unsigned u; memcpy(&u, &f, 4);
    const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(r >> 16);
}
static float bf16_val(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int grid = 256, N = grid * 8 * 32;
    std::vector<float> W(MD * MD), X((size_t)MD * N), img(MD * MD);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& w : W) w = rnd() * 0.15f;      // ~ the scale of a trained 128-wide layer
    for (auto& v : X) v = rnd();              // activations in (-1, 1)
    for (int idx = 0; idx < MD * MD; ++idx) {
        const int ot = idx % T, lane = (idx / T) & 63, ks = idx / (T * 64), kt = ks >> 4, s = ks & 15;
        img[idx] = W[(ot * 32 + (lane & 31)) * MD + kt * 32 + feat32(s, lane >> 5)];
    }
    const int PLANE = T * 2 * T * 64;
    std::vector<unsigned short> planes((size_t)3 * PLANE * 8);
    for (int kt = 0; kt < T; ++kt) for (int u = 0; u < 2; ++u) for (int ot = 0; ot < T; ++ot) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
        const float w = W[(ot * 32 + (lane & 31)) * MD + kt * 32 + feat32(8 * u + e, lane >> 5)];
        const unsigned short a = bf16_bits(w); const float r1 = w - bf16_val(a);
        const unsigned short b = bf16_bits(r1); const float r2 = r1 - bf16_val(b);
        const size_t word = ((size_t)((kt * 2 + u) * T + ot) * 64 + lane) * 8 + e;
        planes[word] = a; planes[(size_t)PLANE * 8 + word] = b; planes[(size_t)2 * PLANE * 8 + word] = bf16_bits(r2);
    }
    float *dimg, *dx, *dy; void* dpl;
    hipMalloc(&dimg, img.size() * 4); hipMalloc(&dx, X.size() * 4); hipMalloc(&dy, X.size() * 4); hipMalloc(&dpl, planes.size() * 2);
    hipMemcpy(dimg, img.data(), img.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> Xd(X.size());
    for (int tile = 0; tile < N / 32; ++tile) for (int lane = 0; lane < 64; ++lane) for (int t = 0; t < T; ++t) for (int r = 0; r < 16; ++r)
        Xd[((size_t)tile * 64 + lane) * 64 + t * 16 + r] = X[(size_t)(t * 32 + feat32(r, lane >> 5)) * N + tile * 32 + (lane & 31)];
    hipMemcpy(dx, Xd.data(), Xd.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpl, planes.data(), planes.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)layer_bf16x3, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PLANE * 16);
    // reference on a sample of columns (float64)
    const int NS = 2048;
    std::vector<double> ref((size_t)MD * NS);
    double ymax = 0;
    for (int f = 0; f < MD; ++f) for (int c = 0; c < NS; ++c) {
        double a = 0;
        for (int k = 0; k < MD; ++k) a += (double)W[f * MD + k] * (double)X[(size_t)k * N + c];
        ref[(size_t)f * NS + c] = a; ymax = fmax(ymax, fabs(a));
    }
    std::vector<float> Y(X.size());
    auto err = [&](const char* name) {
        hipMemcpy(Y.data(), dy, Y.size() * 4, hipMemcpyDeviceToHost);
        double mx = 0, ss = 0;
        for (int f = 0; f < MD; ++f) for (int c = 0; c < NS; ++c) {
            const int t = f >> 5, w = f & 31, hh = (w >> 2) & 1, r = (w & 3) + 4 * (w >> 3), tile = c >> 5, lane = (c & 31) + 32 * hh;
            const double d = fabs((double)Y[((size_t)tile * 64 + lane) * 64 + t * 16 + r] - ref[(size_t)f * NS + c]);
            mx = fmax(mx, d); ss += d * d;
        }
        printf("%-8s max |err| / max |y| = %.3e   rms err / max |y| = %.3e\n", name, mx / ymax, sqrt(ss / (MD * NS)) / ymax);
    };
    layer_f32<<<grid, THREADS, MD * MD * 4>>>(dimg, dx, dy, 1); hipDeviceSynchronize(); err("f32");
    layer_bf16x3<<<grid, THREADS, 3 * PLANE * 16>>>((const bf16x8*)dpl, dx, dy, 1); hipDeviceSynchronize(); err("bf16x3");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 400;
    for (int which = 0; which < 2; ++which) {
        float ms = 0;
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
            if (which == 0) layer_f32<<<grid, THREADS, MD * MD * 4>>>(dimg, dx, dy, reps);
            else layer_bf16x3<<<grid, THREADS, 3 * PLANE * 16>>>((const bf16x8*)dpl, dx, dy, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double flop = 2.0 * MD * MD * (double)N * reps;
        printf("%-8s %.3f ms for %d layers of %d samples: %.1f TFLOP/s f32-equivalent (f32 matrix peak 157.3)\n", which ? "bf16x3" : "f32", ms, reps, N,
               flop / ms / 1e9);
    }
    return 0;
}
