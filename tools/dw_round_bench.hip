// One dW "round" of csrc/ppo_fused.hip in isolation (8 waves, 2 per SIMD, one s_barrier per round, 32 MFMAs per wave and
// round), with its non-MFMA parts switched on one at a time -- what does each cost on top of the 4 224-cycle MFMA block?
//   bit 0  operand reads from the sample-major LDS tile (else register operands)
//   bit 1  ... issued in quarters, one quarter ahead of the MFMAs (else all 48 reads, one wait, 32 MFMAs)
//   bit 2  publish: every wave 2 ds_write_b128 (B slice), the round's owner 16 ds_write_b128 (A tile)
//   bit 3  staging: one global float4 load per thread and round, written to LDS at the end of the round
//   bit 4  first-layer recompute of the B slice (8 elements, 3 inputs, weights in registers, relu) behind the first quarter
//   bit 5  partner waves (4-7) do their recompute behind the third quarter instead
// hipcc --offload-arch=gfx950 -O3 tools/dw_round_bench.hip -o /tmp/dw_round_bench && /tmp/dw_round_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int T = 4, kWaves = 8;
__host__ __device__ constexpr int tpitch(int nt) { return nt * 32 + 4; }
__host__ __device__ constexpr int tsize(int nt) { return 32 * tpitch(nt); }
constexpr int BUF = 2 * tsize(T);
// MODE bit 6: accumulators in AccVGPRs (inline asm), else wherever hipcc puts them (ArchVGPRs at this register budget)
template <bool ACC>
__device__ __forceinline__ void mfma(f32x16& c, float a, float b) {
    if constexpr (ACC) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int MODE>
__global__ __launch_bounds__(512) void rounds(float* out, const float* src, int nrounds, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* X = lds;                 // 2 buffers of (A tile | B tile)
    float* wbuf = lds + 2 * BUF;    // staging destination (64 KB)
    float* xs = wbuf + T * T * 1024;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, h = lane >> 5, li = lane & 31;
    for (int i = tid; i < 2 * BUF + T * T * 1024 + kWaves * 32 * 3; i += 512) lds[i] = 1e-3f * (float)((i * 37) % 101);
    const int ao = (wave * 2) / T, bi0 = (wave * 2) % T;
    f32x16 acc[2], az[T];
    for (int n = 0; n < 2; ++n)
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int t = 0; t < T; ++t)
        for (int r = 0; r < 16; ++r) az[t][r] = (float)(lane + r + t);
    float pv[8], w[8][4];
    for (int i = 0; i < 8; ++i) {
        pv[i] = (float)i;
        for (int j = 0; j < 4; ++j) w[i][j] = 0.01f * (float)(i + j + lane);
    }
    const int pt = wave / 2, pr0 = (wave * 8) % 16;
    const float4* src4 = reinterpret_cast<const float4*>(src);
    float4* dst4 = reinterpret_cast<float4*>(wbuf);
    float bsum = 0.f;
    float dav[4] = {0.f, 0.f, 0.f, 0.f}, dbv[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (clk && blockIdx.x == 0 && tid == 0) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
#pragma unroll 1
    for (int t = 0; t < nrounds; ++t) {
        if constexpr (!(MODE & 128)) LDS_BARRIER();   // bit 7: no barrier (timing only)
        const float* cur = X + (t & 1) * BUF;
        float* nxt = X + ((t + 1) & 1) * BUF;
        const float* Ap = cur + h * tpitch(T) + ao * 32 + li;
        const float* Bp = cur + tsize(T) + h * tpitch(T) + bi0 * 32 + li;
        auto fetch = [&](int ow) {
            const float* x = xs + ((ow & 7) * 32 + li) * 3;
            const float x0 = x[0], x1 = x[1], x2 = x[2];
#pragma unroll
            for (int i = 0; i < 8; ++i) pv[i] = fmaxf(fmaf(x2, w[i][2], fmaf(x1, w[i][1], fmaf(x0, w[i][0], w[i][3]))), 0.f);
        };
        auto publish = [&]() {
            if constexpr (MODE & 8192) {   // bit 13: FEATURE-major images [feature][32 samples], pitch 36: ds_write_b32 publishes, b128 reads
                float* pb = nxt + 128 * 36 + (pt * 32 + 4 * h + 8 * (pr0 >> 2)) * 36 + li;
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[((e & 3) + 8 * (e >> 2)) * 36] = pv[e];
                if (wave == ((t + 1) & 7)) {
                    float* q = nxt + (4 * h) * 36 + li;
#pragma unroll
                    for (int tt = 0; tt < T; ++tt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) q[(tt * 32 + (r & 3) + 8 * (r >> 2)) * 36] = az[tt][r];
                }
                return;
            }
            float* p = nxt + tsize(T) + li * tpitch(T) + pt * 32 + 4 * h + 8 * (pr0 >> 2);
            if constexpr (!(MODE & 2048)) {   // bit 11: no B publish
                *reinterpret_cast<float4*>(p) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                *reinterpret_cast<float4*>(p + 8) = make_float4(pv[4], pv[5], pv[6], pv[7]);
            }
            if ((MODE & 4096) ? false : wave == ((t + 1) & 7)) {   // bit 12: no A publish
                float* q = nxt + li * tpitch(T) + 4 * h;
#pragma unroll
                for (int tt = 0; tt < T; ++tt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(q + tt * 32 + 8 * g) = make_float4(az[tt][4 * g], az[tt][4 * g + 1], az[tt][4 * g + 2], az[tt][4 * g + 3]);
            }
        };
        float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr ((MODE & 1) && (MODE & 2) && (MODE & 1024)) {
            // bit 10: the last quarter's MFMAs of a round are issued BEHIND the next barrier (operands already in registers),
            // in front of which the new round's first reads go out: the matrix pipe has work while they are in flight
            float av[2][4], bv[2][2][4];
            auto rd = [&](int q, float (&a4)[4], float (&b4)[2][4]) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    a4[s] = Ap[2 * (q * 4 + s) * tpitch(T)];
                    b4[0][s] = Bp[2 * (q * 4 + s) * tpitch(T)];
                    b4[1][s] = Bp[2 * (q * 4 + s) * tpitch(T) + 32];
                }
            };
            auto mm = [&](float (&a4)[4], float (&b4)[2][4]) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    mfma<false>(acc[0], a4[s], b4[0][s]);
                    mfma<false>(acc[1], a4[s], b4[1][s]);
                    bsum += a4[s];
                }
            };
            rd(0, av[0], bv[0]);
            rd(1, av[1], bv[1]);
            __builtin_amdgcn_sched_barrier(0);
            mm(dav, dbv);   // round t-1, quarter 3 (zeros in round 0)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE & 4) publish();
            if constexpr (MODE & 8) { asm volatile("" ::: "memory"); sv = src4[(t & 7) * 512 + tid]; }
            __builtin_amdgcn_sched_barrier(0);
            mm(av[0], bv[0]);
            __builtin_amdgcn_sched_barrier(0);
            rd(2, av[0], bv[0]);
            if constexpr (MODE & 16) { if (!((MODE & 32) && wave >= 4)) fetch(t + 2); }
            __builtin_amdgcn_sched_barrier(0);
            mm(av[1], bv[1]);
            __builtin_amdgcn_sched_barrier(0);
            rd(3, dav, dbv);
            __builtin_amdgcn_sched_barrier(0);
            mm(av[0], bv[0]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE & 16) { if ((MODE & 32) && wave >= 4) fetch(t + 2); }
        } else if constexpr ((MODE & 1) && (MODE & 2)) {
            float av[2][4], bv[2][2][4];
            auto rd = [&](int q, int b) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    av[b][s] = Ap[2 * (q * 4 + s) * tpitch(T)];
                    bv[b][0][s] = Bp[2 * (q * 4 + s) * tpitch(T)];
                    bv[b][1][s] = Bp[2 * (q * 4 + s) * tpitch(T) + 32];
                }
            };
            rd(0, 0);
            rd(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE & 4) publish();
            if constexpr (MODE & 8) { asm volatile("" ::: "memory"); sv = src4[(t & 7) * 512 + tid]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q >= 1 && q + 1 < 4) rd(q + 1, (q + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    mfma<(MODE & 64) != 0>(acc[0], av[q & 1][s], bv[q & 1][0][s]);
                    mfma<(MODE & 64) != 0>(acc[1], av[q & 1][s], bv[q & 1][1][s]);
                    bsum += av[q & 1][s];
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE & 16) {
                    const bool late = (MODE & 32) && wave >= 4;
                    if ((q == 0 && !late) || (q == 2 && late)) fetch(t + 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if constexpr (MODE & 4) publish();
            if constexpr (MODE & 8) { asm volatile("" ::: "memory"); sv = src4[(t & 7) * 512 + tid]; }
            if constexpr ((MODE & 16) && !(MODE & 32)) fetch(t + 2);
            if constexpr ((MODE & 16) && (MODE & 32)) { if (wave < 4) fetch(t + 2); }
            float av[16], bv[2][16];
            if constexpr ((MODE & 1) && (MODE & 512)) {   // bit 9: 12 ds_read_b128 instead of 48 ds_read_b32 (sample-contiguous operand layout)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float* Af = cur + (ao * 32 + li) * 36 + 16 * h, *Bf = cur + 128 * 36 + (bi0 * 32 + li) * 36 + 16 * h;
                    const float4 va = *reinterpret_cast<const float4*>(Af + 4 * s);
                    const float4 v0 = *reinterpret_cast<const float4*>(Bf + 4 * s);
                    const float4 v1 = *reinterpret_cast<const float4*>(Bf + 4 * s + 32 * 36);
                    av[4 * s] = va.x; av[4 * s + 1] = va.y; av[4 * s + 2] = va.z; av[4 * s + 3] = va.w;
                    bv[0][4 * s] = v0.x; bv[0][4 * s + 1] = v0.y; bv[0][4 * s + 2] = v0.z; bv[0][4 * s + 3] = v0.w;
                    bv[1][4 * s] = v1.x; bv[1][4 * s + 1] = v1.y; bv[1][4 * s + 2] = v1.z; bv[1][4 * s + 3] = v1.w;
                }
            } else if constexpr (MODE & 1) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    av[s] = Ap[2 * s * tpitch(T)];
                    bv[0][s] = Bp[2 * s * tpitch(T)];
                    bv[1][s] = Bp[2 * s * tpitch(T) + 32];
                }
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) { av[s] = pv[s & 7]; bv[0][s] = w[s & 7][0]; bv[1][s] = w[s & 7][1]; }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                mfma<(MODE & 64) != 0>(acc[0], av[s], bv[0][s]);
                mfma<(MODE & 64) != 0>(acc[1], av[s], bv[1][s]);
                bsum += av[s];
                if constexpr (MODE & 256) {   // bit 8: 16 independent VALU fmas per 2 MFMAs (64 of 128 pipe cycles), same wave
#pragma unroll
                    for (int i = 0; i < 8; ++i) { pv[i] = fmaf(pv[i], 1.0001f, w[i][0]); w[i][1] = fmaf(w[i][1], 0.9999f, w[i][2]); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("" ::: "memory");
            if constexpr ((MODE & 16) && (MODE & 32)) { if (wave >= 4) fetch(t + 2); }
        }
        if constexpr (MODE & 8) dst4[(t & 7) * 512 + tid] = sv;
    }
    if (clk && blockIdx.x == 0 && tid == 0) {
        unsigned long long c1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
        clk[0] = c1 - c0; clk[1] = r1 - r0;
    }
    float s = bsum + dav[0] + dbv[1][3];
    for (int i = 0; i < 8; ++i) s += pv[i] + w[i][1];
    for (int n = 0; n < 2; ++n)
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
static void run(float* out, const float* src, int nrounds, int grid, unsigned long long* clk) {
    const size_t lds_bytes = sizeof(float) * (2 * BUF + T * T * 1024 + kWaves * 32 * 3 + 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(rounds<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rounds<MODE>, dim3(grid), dim3(512), lds_bytes, 0, out, src, 64, (unsigned long long*)nullptr);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rounds<MODE>, dim3(grid), dim3(512), lds_bytes, 0, out, src, nrounds, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double us = best * 1e3 / nrounds;
    unsigned long long hc[2];
    hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
    const double ghz = (double)hc[0] / (double)hc[1] * 0.1;   // s_memtime ticks per 100 MHz s_memrealtime tick
    printf("grid %3d mode %4d%s%s%s%s%s%s%s: %.3f us per round = %.0f shader cycles at %.2f GHz (MFMA block: 4096); %.1f TFLOP/s if 256 CUs\n", grid, MODE,
           MODE & 8192 ? " FEATURE-MAJOR" : "", MODE & 1 ? " reads" : "", MODE & 2 ? " pipelined" : "", MODE & 4 ? " publish" : "", MODE & 8 ? " staging" : "",
           MODE & 16 ? " recompute" : "", MODE & 1024 ? " early/late+deferred" : (MODE & 32 ? " early/late" : ""), us, us * ghz * 1e3, ghz,
           256.0 * 8 * 32 * 4096 / us * 1e-6);
}

int main(int argc, char** argv) {
    const int nrounds = argc > 1 ? atoi(argv[1]) : 4000;
    float *out, *src;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&src, 8 * 512 * 16);
    hipMemset(src, 0, 8 * 512 * 16);
    unsigned long long* clk;
    hipMalloc(&clk, 16);
    for (int grid : {256}) {
        run<0>(out, src, nrounds, grid, clk);
        run<5>(out, src, nrounds, grid, clk);
        run<8709>(out, src, nrounds, grid, clk);     // 1 + 4 + 512 + 8192
        run<13>(out, src, nrounds, grid, clk);
        run<8717>(out, src, nrounds, grid, clk);
        run<29>(out, src, nrounds, grid, clk);
        run<8733>(out, src, nrounds, grid, clk);
        run<61>(out, src, nrounds, grid, clk);
        run<8765>(out, src, nrounds, grid, clk);
    }
    hipDeviceSynchronize();
    return 0;
}
