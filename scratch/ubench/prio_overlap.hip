// Probe: two waves of one SIMD that alternate between a burst of bf16 MFMAs and a burst of VALU instructions
// (how a layer of the critic kernels looks).  Does a raised priority during the VALU burst, with the partner
// half a period out of phase, let the VALU burst issue beside the partner's MFMAs?
// PRIO 0: no priorities; 1: priority up for the VALU burst; 2: priority up for the MFMA burst.
// STAGGER: waves 4..7 start with the VALU burst.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)
template <int NM, int NV, int PRIO, bool STAGGER, int INTER>
__global__ __launch_bounds__(512) void k(float *out, int iters, float a0) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + e + threadIdx.x); b[e] = (__bf16)(a0 - e); }
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = a0 * i + threadIdx.x;
    const float bv = a0 + 1.f;
    const bool late = STAGGER && (threadIdx.x >> 6) >= (blockDim.x >> 7);
    auto valu = [&] {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(2);
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i % 12]) : "v"(bv));
    };
    auto mfma = [&] {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (PRIO == 2) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
    };
    if (late) { valu(); __builtin_amdgcn_sched_barrier(0); }
    for (int it = 0; it < iters; ++it) {
        if (INTER) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < INTER; ++n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i * INTER + n) % 12]) : "v"(bv));
            }
        } else {
            mfma();
            __builtin_amdgcn_sched_barrier(0);
            valu();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int NV, int PRIO, bool STAGGER, int INTER>
int run(const char *what, int threads) {
    const int iters = 10000;
    float *out;
    CHECK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NM, NV, PRIO, STAGGER, INTER>), dim3(256), dim3(threads), 0, 0, out, iters, 1.f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const int wps = threads / 256;
    printf("%-52s %d wave(s)/SIMD: %7.1f ns per (%d MFMA + %d VALU) and SIMD\n", what, wps, ms * 1e6 / iters / wps, NM, INTER ? NM * INTER : NV);
    CHECK(hipFree(out));
    return 0;
}
int main() {
    run<12, 36, 0, false, 0>("warm-up", 512);
    run<12, 0, 0, false, 0>("MFMA only", 256);
    run<0, 36, 0, false, 0>("VALU only", 256);
    run<0, 36, 0, false, 0>("VALU only", 512);
    run<12, 36, 0, false, 0>("bursts", 256);
    run<12, 36, 0, false, 0>("bursts", 512);
    run<12, 36, 0, true, 0>("bursts, staggered", 512);
    run<12, 36, 1, false, 0>("bursts, prio up in VALU", 512);
    run<12, 36, 1, true, 0>("bursts, prio up in VALU, staggered", 512);
    run<12, 36, 2, false, 0>("bursts, prio up in MFMA", 512);
    run<12, 36, 2, true, 0>("bursts, prio up in MFMA, staggered", 512);
    run<12, 36, 0, false, 1>("interleaved 1:1", 256);
    run<12, 36, 0, false, 2>("interleaved 1:2", 256);
    run<12, 36, 0, false, 3>("interleaved 1:3", 256);
    run<12, 36, 0, false, 4>("interleaved 1:4", 256);
    run<12, 36, 0, false, 1>("interleaved 1:1", 512);
    run<12, 36, 0, false, 2>("interleaved 1:2", 512);
    run<12, 36, 0, false, 3>("interleaved 1:3", 512);
    run<12, 36, 0, false, 4>("interleaved 1:4", 512);
    run<12, 24, 0, false, 0>("bursts 12:24", 512);
    run<12, 24, 1, true, 0>("bursts 12:24, prio up in VALU, staggered", 512);
    run<12, 72, 0, false, 0>("bursts 12:72", 512);
    run<12, 72, 1, true, 0>("bursts 12:72, prio up in VALU, staggered", 512);
    return 0;
}
