// Probe: the same work -- 12 bf16 MFMAs and 36 VALU instructions per step -- issued as a burst of MFMAs followed by a
// burst of VALU work (how a layer of the critic kernels looks) or interleaved one MFMA : three VALU, at one and two
// waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)
#define VALU3(i) \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(3 * (i)) % 12]) : "v"(bv)); \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(3 * (i) + 1) % 12]) : "v"(bv)); \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(3 * (i) + 2) % 12]) : "v"(bv));
template <bool INTERLEAVE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + e + threadIdx.x); b[e] = (__bf16)(a0 - e); }
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = a0 * i + threadIdx.x;
    const float bv = a0 + 1.f;
    for (int it = 0; it < iters; ++it) {
        if (INTERLEAVE) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
                VALU3(i)
            }
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 12; ++i) { VALU3(i) }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <bool INTERLEAVE>
int run(const char *what, int waves) {
    const int iters = 10000;
    float *out;
    CHECK(hipMalloc(&out, 256 * 4 * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<INTERLEAVE>), dim3(256 * waves), dim3(256), 0, 0, out, iters, 1.f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-34s %d wave(s) per SIMD: %7.1f ns per step (12 MFMA + 36 VALU) and SIMD\n", what, waves, ms * 1e6 / iters / waves);
    CHECK(hipFree(out));
    return 0;
}
int main() {
    run<false>("warm-up", 2);
    run<false>("bursts", 1); run<true>("interleaved 1 : 3", 1);
    run<false>("bursts", 2); run<true>("interleaved 1 : 3", 2);
    return 0;
}
