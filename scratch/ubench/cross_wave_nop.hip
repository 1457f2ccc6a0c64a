// Probe: a wave of back-to-back bf16 MFMAs beside a wave of VALU work on the same SIMD takes the sum of their times
// (cross_wave_overlap.hip).  Does the VALU wave get its issue slots when the MFMA wave leaves a gap after every MFMA
// (s_nop N: the wave has nothing at the issue port for N + 1 cycles) or when it runs at a lower priority?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)
template <int NOP, int PRIO, bool MFMA_YOUNG>
__global__ __launch_bounds__(512) void k(float *out, int mfma_iters, int valu_iters, float a0) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    const bool mfma_wave = MFMA_YOUNG ? wave >= 4 : wave < 4;
    if (mfma_wave) {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + e + threadIdx.x); b[e] = (__bf16)(a0 - e); }
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP < 0 ? 0 : NOP));
            }
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a0 * i + threadIdx.x;
        const float bv = a0 + 1.f;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int n = 0; n < 32; ++n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n % 8]) : "v"(bv));
        }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NOP, int PRIO, bool MFMA_YOUNG>
int run(const char *what, int mi, int vi) {
    float *out;
    CHECK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NOP, PRIO, MFMA_YOUNG>), dim3(256), dim3(512), 0, 0, out, mi, vi, 1.f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-64s %8.3f ms\n", what, ms);
    CHECK(hipFree(out));
    return 0;
}
int main() {
    const int N = 20000;
    run<-1, 0, false>("warm-up", N, N);
    run<-1, 0, false>("MFMA wave alone", N, 0);
    run<-1, 0, false>("VALU wave alone", 0, N);
    run<-1, 0, false>("MFMA (older) + VALU", N, N);
    run<-1, 0, true>("MFMA (younger) + VALU", N, N);
    run<-1, 1, false>("MFMA (older, prio 0) + VALU (prio 3)", N, N);
    run<-1, 1, true>("MFMA (younger, prio 0) + VALU (prio 3)", N, N);
    run<0, 0, false>("MFMA + s_nop 0 alone", N, 0);
    run<0, 0, false>("MFMA + s_nop 0 (older) + VALU", N, N);
    run<1, 0, false>("MFMA + s_nop 1 alone", N, 0);
    run<1, 0, false>("MFMA + s_nop 1 (older) + VALU", N, N);
    run<3, 0, false>("MFMA + s_nop 3 alone", N, 0);
    run<3, 0, false>("MFMA + s_nop 3 (older) + VALU", N, N);
    run<3, 0, true>("MFMA + s_nop 3 (younger) + VALU", N, N);
    run<7, 0, false>("MFMA + s_nop 7 alone", N, 0);
    run<7, 0, false>("MFMA + s_nop 7 (older) + VALU", N, N);
    run<7, 0, true>("MFMA + s_nop 7 (younger) + VALU", N, N);
    run<11, 0, false>("MFMA + s_nop 11 alone", N, 0);
    run<11, 0, false>("MFMA + s_nop 11 (older) + VALU", N, N);
    return 0;
}
