// Probe: do the VALU instructions of one wave issue beside the bf16 MFMAs of ANOTHER wave of the same SIMD?
// Workgroups of 8 waves (two per SIMD: waves w and w + 4 share one); waves 0..3 run MFMAs, waves 4..7 VALU work.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)
template <bool FP32>
__global__ __launch_bounds__(512) void k(float *out, int mfma_iters, int valu_iters, float a0) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + e + threadIdx.x); b[e] = (__bf16)(a0 - e); }
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (FP32) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0 + i, a0, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
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
template <bool FP32>
int run(const char *what, int mi, int vi) {
    float *out;
    CHECK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<FP32>), dim3(256), dim3(512), 0, 0, out, mi, vi, 1.f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-44s %8.3f ms\n", what, ms);
    CHECK(hipFree(out));
    return 0;
}
int main() {
    // 8 MFMAs per iteration at 16 cycles ~ 128 cycles; 32 VALU per iteration at ~4 cycles ~ 128 cycles
    const int N = 20000;
    run<false>("warm-up", N, N);
    run<false>("bf16 MFMA wave alone", N, 0);
    run<false>("VALU wave alone", 0, N);
    run<false>("bf16 MFMA wave + VALU wave on one SIMD", N, N);
    run<true>("fp32 MFMA wave alone", N / 2, 0);
    run<true>("fp32 MFMA wave + VALU wave on one SIMD", N / 2, N);
    return 0;
}
