// Probe: one wave per SIMD issuing v_mfma_f32_16x16x4_f32 on CH accumulator chains with NV independent VALU
// FMAs between consecutive MFMAs: does VALU work issue in the shadow of the MFMA pipe, and how long must a
// dependent MFMA wait for its accumulator?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH, int NV>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    f32x4 acc[CH];
    for (int i = 0; i < CH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av = a + threadIdx.x, bv = b + threadIdx.x;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NV; ++n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n % 8]) : "v"(bv));
            }
    }
    float s = 0.f;
    for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH, int NV>
void run(int blocks_per_cu, int iters) {
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k<CH, NV>), dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CH, NV>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double n_mfma = (double)iters * 8 * CH;   // per wave
    printf("chains=%d valu/mfma=%2d waves/SIMD=%d: %.3f ms, %.1f ns per MFMA per SIMD (%.1f TFLOP/s)\n", CH, NV, blocks_per_cu, ms,
           ms * 1e6 / n_mfma / blocks_per_cu, (double)grid * 4 * n_mfma * 2048.0 / ms * 1e-9);
    hipFree(out);
}
int main() {
    run<1, 0>(1, 4000);
    run<2, 0>(1, 2000);
    run<4, 0>(1, 1000);
    run<2, 4>(1, 2000);
    run<2, 8>(1, 2000);
    run<2, 12>(1, 2000);
    run<2, 16>(1, 2000);
    run<4, 8>(1, 1000);
    run<2, 8>(2, 2000);
    run<2, 16>(2, 2000);
    return 0;
}
