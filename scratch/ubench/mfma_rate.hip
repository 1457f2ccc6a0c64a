// Peak-rate probe: independent v_mfma_f32_16x16x4_f32 chains, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    f32x4 acc[CH];
    for (int i = 0; i < CH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av = a + threadIdx.x, bv = b + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
void run(int blocks_per_cu, int iters) {
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 8 * CH * 2048.0;
    printf("chains=%d waves/SIMD=%d: %.3f ms, %.1f TFLOP/s\n", CH, blocks_per_cu, ms, flop / ms * 1e-9);
    hipFree(out);
}
int main() {
    run<1>(1, 20000);
    run<2>(1, 10000);
    run<4>(1, 5000);
    run<4>(2, 5000);
    run<2>(2, 10000);
    run<1>(2, 20000);
    run<1>(4, 20000);
    run<4>(4, 5000);
    return 0;
}
