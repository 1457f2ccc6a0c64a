// Probe: issue rate of v_mfma_f32_16x16x16_bf16 (the K = 16 form) beside the K = 32 form, bare and with VALU fillers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int NV, bool K16>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + e + threadIdx.x); b[e] = (__bf16)(a0 - e); }
    s16x4 a4 = {(short)(threadIdx.x), 1, 2, 3}, b4 = {4, 5, 6, (short)(a0)};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a0 * i;
    const float bv = a0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (K16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NV; ++n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n % 8]) : "v"(bv));
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, bool K16>
void run(int waves) {
    const int iters = 4000;
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, K16>), dim3(256 * waves), dim3(256), 0, 0, out, iters, 1.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("bf16 16x16x%d + %2d v_fma_f32 each, %d wave(s) per SIMD: %.2f ns per MFMA per SIMD\n", K16 ? 16 : 32, NV, waves, ms * 1e6 / (iters * 16.0) / waves);
    hipFree(out);
}
int main() {
    run<0, false>(1); run<0, true>(1);
    run<2, false>(1); run<2, true>(1);
    run<4, false>(1); run<4, true>(1);
    run<0, false>(2); run<0, true>(2);
    run<2, true>(2); run<4, false>(2); run<4, true>(2);
    return 0;
}
