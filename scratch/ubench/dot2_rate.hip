// Issue-rate probe: v_dot2c_i32_i16 vs v_mad_i32_i24 vs v_fma_f32, independent chains, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int a, int b) {
    int acc[8];
    float facc[8];
    for (int i = 0; i < 8; ++i) { acc[i] = i; facc[i] = (float)i; }
    short2_t av = {(short)(a + threadIdx.x), (short)b};
    int bw = b * 65537 + threadIdx.x;
    float fa = (float)a, fb = (float)b * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_sdot2(av, __builtin_bit_cast(short2_t, bw), acc[i], false);
                if (MODE == 1) acc[i] = __mul24(acc[i], a) + bw;
                if (MODE == 2) facc[i] = __builtin_fmaf(facc[i], fb, fa);
            }
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i] + (int)facc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name) {
    int *out;
    (void)hipMalloc(&out, 256 * 4 * 256 * sizeof(int));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int grid = 256 * 4, iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10, 3, 5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 3, 5);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double lane_instr = (double)grid * 256 * iters * 64;
    printf("%-18s %.3f ms  %.1f T lane-instr/s\n", name, ms, lane_instr / ms * 1e-9);
    (void)hipFree(out);
}
int main() {
    run<0>("v_dot2_i32_i16");
    run<1>("v_mad_i32_i24");
    run<2>("v_fma_f32");
    return 0;
}
