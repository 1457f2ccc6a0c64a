// Probe: cost of different VALU instruction kinds issued between fp32 MFMAs (one wave per SIMD, two
// accumulator chains, 8 VALU instructions per MFMA).  The MFMA alone takes ~14.5 ns per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float av = a + threadIdx.x, bv = b + threadIdx.x;
    float v[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) {
        v[i] = a * i;
        p[i] = f32x2{a * i, b * i};
    }
    const f32x2 pb = {bv, av};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n]) : "v"(bv));
                    if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[n]) : "v"(bv));
                    if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "+v"(v[n]) : "v"(bv));
                    if (KIND == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[n]) : "v"(bv));
                    if (KIND == 4) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[n]) : "v"(bv) : "vcc");
                    if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[n]) : "v"(pb));
                    if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[n]) : "v"(pb));
                    if (KIND == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[n]) : "v"(bv));
                    if (KIND == 8) asm volatile("s_nop 0");
                    if (KIND == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[n]) : "v"(bv));
                    if (KIND == 10) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[n]));
                    if (KIND == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[n]) : "v"(pb));
                    if (KIND == 12) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[n]) : "v"(bv));
                    if (KIND == 13) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[n]));
                    if (KIND == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[n]) : "v"(bv));
                    if (KIND == 15) asm volatile("v_exp_f32 %0, %0" : "+v"(v[n]));
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
void run(const char *name, int waves = 1) {
    const int iters = 2000;
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));   // up to 8 workgroups per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(256 * waves), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(256 * waves), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (iters * 16.0) / waves;   // per MFMA per SIMD
    printf("%-28s waves/SIMD %d: %.1f ns per (MFMA + 8 instr): %.2f ns per instruction on top of 14.5\n", name, waves, per, (per - 14.5) / 8);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32");
    run<1>("v_mul_f32");
    run<2>("v_mov_b32");
    run<3>("v_and_b32");
    run<4>("v_cmp_gt_f32 + v_cndmask (2)");
    run<5>("v_pk_fma_f32");
    run<6>("v_pk_mul_f32");
    run<7>("v_max_f32");
    run<8>("s_nop 0");
    run<9>("v_add_u32");
    run<10>("v_lshlrev_b32");
    run<11>("v_pk_add_f32");
    run<12>("v_med3_f32");
    run<13>("v_rndne_f32");
    run<14>("v_cndmask_b32");
    run<15>("v_exp_f32");
    run<0>("v_fma_f32", 2);
    run<0>("v_fma_f32", 4);
    run<5>("v_pk_fma_f32", 4);
    run<2>("v_mov_b32", 4);
    run<12>("v_med3_f32", 4);
    return 0;
}
