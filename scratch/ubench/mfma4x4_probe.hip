// Probe: operand/result layout and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_layout(float *out) {
    const int lane = threadIdx.x;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + lane), (float)(100 + lane), d, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[i * 64 + lane] = d[i];
}
__global__ void k_rate(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;
    float a = lane * 1e-3f, b = 1.f;
    for (int it = 0; it < iters; ++it) {
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3];
}
__global__ void k_rate_dep(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f};
    float a = lane * 1e-3f, b = 1.f;
    for (int it = 0; it < iters; ++it) {
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0];
}
int main() {
    float *d;
    hipMalloc(&d, 1 << 22);
    k_layout<<<1, 64>>>(d);
    std::vector<float> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i)
        for (int l = 0; l < 64; ++l) {
            int fa = -1, fb = -1;
            for (int la = 0; la < 64 && fa < 0; ++la)
                for (int lb = 0; lb < 64; ++lb)
                    if ((float)(1 + la) * (float)(100 + lb) == h[i * 64 + l]) { fa = la; fb = lb; break; }
            if (l < 12 || l > 59) printf("vgpr %d lane %2d = A lane %2d x B lane %2d\n", i, l, fa, fb);
        }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int dep = 0; dep < 2; ++dep) {
        const int iters = 100000, blocks = 1024, threads = 256;   // one wave per SIMD
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (dep) k_rate_dep<<<blocks, threads>>>(d, iters); else k_rate<<<blocks, threads>>>(d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double per = ms * 1e-3 / (4.0 * iters) * 2.4e9;   // one wave per SIMD: time per instruction in 2.4 GHz cycles
        printf("%s: %.3f ms, %.2f cycles (at 2.4 GHz) per 4x4x1 per wave\n", dep ? "dependent" : "independent", ms, per);
    }
    return 0;
}
