// Probe for a later round: a 32 x 64 layer on a tile of 16 samples, (a) on v_mfma_f32_16x16x4_f32 and
// (b) on v_mfma_f32_16x16x32_bf16 with both operands split three ways into bf16 (x = hi + mid + lo, six
// products hh, hm, mh, hl, lh, mm).  Prints the error of both against fp64 and the MFMA time per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// W [32][64] row-major, X [64][16] row-major (k, sample); out [32][16]
__global__ void k_fp32(const float *W, const float *X, float *out, int iters) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    float a[2][16], b[16];
    for (int t = 0; t < 2; ++t)
        for (int s = 0; s < 16; ++s) a[t][s] = W[(16 * t + j) * 64 + 4 * s + g];
    for (int s = 0; s < 16; ++s) b[s] = X[(4 * s + g) * 16 + j];
    f32x4 d[2];
    for (int it = 0; it < iters; ++it) {
        d[0] = d[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s], b[s], d[t], 0, 0, 0);
        asm volatile("" : "+v"(d[0]), "+v"(d[1]));
    }
    for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 4; ++r) out[(16 * t + 4 * g + r) * 16 + j] = d[t][r];
}
__global__ void k_bf16(const float *W, const float *X, float *out, int iters) {
    const int lane = threadIdx.x, j = lane & 15, kg = lane >> 4, g = kg;
    bf16x8 a[2][2][3], b[2][3];   // [row tile][k step][term], [k step][term]
    for (int t = 0; t < 2; ++t)
        for (int q = 0; q < 2; ++q)
            for (int e = 0; e < 8; ++e) {
                __bf16 h, m, l;
                split3(W[(16 * t + j) * 64 + 32 * q + 8 * kg + e], h, m, l);
                a[t][q][0][e] = h; a[t][q][1][e] = m; a[t][q][2][e] = l;
            }
    for (int q = 0; q < 2; ++q)
        for (int e = 0; e < 8; ++e) {
            __bf16 h, m, l;
            split3(X[(32 * q + 8 * kg + e) * 16 + j], h, m, l);
            b[q][0][e] = h; b[q][1][e] = m; b[q][2][e] = l;
        }
    f32x4 d[2];
    for (int it = 0; it < iters; ++it) {
        d[0] = d[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // small terms first
        const int pa[6] = {1, 2, 0, 1, 0, 0}, pb[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][q][pa[p]], b[q][pb[p]], d[t], 0, 0, 0);
        asm volatile("" : "+v"(d[0]), "+v"(d[1]));
    }
    for (int t = 0; t < 2; ++t)
        for (int r = 0; r < 4; ++r) out[(16 * t + 4 * g + r) * 16 + j] = d[t][r];
}
int main() {
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> W(32 * 64), X(64 * 16);
    for (auto &v : W) v = nd(rng) * 0.2f;
    for (auto &v : X) v = nd(rng);
    std::vector<double> ref(32 * 16, 0.0);
    for (int o = 0; o < 32; ++o)
        for (int n = 0; n < 16; ++n)
            for (int k = 0; k < 64; ++k) ref[o * 16 + n] += (double)W[o * 64 + k] * (double)X[k * 16 + n];
    float *dW, *dX, *dO;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dO, 1 << 22);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> o(32 * 16);
    double scale = 0;
    for (double v : ref) scale = fmax(scale, fabs(v));
    for (int which = 0; which < 2; ++which) {
        if (which == 0) k_fp32<<<1, 64>>>(dW, dX, dO, 1); else k_bf16<<<1, 64>>>(dW, dX, dO, 1);
        hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost);
        double e = 0;
        for (int i = 0; i < 512; ++i) e = fmax(e, fabs((double)o[i] - ref[i]));
        printf("%s: max |error| %.3e (%.2e of the largest output)\n", which ? "bf16 x 3 split, 6 products" : "fp32 MFMA", e, e / scale);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int which = 0; which < 2; ++which)
        for (int waves = 1; waves <= 2; ++waves) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                // 1024 x waves workgroups of one wave: `waves` waves per SIMD
                if (which == 0) k_fp32<<<1024 * waves, 64>>>(dW, dX, dO, iters); else k_bf16<<<1024 * waves, 64>>>(dW, dX, dO, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s, %d wave(s) per SIMD: %.1f ns per tile per SIMD (%d MFMAs)\n", which ? "bf16 x 3" : "fp32", waves,
                   ms * 1e6 / iters / waves, which ? 24 : 32);
        }
    return 0;
}
