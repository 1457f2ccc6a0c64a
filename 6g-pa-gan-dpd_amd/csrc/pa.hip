// Memory-polynomial PA twin (reference models/pa_digital_twin.py:125-170):
//   y[n] = sum_k sum_m h[k,m] * x[n+M-1-m] * |x[n+M-1-m]|^(2k),  k = 0..K-1 (orders 1,3,5,..), m = 0..M-1
// evaluated per tap as x * P_m(|x|^2) with P_m a complex polynomial (Horner).  Forward: one lane per
// output sample.  Backward: one lane per INPUT sample gathers its <= M contributions, so there
// are no atomics; d|x|/dx at 0 is 0 because every k >= 1 term carries a factor x.
#include "common.h"

namespace {

constexpr int MAXK = 8, MAXM = 16;

struct Coefs {
    float re[MAXK * MAXM];
    float im[MAXK * MAXM];
};

DEV void load_coefs(float *sre, float *sim, const float *__restrict__ cre, const float *__restrict__ cim, int K, int M) {
    for (int i = threadIdx.x; i < K * M; i += blockDim.x) {
        sre[i] = cre[i];
        sim[i] = cim[i];
    }
    __syncthreads();
}

// QUANT: x is the generator's unquantised output and absmax3 its abs-max: the 16-bit output quantiser is
// applied on load and the quantised samples are written back in place.  Quantising is idempotent for a
// fixed scale, so it does not matter whether a neighbour lane has already rewritten a sample.
template <bool QUANT>
__global__ __launch_bounds__(256) void k_pa_fwd(float2 *__restrict__ x, int64_t S, int64_t OL, int64_t N,
                                                const float *__restrict__ cre, const float *__restrict__ cim, int K,
                                                int M, const float *__restrict__ absmax3, float2 *__restrict__ y) {
    __shared__ float sre[MAXK * MAXM], sim[MAXK * MAXM];
    load_coefs(sre, sim, cre, cim, K, M);
    QS qs = qs_one();
    if (QUANT) qs = qs_of(absmax3[0]);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / OL, n = i - b * OL;
        float2 *row = x + b * S + n;
        float yr = 0.f, yi = 0.f;
        for (int m = 0; m < M; ++m) {
            float2 v = row[M - 1 - m];
            if (QUANT) {
                v.x = fq16(v.x, qs);
                v.y = fq16(v.y, qs);
                // sample n + M - 1 - m is rewritten by the output that reads it through tap M - 1 (the
                // row's first output also rewrites the M - 1 samples before it)
                if (m == 0 || n == 0) row[M - 1 - m] = v;
            }
            const float p = v.x * v.x + v.y * v.y;
            float pr = sre[(K - 1) * M + m], pi = sim[(K - 1) * M + m];
            for (int k = K - 2; k >= 0; --k) {
                pr = __builtin_fmaf(pr, p, sre[k * M + m]);
                pi = __builtin_fmaf(pi, p, sim[k * M + m]);
            }
            yr += v.x * pr - v.y * pi;
            yi += v.x * pi + v.y * pr;
        }
        y[i] = make_float2(yr, yi);
    }
}

__global__ __launch_bounds__(256) void k_pa_bwd(const float2 *__restrict__ x, int64_t S, int64_t OL, int64_t NX,
                                                const float *__restrict__ cre, const float *__restrict__ cim, int K,
                                                int M, const float2 *__restrict__ gy, float2 *__restrict__ gx) {
    __shared__ float sre[MAXK * MAXM], sim[MAXK * MAXM];
    load_coefs(sre, sim, cre, cim, K, M);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < NX; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / S, t = i - b * S;
        const float2 v = x[i];
        const float p = v.x * v.x + v.y * v.y;
        float gr = 0.f, gi = 0.f;
        for (int m = 0; m < M; ++m) {
            const int64_t n = t - (M - 1) + m;   // output that reads x[t] through tap m
            if (n < 0 || n >= OL) continue;
            const float2 c = gy[b * OL + n];
            // P_m(p) and P_m'(p)
            float pr = sre[(K - 1) * M + m], pi = sim[(K - 1) * M + m], dr = 0.f, di = 0.f;
            for (int k = K - 2; k >= 0; --k) {
                dr = __builtin_fmaf(dr, p, pr);
                di = __builtin_fmaf(di, p, pi);
                pr = __builtin_fmaf(pr, p, sre[k * M + m]);
                pi = __builtin_fmaf(pi, p, sim[k * M + m]);
            }
            // f = x * P(p): fr = xr Pr - xi Pi, fi = xr Pi + xi Pr
            const float ar = v.x * dr - v.y * di;   // d fr / d p
            const float ai = v.x * di + v.y * dr;   // d fi / d p
            const float s = 2.f * (c.x * ar + c.y * ai);
            gr += c.x * pr + c.y * pi + s * v.x;
            gi += -c.x * pi + c.y * pr + s * v.y;
        }
        gx[i] = make_float2(gr, gi);
    }
}

// gy = (y - t) * scale element by element, and one partial sum of (y - t)^2 per workgroup (fixed order inside).
// gy may be y itself (the chain's fallback path overwrites the PA output with its gradient: every element is read
// and then written by the same thread), so neither of the two is declared restrict.
__global__ __launch_bounds__(256) void k_mse_grad(const float *y, const float *__restrict__ t, int64_t n,
                                                  float scale, float *gy, double *__restrict__ part) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = y[i] - t[i];
        gy[i] = d * scale;
        acc += (double)d * (double)d;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void k_mse_final(const double *__restrict__ part, int n, double inv_n, float *__restrict__ loss) {
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) a += part[i];
    a = wave_sum_d(a);
    if (threadIdx.x == 0) loss[0] = (float)(a * inv_n);
}

}  // namespace

// gy = d mean((y - t)^2) / d y over n elements and loss[0] = that mean; part: scratch of 1024 doubles
int pa_launch_mse_grad(const float *y, const float *t, int64_t n, float *gy, double *part, float *loss, hipStream_t st) {
    const int grid = grid_for(n, 256, 1024);
    hipLaunchKernelGGL(k_mse_grad, dim3(grid), dim3(256), 0, st, y, t, n, (float)(2.0 / (double)n), gy, part);
    DPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mse_final, dim3(1), dim3(64), 0, st, part, grid, 1.0 / (double)n, loss);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

static int pa_check(const float *x, int64_t B, int64_t S, const float *cre, const float *cim, int K, int M) {
    DPD_REQUIRE((x || B == 0) && cre && cim, DPD_ERR_INVALID_ARG, "pa: null pointer");
    DPD_REQUIRE(K >= 1 && K <= MAXK && M >= 1 && M <= MAXM, DPD_ERR_INVALID_ARG, "pa: K=%d (1..8) or M=%d (1..16) unsupported", K, M);
    DPD_REQUIRE(B >= 0, DPD_ERR_INVALID_ARG, "pa: negative batch");
    DPD_REQUIRE(S >= M, DPD_ERR_SHAPE, "Sequence length %lld must be >= memory depth %d", (long long)S, M);
    return DPD_OK;
}

extern "C" int dpd_pa_forward(const float *x, int64_t B, int64_t S, const float *coef_re, const float *coef_im, int K,
                              int M, float *y, void *stream) {
    int rc = pa_check(x, B, S, coef_re, coef_im, K, M);
    if (rc) return rc;
    DPD_REQUIRE(y || B == 0, DPD_ERR_INVALID_ARG, "pa_forward: y is null");
    const int64_t OL = S - M + 1, N = B * OL;
    if (N == 0) return DPD_OK;
    hipLaunchKernelGGL(k_pa_fwd<false>, dim3(grid_for(N, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<float2 *>(const_cast<float *>(x)), S, OL, N, coef_re, coef_im, K, M, nullptr,
                       reinterpret_cast<float2 *>(y));
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_quant_pa_forward(float *x, int64_t B, int64_t S, const float *absmax3, const float *coef_re,
                                    const float *coef_im, int K, int M, float *y, void *stream) {
    int rc = pa_check(x, B, S, coef_re, coef_im, K, M);
    if (rc) return rc;
    DPD_REQUIRE((y || B == 0) && absmax3, DPD_ERR_INVALID_ARG, "quant_pa_forward: null pointer");
    const int64_t OL = S - M + 1, N = B * OL;
    if (N == 0) return DPD_OK;
    hipLaunchKernelGGL(k_pa_fwd<true>, dim3(grid_for(N, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<float2 *>(x), S, OL, N, coef_re, coef_im, K, M, absmax3,
                       reinterpret_cast<float2 *>(y));
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_pa_backward(const float *x, int64_t B, int64_t S, const float *coef_re, const float *coef_im, int K,
                               int M, const float *gy, float *gx, void *stream) {
    int rc = pa_check(x, B, S, coef_re, coef_im, K, M);
    if (rc) return rc;
    DPD_REQUIRE((gy && gx) || B == 0, DPD_ERR_INVALID_ARG, "pa_backward: null gradient pointer");
    const int64_t OL = S - M + 1, NX = B * S;
    if (NX == 0) return DPD_OK;
    hipLaunchKernelGGL(k_pa_bwd, dim3(grid_for(NX, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(x), S, OL, NX, coef_re, coef_im, K, M,
                       reinterpret_cast<const float2 *>(gy), reinterpret_cast<float2 *>(gx));
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}
