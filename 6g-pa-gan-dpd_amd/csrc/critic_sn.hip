// Spectral normalisation of the critic's four Linear layers in one launch each way (reference
// models/discriminator.py:28-42: nn.utils.spectral_norm with its legacy hook semantics).
// Forward (one workgroup per layer): in training mode one power iteration
//     v <- normalize(W^T u),  u <- normalize(W v)      (in place on the module's buffers, eps = 1e-12)
// then sigma = u^T W v and the flat operand of the critic kernels wb = [W1/sigma1 | b1 | ... | W4/sigma4 | b4].
// The u, v used are also copied out: later forwards overwrite the buffers, the backward needs these.
// Backward: with u, v constants, W_bar = W / sigma(W), sigma = u^T W v:
//     dL/dW = (G - <G, W_bar> u v^T) / sigma,   dL/db = G_b.
#include "common.h"

namespace {

struct SnLayers {
    const float *w[4];
    const float *b[4];
    float *u[4];
    float *v[4];
};
__constant__ const int kOut[4] = {64, 32, 16, 1}, kIn[4] = {4, 64, 32, 16};
__constant__ const int kWOff[4] = {C_W1, C_W2, C_W3, C_W4}, kBOff[4] = {C_B1, C_B2, C_B3, C_B4};
__constant__ const int kUvOff[4] = {0, 68, 164, 212};      // u then v of each layer in the saved copy (229 floats)

DEV float block_sum64(float x, float *red) {   // sum over the first 64 threads' values (others pass 0)
    x = wave_sum(x);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// n_eval successive forwards in one launch (the critic update evaluates the critic three times, each evaluation with
// its own power iteration): evaluation k writes wb + k * C_NP, uv_saved + k * 229, sigma_out + k * 4; u and v stay in
// LDS between the evaluations.  Operation for operation what n_eval launches with n_eval = 1 compute.
__global__ __launch_bounds__(256) void k_sn_forward(SnLayers p, int power_iteration, float eps, int n_eval, float *__restrict__ wb,
                                                    float *__restrict__ uv_saved, float *__restrict__ sigma_out) {
    __shared__ float su[64], sv[64], red[4];
    const int l = blockIdx.x, O = kOut[l], I = kIn[l], tid = threadIdx.x;
    const float *W = p.w[l];
    if (tid < O) su[tid] = p.u[l][tid];
    if (tid < I) sv[tid] = p.v[l][tid];
    __syncthreads();
    for (int k = 0; k < n_eval; ++k, wb += C_NP, uv_saved += 229, sigma_out += 4) {
    if (power_iteration) {
        float t = 0.f;
        if (tid < I)
            for (int i = 0; i < O; ++i) t = __builtin_fmaf(W[i * I + tid], su[i], t);
        const float nv = sqrtf(block_sum64(tid < I ? t * t : 0.f, red));
        if (tid < I) sv[tid] = t / fmaxf(nv, eps);
        __syncthreads();
    }
    float s = 0.f;
    if (tid < O)
        for (int j = 0; j < I; ++j) s = __builtin_fmaf(W[tid * I + j], sv[j], s);
    if (power_iteration) {
        const float nu = sqrtf(block_sum64(tid < O ? s * s : 0.f, red));
        if (tid < O) su[tid] = s / fmaxf(nu, eps);
        __syncthreads();
    }
    const float sigma = block_sum64(tid < O ? su[tid] * s : 0.f, red);
    if (tid < O) uv_saved[kUvOff[l] + tid] = su[tid];
    if (tid < I) uv_saved[kUvOff[l] + O + tid] = sv[tid];
    if (tid == 0) sigma_out[l] = sigma;
    for (int e = tid; e < O * I; e += 256) wb[kWOff[l] + e] = W[e] / sigma;
    if (tid < O) wb[kBOff[l] + tid] = p.b[l][tid];
    __syncthreads();
    }
    if (power_iteration) {
        if (tid < O) p.u[l][tid] = su[tid];
        if (tid < I) p.v[l][tid] = sv[tid];
    }
}

// n_eval = 1: gout = the gradient of one evaluation.  n_eval = 3 (the critic update: real, fake, interpolate):
// gout = (g_0 + g_1) + coef2 * g_2 with g_k the gradient of evaluation k (gwb + k * C_NP, uv_saved + k * 229,
// sigma_in + k * 4), the sum formed in that order with separate roundings.
__global__ __launch_bounds__(256) void k_sn_backward(SnLayers p, const float *__restrict__ gwb,
                                                     const float *__restrict__ uv_saved, const float *__restrict__ sigma_in,
                                                     int n_eval, float coef2, float *__restrict__ gout) {
    __shared__ double red[3][4];
    const int l = blockIdx.x, O = kOut[l], I = kIn[l], tid = threadIdx.x;
    const float *W = p.w[l];
    // <G, W> in fp64: the result is subtracted from G entry by entry (cancellation)
    for (int k = 0; k < n_eval; ++k) {
        double part = 0.0;
        for (int e = tid; e < O * I; e += 256) part += (double)gwb[k * C_NP + kWOff[l] + e] * (double)W[e];
        part = wave_sum_d(part);
        if ((tid & 63) == 0) red[k][tid >> 6] = part;
    }
    __syncthreads();
    float inner[3], sigma[3];
    for (int k = 0; k < n_eval; ++k) {
        sigma[k] = sigma_in[4 * k + l];
        inner[k] = (float)(((red[k][0] + red[k][1]) + (red[k][2] + red[k][3])) / (double)sigma[k]);     // <G, W / sigma>
    }
    auto term = [&](int k, int e, int i, int j) {
        const float *u = uv_saved + 229 * k + kUvOff[l], *v = u + O;
        return (gwb[k * C_NP + kWOff[l] + e] - inner[k] * u[i] * v[j]) / sigma[k];
    };
    for (int e = tid; e < O * I; e += 256) {
        const int i = e / I, j = e - i * I;
        float g = term(0, e, i, j);
        if (n_eval == 3) g = (g + term(1, e, i, j)) + coef2 * term(2, e, i, j);
        gout[kWOff[l] + e] = g;
    }
    if (tid < O) {
        float g = gwb[kBOff[l] + tid];
        if (n_eval == 3) g = (g + gwb[C_NP + kBOff[l] + tid]) + coef2 * gwb[2 * C_NP + kBOff[l] + tid];
        gout[kBOff[l] + tid] = g;
    }
}

int fill(SnLayers &p, const float *const *w, const float *const *b, float *const *u, float *const *v) {
    for (int l = 0; l < 4; ++l) {
        DPD_REQUIRE(w[l] && b[l] && u[l] && v[l], DPD_ERR_INVALID_ARG, "critic_sn: null pointer for layer %d", l);
        p.w[l] = w[l]; p.b[l] = b[l]; p.u[l] = u[l]; p.v[l] = v[l];
    }
    return DPD_OK;
}

}  // namespace

extern "C" int dpd_critic_sn_forward(const float *const *w_orig, const float *const *bias, float *const *u,
                                     float *const *v, int power_iteration, float eps, float *wb, float *uv_saved,
                                     float *sigma, void *stream) {
    DPD_REQUIRE(w_orig && bias && u && v && wb && uv_saved && sigma, DPD_ERR_INVALID_ARG, "critic_sn_forward: null pointer");
    SnLayers p;
    int rc = fill(p, w_orig, bias, u, v);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sn_forward, dim3(4), dim3(256), 0, as_stream(stream), p, power_iteration, eps, 1, wb, uv_saved, sigma);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_critic_sn_forward3(const float *const *w_orig, const float *const *bias, float *const *u,
                                      float *const *v, int power_iteration, float eps, float *wb, float *uv_saved,
                                      float *sigma, void *stream) {
    DPD_REQUIRE(w_orig && bias && u && v && wb && uv_saved && sigma, DPD_ERR_INVALID_ARG, "critic_sn_forward3: null pointer");
    SnLayers p;
    int rc = fill(p, w_orig, bias, u, v);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sn_forward, dim3(4), dim3(256), 0, as_stream(stream), p, power_iteration, eps, 3, wb, uv_saved, sigma);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_critic_sn_backward(const float *const *w_orig, const float *gwb, const float *uv_saved,
                                      const float *sigma, float *grads, void *stream) {
    DPD_REQUIRE(w_orig && gwb && uv_saved && sigma && grads, DPD_ERR_INVALID_ARG, "critic_sn_backward: null pointer");
    SnLayers p;
    for (int l = 0; l < 4; ++l) {
        DPD_REQUIRE(w_orig[l], DPD_ERR_INVALID_ARG, "critic_sn_backward: null weight pointer for layer %d", l);
        p.w[l] = w_orig[l]; p.b[l] = nullptr; p.u[l] = nullptr; p.v[l] = nullptr;
    }
    hipLaunchKernelGGL(k_sn_backward, dim3(4), dim3(256), 0, as_stream(stream), p, gwb, uv_saved, sigma, 1, 0.f, grads);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_critic_sn_backward3(const float *const *w_orig, const float *gwb, const float *uv_saved,
                                       const float *sigma, float coef2, float *grads, void *stream) {
    DPD_REQUIRE(w_orig && gwb && uv_saved && sigma && grads, DPD_ERR_INVALID_ARG, "critic_sn_backward3: null pointer");
    SnLayers p;
    for (int l = 0; l < 4; ++l) {
        DPD_REQUIRE(w_orig[l], DPD_ERR_INVALID_ARG, "critic_sn_backward3: null weight pointer for layer %d", l);
        p.w[l] = w_orig[l]; p.b[l] = nullptr; p.u[l] = nullptr; p.v[l] = nullptr;
    }
    hipLaunchKernelGGL(k_sn_backward, dim3(4), dim3(256), 0, as_stream(stream), p, gwb, uv_saved, sigma, 3, coef2, grads);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}
