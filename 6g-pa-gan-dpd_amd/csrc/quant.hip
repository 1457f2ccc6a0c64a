// Stand-alone quantisation primitives: fixed-scale (utils/quantization.py:88-161 of the reference)
// and data-scaled (models/tdnn_generator.py:74-103).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void k_quantize_fixed(const float *__restrict__ x, int64_t n, float scale,
                                                        float qmin, float qmax, int dequant,
                                                        float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float q = fminf(fmaxf(rintf(x[i] * scale), qmin), qmax);
        out[i] = dequant ? q / scale : q;
    }
}

// rows == 1: grid-stride over the tensor, one atomic max per wave; rows > 1: blockIdx.y = row
__global__ __launch_bounds__(256) void k_row_absmax(const float *__restrict__ x, int64_t cols,
                                                    float *__restrict__ absmax) {
    const float *row = x + (int64_t)blockIdx.y * cols;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cols; i += (int64_t)gridDim.x * 256)
        m = fmaxf(m, fabsf(row[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomic_max_nonneg(absmax + blockIdx.y, m);
}

__global__ __launch_bounds__(256) void k_row_fq(const float *__restrict__ x, int64_t cols, float qmax,
                                                const float *__restrict__ absmax, float *__restrict__ out) {
    const int64_t base = (int64_t)blockIdx.y * cols;
    const float s = scale_of(absmax[blockIdx.y], qmax);
    const float qmin = -qmax - 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cols; i += (int64_t)gridDim.x * 256) {
        float q = fminf(fmaxf(rintf(x[base + i] / s), qmin), qmax);
        out[base + i] = q * s;
    }
}

struct QRange {
    float scale, qmin, qmax;
};
DEV float fq_fixed(float v, QRange r) { return fminf(fmaxf(rintf(v * r.scale), r.qmin), r.qmax) / r.scale; }

// out[n, o] = fq_a( sum_i fq_a(x[n, i]) * fq_w(w[o, i]) + bias[o] ).  Every product of two fp32 values is
// exact in fp64 and the fp64 sum of `in` of them carries an error far below the output grid, so the result
// is that of exact arithmetic (for the power-of-two scales of the Q8.8 / Q1.15 formats: of the integer
// datapath) and does not depend on a summation order.  One lane per output; the row of x is re-read by the
// lanes of a workgroup through the cache.
__global__ __launch_bounds__(256) void k_qlinear_fwd(const float *__restrict__ x, const float *__restrict__ w,
                                                     const float *__restrict__ bias, int64_t N, int in, int out_f,
                                                     QRange ra, QRange rw, float *__restrict__ out) {
    const int64_t total = N * out_f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t n = e / out_f;
        const int o = (int)(e - n * out_f);
        const float *xr = x + n * in, *wr = w + (int64_t)o * in;
        double acc = 0.0;
        for (int i = 0; i < in; ++i) acc += (double)fq_fixed(xr[i], ra) * (double)fq_fixed(wr[i], rw);
        if (bias) acc += (double)bias[o];
        const double q = fmin(fmax(rint(acc * (double)ra.scale), (double)ra.qmin), (double)ra.qmax);
        out[e] = (float)q / ra.scale;
    }
}

// Straight-through backward: gx = gy fq_w(W), gw = gy^T fq_a(x), gb = column sums of gy (fp32, fixed order).
__global__ __launch_bounds__(256) void k_qlinear_gx(const float *__restrict__ gy, const float *__restrict__ w, int64_t N,
                                                    int in, int out_f, QRange rw, float *__restrict__ gx) {
    const int64_t total = N * in;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t n = e / in;
        const int i = (int)(e - n * in);
        float acc = 0.f;
        for (int o = 0; o < out_f; ++o) acc = __builtin_fmaf(gy[n * out_f + o], fq_fixed(w[(int64_t)o * in + i], rw), acc);
        gx[e] = acc;
    }
}
__global__ __launch_bounds__(256) void k_qlinear_gw(const float *__restrict__ gy, const float *__restrict__ x, int64_t N,
                                                    int in, int out_f, QRange ra, float *__restrict__ gw,
                                                    float *__restrict__ gb) {
    const int total = out_f * (in + 1);          // column `in` of a row is the bias gradient
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int o = e / (in + 1), i = e - o * (in + 1);
        double acc = 0.0;
        for (int64_t n = 0; n < N; ++n) acc += (double)gy[n * out_f + o] * (i < in ? (double)fq_fixed(x[n * in + i], ra) : 1.0);
        if (i < in) gw[o * in + i] = (float)acc;
        else if (gb) gb[o] = (float)acc;
    }
}

// EMA update of n running scales from n abs-max values, `times` times over (FakeQuantizeModule.observe,
// models/tdnn_generator.py:121-130: scale = clamp(absmax, 1e-8) / q_max; running = (1 - momentum) * running +
// momentum * scale, every product and the sum rounded separately as the element-wise tensor arithmetic does)
__global__ __launch_bounds__(64) void k_qat_observe(const float *__restrict__ absmax, float *__restrict__ running, int n,
                                                    float keep, float momentum, float q_max, int times) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float scale = fmaxf(absmax[i], 1e-8f) / q_max;
    const float add = momentum * scale;
    float r = running[i];
    for (int k = 0; k < times; ++k) r = keep * r + add;
    running[i] = r;
}

}  // namespace

static QRange qrange_of(float scale, int bits, int is_signed) {
    return QRange{scale, is_signed ? -(float)(1 << (bits - 1)) : 0.f,
                  is_signed ? (float)((1 << (bits - 1)) - 1) : (float)((1 << bits) - 1)};
}

extern "C" int dpd_quantized_linear(const float *x, const float *w, const float *bias, int64_t N, int in_features,
                                    int out_features, float act_scale, int act_bits, int act_signed, float w_scale,
                                    int w_bits, int w_signed, float *out, void *stream) {
    DPD_REQUIRE(((x && out) || N == 0) && w && N >= 0 && in_features >= 1 && out_features >= 1, DPD_ERR_INVALID_ARG,
                "quantized_linear: bad argument");
    DPD_REQUIRE(act_bits >= 2 && act_bits <= 24 && w_bits >= 2 && w_bits <= 24, DPD_ERR_INVALID_ARG, "quantized_linear: bits not in 2..24");
    if (N == 0) return DPD_OK;
    hipLaunchKernelGGL(k_qlinear_fwd, dim3(grid_for(N * out_features, 256)), dim3(256), 0, as_stream(stream), x, w, bias, N,
                       in_features, out_features, qrange_of(act_scale, act_bits, act_signed),
                       qrange_of(w_scale, w_bits, w_signed), out);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_quantized_linear_backward(const float *x, const float *w, const float *gy, int64_t N, int in_features,
                                             int out_features, float act_scale, int act_bits, int act_signed,
                                             float w_scale, int w_bits, int w_signed, float *gx, float *gw, float *gb,
                                             void *stream) {
    DPD_REQUIRE(((x && gy) || N == 0) && w && N >= 0 && in_features >= 1 && out_features >= 1, DPD_ERR_INVALID_ARG,
                "quantized_linear_backward: bad argument");
    hipStream_t st = as_stream(stream);
    if (gx && N > 0) {
        hipLaunchKernelGGL(k_qlinear_gx, dim3(grid_for(N * in_features, 256)), dim3(256), 0, st, gy, w, N, in_features,
                           out_features, qrange_of(w_scale, w_bits, w_signed), gx);
        DPD_LAUNCH_CHECK();
    }
    if (gw) {
        hipLaunchKernelGGL(k_qlinear_gw, dim3(grid_for((int64_t)out_features * (in_features + 1), 256)), dim3(256), 0, st, gy,
                           x, N, in_features, out_features, qrange_of(act_scale, act_bits, act_signed), gw, gb);
        DPD_LAUNCH_CHECK();
    }
    return DPD_OK;
}

extern "C" int dpd_quantize_fixed(const float *x, int64_t n, float scale, int bits, int is_signed, int dequant,
                                  float *out, void *stream) {
    DPD_REQUIRE(((x && out) || n == 0) && n >= 0, DPD_ERR_INVALID_ARG, "quantize_fixed: bad argument");
    DPD_REQUIRE(bits >= 2 && bits <= 24, DPD_ERR_INVALID_ARG, "quantize_fixed: bits %d not in 2..24", bits);
    if (n == 0) return DPD_OK;
    float qmin = is_signed ? -(float)(1 << (bits - 1)) : 0.f;
    float qmax = is_signed ? (float)((1 << (bits - 1)) - 1) : (float)((1 << bits) - 1);
    hipLaunchKernelGGL(k_quantize_fixed, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), x, n, scale, qmin,
                       qmax, dequant, out);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_fake_quantize_dynamic(const float *x, int64_t rows, int64_t cols, int per_row, int n_bits,
                                         float *out, float *absmax_out, void *stream) {
    DPD_REQUIRE(((x && out) || rows == 0 || cols == 0) && absmax_out, DPD_ERR_INVALID_ARG, "fake_quantize_dynamic: null pointer");
    DPD_REQUIRE(rows >= 0 && cols >= 0 && rows <= 65535, DPD_ERR_INVALID_ARG, "fake_quantize_dynamic: bad shape");
    DPD_REQUIRE(n_bits >= 2 && n_bits <= 24, DPD_ERR_INVALID_ARG, "fake_quantize_dynamic: n_bits %d", n_bits);
    if (rows * cols == 0) return DPD_OK;
    if (!per_row) {
        cols = rows * cols;
        rows = 1;
    }
    hipStream_t st = as_stream(stream);
    DPD_HIP(hipMemsetAsync(absmax_out, 0, rows * sizeof(float), st));
    const int gx = grid_for(cols, 256, rows == 1 ? 2048 : 64);
    hipLaunchKernelGGL(k_row_absmax, dim3(gx, (unsigned)rows), dim3(256), 0, st, x, cols, absmax_out);
    DPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_row_fq, dim3(gx, (unsigned)rows), dim3(256), 0, st, x, cols,
                       (float)((1 << (n_bits - 1)) - 1), absmax_out, out);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_qat_observe(const float *absmax, float *running, int n, float momentum, int n_bits, int times,
                               void *stream) {
    DPD_REQUIRE(absmax && running, DPD_ERR_INVALID_ARG, "qat_observe: null pointer");
    DPD_REQUIRE(n >= 1 && n <= 64 && times >= 0, DPD_ERR_INVALID_ARG, "qat_observe: n %d not in 1..64 or times %d < 0", n, times);
    DPD_REQUIRE(n_bits >= 2 && n_bits <= 24, DPD_ERR_INVALID_ARG, "qat_observe: n_bits %d", n_bits);
    if (times == 0) return DPD_OK;
    // (1 - momentum) is formed in double and rounded once, as the host language does before it meets the tensor
    hipLaunchKernelGGL(k_qat_observe, dim3(1), dim3(64), 0, as_stream(stream), absmax, running, n,
                       (float)(1.0 - (double)momentum), momentum, (float)((1 << (n_bits - 1)) - 1), times);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}
