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

}  // namespace

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
