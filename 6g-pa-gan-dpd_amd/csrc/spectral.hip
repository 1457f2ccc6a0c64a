// Per-row spectral statistics and their gradient (reference utils/spectral_loss.py:24-167):
// one workgroup per IQ row, the row's L-point complex FFT done in LDS (in-place radix-2,
// decimation in time, twiddles from sincospi), band powers from the reference's three fftfreq
// masks.  The loss arithmetic on the [B] statistics (dB conversion, clamps, means) is left to the
// caller; the backward kernel takes the gradient w.r.t. the six statistics of each row:
//   d sum|p-t| / dp = sign(p-t);  d sum|p-t|^2 / dp = 2 (p-t);
//   d P_band / dp   = (2/L) * IDFT_unnormalised(mask * X),  X = FFT(p),  P_band = sum mask |X|^2 / L.
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXL = 4096;

DEV unsigned bitrev(unsigned i, int log2n) { return __brev(i) >> (32 - log2n); }

DEV void make_twiddles(float2 *tw, int L) {
    for (int k = threadIdx.x; k < L / 2; k += TPB) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)L, &s, &c);
        tw[k] = make_float2(c, s);
    }
}

// in-place DIT FFT of data already in bit-reversed order; natural order on return
DEV void fft_stages(float2 *d, const float2 *tw, int L, int log2n) {
    for (int s = 0; s < log2n; ++s) {
        const int half = 1 << s;
        const int tstep = L >> (s + 1);
        __syncthreads();
        for (int t = threadIdx.x; t < L / 2; t += TPB) {
            const int grp = t >> s, pos = t & (half - 1);
            const int i = (grp << (s + 1)) + pos, j = i + half;
            const float2 w = tw[pos * tstep];
            const float2 a = d[i], b = d[j];
            const float br = b.x * w.x - b.y * w.y, bi = b.x * w.y + b.y * w.x;
            d[i] = make_float2(a.x + br, a.y + bi);
            d[j] = make_float2(a.x - br, a.y - bi);
        }
    }
    __syncthreads();
}

DEV float block_sum(float v, float *red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(TPB) void k_spectral_stats(const float2 *__restrict__ pred,
                                                        const float2 *__restrict__ target, int64_t L, int log2n,
                                                        const uint8_t *__restrict__ band, float *__restrict__ stats) {
    extern __shared__ float2 lds2[];
    float2 *d = lds2, *tw = lds2 + L;
    __shared__ float red[4];
    const int64_t b = blockIdx.x;
    const float2 *p = pred + b * L;
    const bool spectrum = band != nullptr;   // block-uniform; without masks only the time-domain sums are formed
    if (spectrum) make_twiddles(tw, (int)L);
    float s_abs = 0.f, s_err = 0.f, s_ref = 0.f;
    for (int i = threadIdx.x; i < L; i += TPB) {
        const float2 v = p[i];
        if (spectrum) d[bitrev(i, log2n)] = v;
        if (target) {
            const float2 t = target[b * L + i];
            const float dx = v.x - t.x, dy = v.y - t.y;
            s_abs += fabsf(dx) + fabsf(dy);
            s_err += dx * dx + dy * dy;
            s_ref += t.x * t.x + t.y * t.y;
        }
    }
    float pm = 0.f, pl = 0.f, pu = 0.f;
    const float invL = 1.0f / (float)L;
    if (spectrum) fft_stages(d, tw, (int)L, log2n);
    for (int k = threadIdx.x; spectrum && k < L; k += TPB) {
        const float2 X = d[k];
        const float pw = (X.x * X.x + X.y * X.y) * invL;
        pm += band[k] ? pw : 0.f;
        pl += band[L + k] ? pw : 0.f;
        pu += band[2 * L + k] ? pw : 0.f;
    }
    float r0 = block_sum(s_abs, red), r1 = block_sum(s_err, red), r2 = block_sum(s_ref, red);
    float r3 = block_sum(pm, red), r4 = block_sum(pl, red), r5 = block_sum(pu, red);
    if (threadIdx.x == 0) {
        float *o = stats + b * 8;
        o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4; o[5] = r5; o[6] = 0.f; o[7] = 0.f;
    }
}

__global__ __launch_bounds__(TPB) void k_spectral_bwd(const float2 *__restrict__ pred,
                                                      const float2 *__restrict__ target, int64_t L, int log2n,
                                                      const uint8_t *__restrict__ band,
                                                      const float *__restrict__ gstats, float2 *__restrict__ dpred) {
    extern __shared__ float2 lds2[];
    float2 *d = lds2, *tw = lds2 + L;
    const int64_t b = blockIdx.x;
    const float2 *p = pred + b * L;
    const float *gs = gstats + b * 8;
    const float g_abs = gs[0], g_err = gs[1], g_m = gs[3], g_l = gs[4], g_u = gs[5];
    const bool need_fft = band != nullptr && ((g_m != 0.f) || (g_l != 0.f) || (g_u != 0.f));   // block-uniform
    if (need_fft) {
        make_twiddles(tw, (int)L);
        for (int i = threadIdx.x; i < L; i += TPB) d[bitrev(i, log2n)] = p[i];
        fft_stages(d, tw, (int)L, log2n);
        // Y = conj(w X), permuted to bit-reversed order for the second transform
        for (int k = threadIdx.x; k < L; k += TPB) {
            const float w = (band[k] ? g_m : 0.f) + (band[L + k] ? g_l : 0.f) + (band[2 * L + k] ? g_u : 0.f);
            const float2 X = d[k];
            d[k] = make_float2(w * X.x, -w * X.y);
        }
        __syncthreads();
        for (int k = threadIdx.x; k < L; k += TPB) {
            const unsigned r = bitrev(k, log2n);
            if ((unsigned)k < r) {
                const float2 a = d[k], c = d[r];
                d[k] = c;
                d[r] = a;
            }
        }
        fft_stages(d, tw, (int)L, log2n);
    }
    const float c2 = 2.0f / (float)L;
    for (int i = threadIdx.x; i < L; i += TPB) {
        float gx = 0.f, gy = 0.f;
        if (need_fft) {
            const float2 Z = d[i];   // IDFT_unnormalised(w X) = conj(Z)
            gx = c2 * Z.x;
            gy = -c2 * Z.y;
        }
        if (target) {
            const float2 v = p[i], t = target[b * L + i];
            const float dx = v.x - t.x, dy = v.y - t.y;
            gx += g_abs * (dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f)) + 2.f * g_err * dx;
            gy += g_abs * (dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f)) + 2.f * g_err * dy;
        }
        dpred[b * L + i] = make_float2(gx, gy);
    }
}

int check_len(int64_t L, int *log2n) {
    int l2 = 0;
    while ((1ll << l2) < L) ++l2;
    *log2n = l2;
    return (L >= 2 && L <= MAXL && (1ll << l2) == L) ? 1 : 0;
}

}  // namespace

extern "C" int dpd_spectral_stats(const float *pred, const float *target, int64_t B, int64_t L, const uint8_t *band,
                                  float *stats, void *stream) {
    DPD_REQUIRE(pred && stats && (band || target), DPD_ERR_INVALID_ARG, "spectral_stats: null pointer");
    DPD_REQUIRE(L >= 1, DPD_ERR_INVALID_ARG, "spectral_stats: empty rows");
    int log2n = 0;
    DPD_REQUIRE(!band || check_len(L, &log2n), DPD_ERR_INVALID_ARG,
                "spectral_stats: row length %lld is not a power of two in 2..4096", (long long)L);
    if (B <= 0) return DPD_OK;
    const size_t lds = band ? (size_t)(L + L / 2) * sizeof(float2) : 0;
    hipLaunchKernelGGL(k_spectral_stats, dim3((unsigned)B), dim3(TPB), lds, as_stream(stream),
                       reinterpret_cast<const float2 *>(pred), reinterpret_cast<const float2 *>(target), L, log2n,
                       band, stats);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_spectral_backward(const float *pred, const float *target, int64_t B, int64_t L,
                                     const uint8_t *band, const float *gstats, float *dpred, void *stream) {
    DPD_REQUIRE(pred && gstats && dpred && (band || target), DPD_ERR_INVALID_ARG, "spectral_backward: null pointer");
    DPD_REQUIRE(L >= 1, DPD_ERR_INVALID_ARG, "spectral_backward: empty rows");
    int log2n = 0;
    DPD_REQUIRE(!band || check_len(L, &log2n), DPD_ERR_INVALID_ARG,
                "spectral_backward: row length %lld is not a power of two in 2..4096", (long long)L);
    if (B <= 0) return DPD_OK;
    const size_t lds = band ? (size_t)(L + L / 2) * sizeof(float2) : 0;
    hipLaunchKernelGGL(k_spectral_bwd, dim3((unsigned)B), dim3(TPB), lds, as_stream(stream),
                       reinterpret_cast<const float2 *>(pred), reinterpret_cast<const float2 *>(target), L, log2n,
                       band, gstats, reinterpret_cast<float2 *>(dpred));
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}
