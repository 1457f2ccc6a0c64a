// Signal impairments on IQ rows: the training-time augmentation of the reference's notebook (training_colab.ipynb,
// data-loading cell, `augment_signal`: additive noise at a drawn SNR, phase offset, gain variation, thermal drift
// with AM-AM compression) and the noise model of the PA twin (PADigitalTwin.add_noise, models/pa_digital_twin.py:280-309:
// AWGN relative to the signal power plus per-sample phase noise).  Both are one streaming pass over the samples
// after a per-row power sum: 16 B/sample (8 in, 8 out), HBM-bound.  The normal deviates either come from the caller
// (a tensor of draws: results are then a pure function of the arguments) or are generated in the kernel from a
// counter-based generator (Philox-4x32-10 keyed by seed, counter = sample index; Box-Muller), which saves the
// 8 B/sample the draws would cost.
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int CHUNK = 4096;          // samples per partial sum of a row's power

// ---- counter-based normal deviates -----------------------------------------------------------------
DEV void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
DEV void philox4x32(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// two standard normal deviates from two 32-bit words (u in (0, 1]: the logarithm stays finite)
DEV void box_muller(unsigned a, unsigned b, float &z0, float &z1) {
    const float u = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f), v = (float)(b >> 8) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u));
    float s, c;
    sincosf(6.283185307179586f * v, &s, &c);
    z0 = r * c;
    z1 = r * s;
}
// the three deviates of sample n of stream `seed`: (I noise, Q noise, phase noise)
DEV void sample_normals(unsigned long long seed, int64_t n, float &zi, float &zq, float &zp) {
    unsigned c[4] = {(unsigned)n, (unsigned)((unsigned long long)n >> 32), 0u, 0u};
    philox4x32(c, (unsigned)seed, (unsigned)(seed >> 32));
    float unused;
    box_muller(c[0], c[1], zi, zq);
    box_muller(c[2], c[3], zp, unused);
}

// ---- per-row power: sum of squares of both components, in a fixed order ---------------------------------
__global__ __launch_bounds__(TPB) void k_row_power_partial(const float2 *__restrict__ x, int64_t S, int chunks,
                                                           double *__restrict__ part) {
    __shared__ double red[4];
    const int64_t row = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const int64_t lo = (int64_t)chunk * CHUNK, hi = lo + CHUNK < S ? lo + CHUNK : S;
    double acc = 0.0;
    for (int64_t s = lo + threadIdx.x; s < hi; s += TPB) {
        const float2 v = x[row * S + s];
        acc += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// power[row] = mean over the 2 S values of the row of v^2 (the reference's (x ** 2).mean())
__global__ __launch_bounds__(64) void k_row_power_final(const double *__restrict__ part, int64_t B, int chunks, int64_t S,
                                                        float *__restrict__ power) {
    for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
        double a = 0.0;
        for (int c = threadIdx.x; c < chunks; c += 64) a += part[row * chunks + c];
        a = wave_sum_d(a);
        if (threadIdx.x == 0) power[row] = (float)(a / (2.0 * (double)S));
    }
}

// ---- the impairment pass ----------------------------------------------------------------------------------
// p[8] per row: 0 noise_rel (noise variance = power * noise_rel), 1 noise_mul (0: no additive noise),
//               2, 3 cos / sin of the row's phase offset, 4 gain,
//               5 thermal gain shift (0: no thermal stage), 6 thermal phase shift (rad), 7 thermal compression
// PA_NOISE = false, the order of `augment_signal`: v = x + noise; rotate by the phase offset; gain; thermal stage
//            (gain_shift * (1 - 0.1 |v|^2 compression) and a rotation by the phase shift).
// PA_NOISE = true, the order of PADigitalTwin.add_noise: v = x rotated by pn_std * (phase deviate of the sample), plus noise.
template <bool PA_NOISE>
__global__ __launch_bounds__(TPB) void k_impair(const float2 *__restrict__ x, int64_t B, int64_t S, const float *__restrict__ params,
                                                const float *__restrict__ power, const float2 *__restrict__ noise,
                                                const float *__restrict__ pnoise, float pn_std, unsigned long long seed,
                                                float2 *__restrict__ out) {
    const int64_t n_all = B * S;
    for (int64_t n = (int64_t)blockIdx.x * TPB + threadIdx.x; n < n_all; n += (int64_t)gridDim.x * TPB) {
        const int64_t row = n / S;
        const float *p = params + row * 8;
        const float nmul = p[1];
        float2 v = x[n];
        float zi = 0.f, zq = 0.f, zp = 0.f;
        if (noise) {
            const float2 z = noise[n];
            zi = z.x; zq = z.y;
            if (PA_NOISE && pnoise) zp = pnoise[n];
        } else if (nmul != 0.f || (PA_NOISE && pn_std != 0.f)) {
            sample_normals(seed, n, zi, zq, zp);
        }
        const float nstd = nmul != 0.f ? sqrtf(power[row] * p[0]) * nmul : 0.f;
        if (PA_NOISE) {
            if (pn_std != 0.f) {
                float s, c;
                sincosf(zp * pn_std, &s, &c);
                v = make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
            }
            out[n] = make_float2(v.x + zi * nstd, v.y + zq * nstd);
            continue;
        }
        v = make_float2(v.x + zi * nstd, v.y + zq * nstd);
        v = make_float2(v.x * p[2] - v.y * p[3], v.x * p[3] + v.y * p[2]);
        v = make_float2(v.x * p[4], v.y * p[4]);
        if (p[5] != 0.f) {
            const float mag = sqrtf(v.x * v.x + v.y * v.y);
            const float g = p[5] * (1.0f - 0.1f * (mag * mag) * p[7]);
            float s, c;
            sincosf(p[6], &s, &c);
            v = make_float2(g * (v.x * c - v.y * s), g * (v.x * s + v.y * c));
        }
        out[n] = v;
    }
}

int row_power(const float *x, int64_t B, int64_t S, float *power, double *part, hipStream_t st) {
    const int chunks = (int)((S + CHUNK - 1) / CHUNK);
    hipLaunchKernelGGL(k_row_power_partial, dim3((unsigned)(B * chunks)), dim3(TPB), 0, st, reinterpret_cast<const float2 *>(x), S, chunks, part);
    DPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_row_power_final, dim3(grid_for(B, 1, 2048)), dim3(64), 0, st, part, B, chunks, S, power);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

}  // namespace

extern "C" size_t dpd_impair_workspace_bytes(int64_t B, int64_t S) {
    if (B < 0 || S < 1) return 0;
    const int64_t chunks = (S + CHUNK - 1) / CHUNK;
    return (size_t)(B * chunks) * sizeof(double) + (size_t)B * sizeof(float) + 256;
}

static int impair_common(bool pa_noise, const float *x, int64_t B, int64_t S, const float *params, const float *noise,
                         const float *pnoise, float pn_std, unsigned long long seed, float *out, void *ws, size_t ws_bytes,
                         hipStream_t st) {
    DPD_REQUIRE(B >= 0 && S >= 1, DPD_ERR_INVALID_ARG, "impair: bad shape B=%lld S=%lld", (long long)B, (long long)S);
    if (B == 0) return DPD_OK;
    DPD_REQUIRE(x && params && out, DPD_ERR_INVALID_ARG, "impair: null pointer");
    DPD_REQUIRE(B * ((S + CHUNK - 1) / CHUNK) < ((int64_t)1 << 31), DPD_ERR_INVALID_ARG, "impair: too many rows for one launch");
    DPD_REQUIRE(ws && ws_bytes >= dpd_impair_workspace_bytes(B, S), DPD_ERR_WORKSPACE,
                "impair: workspace of %zu bytes needed, %zu given", dpd_impair_workspace_bytes(B, S), ws_bytes);
    const int64_t chunks = (S + CHUNK - 1) / CHUNK;
    double *part = reinterpret_cast<double *>(ws);
    float *power = reinterpret_cast<float *>(part + B * chunks);
    int rc = row_power(x, B, S, power, part, st);
    if (rc) return rc;
    const int grid = grid_for(B * S, TPB, 256 * 16);
    if (pa_noise)
        hipLaunchKernelGGL(k_impair<true>, dim3(grid), dim3(TPB), 0, st, reinterpret_cast<const float2 *>(x), B, S, params, power,
                           reinterpret_cast<const float2 *>(noise), pnoise, pn_std, seed, reinterpret_cast<float2 *>(out));
    else
        hipLaunchKernelGGL(k_impair<false>, dim3(grid), dim3(TPB), 0, st, reinterpret_cast<const float2 *>(x), B, S, params, power,
                           reinterpret_cast<const float2 *>(noise), pnoise, pn_std, seed, reinterpret_cast<float2 *>(out));
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_augment_rows(const float *x, int64_t B, int64_t S, const float *params, const float *noise,
                                uint64_t seed, float *out, void *ws, size_t ws_bytes, void *stream) {
    return impair_common(false, x, B, S, params, noise, nullptr, 0.f, seed, out, ws, ws_bytes, as_stream(stream));
}

extern "C" int dpd_pa_add_noise(const float *x, int64_t B, int64_t S, const float *params, float phase_noise_rad,
                                const float *noise, const float *phase_noise, uint64_t seed, float *out, void *ws,
                                size_t ws_bytes, void *stream) {
    DPD_REQUIRE(!noise || phase_noise || phase_noise_rad == 0.f, DPD_ERR_INVALID_ARG,
                "pa_add_noise: explicit noise draws need explicit phase-noise draws as well");
    return impair_common(true, x, B, S, params, noise, phase_noise, phase_noise_rad, seed, out, ws, ws_bytes, as_stream(stream));
}
