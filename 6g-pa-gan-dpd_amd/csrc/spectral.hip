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

// Row lengths that are not a power of two (the reference's torch.fft.fft takes any length, e.g. the 4,087
// PA outputs of a 4,096-sample validation slice): the band powers from a direct DFT, O(L^2) per row, with
// the L twiddles exp(-2 pi i r / L) tabulated in LDS and the phase index r = k n mod L advanced exactly in
// integers.  One workgroup per row; a metrics path (forward only), not a training path.
__global__ __launch_bounds__(TPB) void k_spectral_stats_dft(const float2 *__restrict__ pred,
                                                            const float2 *__restrict__ target, int64_t L,
                                                            const uint8_t *__restrict__ band, float *__restrict__ stats) {
    extern __shared__ float2 lds2[];
    float2 *tw = lds2;
    __shared__ float red[4];
    const int64_t b = blockIdx.x;
    const float2 *p = pred + b * L;
    const int n_len = (int)L;
    for (int r = threadIdx.x; r < n_len; r += TPB) {
        float sn, cs;
        sincospif(2.0f * (float)r / (float)n_len, &sn, &cs);
        tw[r] = make_float2(cs, -sn);
    }
    float s_abs = 0.f, s_err = 0.f, s_ref = 0.f;
    for (int i = threadIdx.x; target && i < n_len; i += TPB) {
        const float2 v = p[i], t = target[b * L + i];
        const float dx = v.x - t.x, dy = v.y - t.y;
        s_abs += fabsf(dx) + fabsf(dy);
        s_err += dx * dx + dy * dy;
        s_ref += t.x * t.x + t.y * t.y;
    }
    __syncthreads();
    float pm = 0.f, pl = 0.f, pu = 0.f;
    const float invL = 1.0f / (float)L;
    for (int k = threadIdx.x; k < n_len; k += TPB) {
        float xr = 0.f, xi = 0.f;
        int r = 0;
        for (int n = 0; n < n_len; ++n) {
            const float2 v = p[n], w = tw[r];
            xr += v.x * w.x - v.y * w.y;
            xi += v.x * w.y + v.y * w.x;
            r += k;
            if (r >= n_len) r -= n_len;
        }
        const float pw = (xr * xr + xi * xi) * invL;
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

// The gradient for row lengths that are not a power of two: both transforms as direct sums over the tabulated
// twiddles (the reference differentiates torch.fft.fft of any length, utils/spectral_loss.py:99-103).  The weighted
// spectrum Y[k] = w_k X[k] is parked in LDS next to the twiddles (2 L complex words); the row itself is read from
// global memory by every thread at the same address (a broadcast).  O(L^2) per row and workgroup.
__global__ __launch_bounds__(TPB) void k_spectral_bwd_dft(const float2 *__restrict__ pred, const float2 *__restrict__ target,
                                                          int64_t L, const uint8_t *__restrict__ band,
                                                          const float *__restrict__ gstats, float2 *__restrict__ dpred) {
    extern __shared__ float2 lds2[];
    float2 *tw = lds2, *Y = lds2 + L;
    const int64_t b = blockIdx.x;
    const float2 *p = pred + b * L;
    const float *gs = gstats + b * 8;
    const float g_abs = gs[0], g_err = gs[1], g_m = gs[3], g_l = gs[4], g_u = gs[5];
    const bool need_dft = band != nullptr && ((g_m != 0.f) || (g_l != 0.f) || (g_u != 0.f));   // block-uniform
    const int n_len = (int)L;
    if (need_dft) {
        for (int r = threadIdx.x; r < n_len; r += TPB) {
            float sn, cs;
            sincospif(2.0f * (float)r / (float)n_len, &sn, &cs);
            tw[r] = make_float2(cs, -sn);
        }
        __syncthreads();
        for (int k = threadIdx.x; k < n_len; k += TPB) {
            const float w = (band[k] ? g_m : 0.f) + (band[L + k] ? g_l : 0.f) + (band[2 * L + k] ? g_u : 0.f);
            float xr = 0.f, xi = 0.f;
            if (w != 0.f) {
                int r = 0;
                for (int n = 0; n < n_len; ++n) {
                    const float2 v = p[n], t = tw[r];
                    xr += v.x * t.x - v.y * t.y;
                    xi += v.x * t.y + v.y * t.x;
                    r += k;
                    if (r >= n_len) r -= n_len;
                }
            }
            Y[k] = make_float2(w * xr, w * xi);
        }
        __syncthreads();
    }
    const float c2 = 2.0f / (float)L;
    for (int i = threadIdx.x; i < n_len; i += TPB) {
        float gx = 0.f, gy = 0.f;
        if (need_dft) {
            float zr = 0.f, zi = 0.f;     // Z[i] = sum_k Y[k] exp(+2 pi i k i / L)
            int r = 0;
            for (int k = 0; k < n_len; ++k) {
                const float2 y = Y[k], t = tw[r];
                zr += y.x * t.x + y.y * t.y;
                zi += y.y * t.x - y.x * t.y;
                r += i;
                if (r >= n_len) r -= n_len;
            }
            gx = c2 * zr;
            gy = c2 * zi;
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


// ------------------------------------------------------------------ L = 1024: one wave per row, FFT in registers
// 1024 = 16 x 16 x 4.  Lane l holds x[64 m + l], m = 0..15 (coalesced loads).  Forward transform:
//   S1 16-point DFT over m -> A[k1];  T1 A[k1] *= W1024^(l k1);
//   P1 through LDS: lane (k1', a') = (l >> 2, l & 3) takes A_{4p + a'}[k1'], p = 0..15;
//   S2 16-point DFT over p -> C[q];   T2 C[q] *= W64^(a' q);
//   P2 through LDS: lane (k1', u) takes C_{a}[4 qi + u] for qi, a = 0..3;
//   S3 4-point DFT over a -> X[k1' + 16 (4 qi + u) + 256 r] in register 4 qi + r.
// The DFT matrix is symmetric, so running the same stages in reverse order (S3, P2^-1, T2, S2, P1^-1, T1, S1)
// transforms a vector held in the output layout back into the input layout: that is the second transform of
// the backward pass, and its result lands where the coalesced store wants it.  All LDS traffic is wave-private.
namespace w1024 {

constexpr int ROWS_PER_BLOCK = 4;          // waves
constexpr int PITCH = 68;                  // complex words per k1 row of the transposition buffer (136 = 8 mod 32 banks)
constexpr int BUF = 16 * PITCH;

struct cpx {
    float x, y;
};
DEV cpx operator+(cpx a, cpx b) { return {a.x + b.x, a.y + b.y}; }
DEV cpx operator-(cpx a, cpx b) { return {a.x - b.x, a.y - b.y}; }
DEV cpx cmul(cpx a, cpx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
DEV cpx mul_mi(cpx a) { return {a.y, -a.x}; }     // a * (-i)

DEV void fft4(cpx &v0, cpx &v1, cpx &v2, cpx &v3) {
    const cpx t0 = v0 + v2, t1 = v0 - v2, t2 = v1 + v3, t3 = mul_mi(v1 - v3);
    v0 = t0 + t2;
    v1 = t1 + t3;
    v2 = t0 - t2;
    v3 = t1 - t3;
}

// natural-order 16-point DFT of v[0..15] (index m = 4a + b -> k = c + 4d)
DEV void fft16(cpx (&v)[16]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) fft4(v[b], v[4 + b], v[8 + b], v[12 + b]);      // v[4c + b] = y[b][c]
    const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    const cpx w1 = {c1, -s1}, w2 = {h, -h}, w3 = {s1, -c1}, w6 = {-h, -h}, w9 = {-c1, s1};
    v[4 * 1 + 1] = cmul(v[4 * 1 + 1], w1);
    v[4 * 2 + 1] = cmul(v[4 * 2 + 1], w2);
    v[4 * 3 + 1] = cmul(v[4 * 3 + 1], w3);
    v[4 * 1 + 2] = cmul(v[4 * 1 + 2], w2);
    v[4 * 2 + 2] = mul_mi(v[4 * 2 + 2]);
    v[4 * 3 + 2] = cmul(v[4 * 3 + 2], w6);
    v[4 * 1 + 3] = cmul(v[4 * 1 + 3], w3);
    v[4 * 2 + 3] = cmul(v[4 * 2 + 3], w6);
    v[4 * 3 + 3] = cmul(v[4 * 3 + 3], w9);
    cpx o[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cpx a0 = v[4 * c + 0], a1 = v[4 * c + 1], a2 = v[4 * c + 2], a3 = v[4 * c + 3];
        fft4(a0, a1, a2, a3);
        o[c] = a0;
        o[c + 4] = a1;
        o[c + 8] = a2;
        o[c + 12] = a3;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = o[k];
}

// Twiddle tables in LDS, shared by the block: t1[k][l] = W1024^(l k) (read by lane l: conflict-free),
// t2[q][a] = W64^(a q).  Keeping them out of the register file is worth a resident wave per SIMD.
struct Twiddles {
    const float2 *t1, *t2;
};
// Residue (sample index mod 64) a lane holds after the loads.  8-byte loads: lane l holds x[64 m + l].  16-byte loads
// (WIDE): lane l loads the sample pairs (128 m' + 2 l, + 1); one half-wave exchange per register then leaves lanes
// 0..31 with residue 2 l (all sixteen m) and lanes 32..63 with residue 2 (l - 32) + 1 -- see exchange_halves.
template <bool WIDE>
DEV int residue_of(int lane) { return WIDE ? 2 * (lane & 31) + (lane >> 5) : lane; }
template <bool WIDE>
DEV void make_tw(float2 *t1, float2 *t2) {
    for (int i = threadIdx.x; i < 1024; i += ROWS_PER_BLOCK * 64) {
        float s, c;
        sincospif(-2.0f * (float)(residue_of<WIDE>(i & 63) * (i >> 6)) / 1024.0f, &s, &c);
        t1[i] = make_float2(c, s);
    }
    for (int i = threadIdx.x; i < 64; i += ROWS_PER_BLOCK * 64) {
        float s, c;
        sincospif(-2.0f * (float)((i & 3) * (i >> 2)) / 64.0f, &s, &c);
        t2[i] = make_float2(c, s);
    }
    __syncthreads();
}
// WIDE loads: registers A (first sample of the pair) and B (second) of the two wave halves -> (even m, odd m) of this
// lane's residue.  v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second:
// lanes 0..31 keep A and receive the upper half's A (same residue 2 l, next m); lanes 32..63 receive the lower half's
// B and keep their own B (residue 2 l + 1).  The same call applied to (even m, odd m) is the way back.
DEV void exchange_halves(float &a, float &b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    unsigned u0 = r[0], u1 = r[1];
    asm volatile("" : "+v"(u0), "+v"(u1));   // keeps the two results of the swap apart (the compiler folds them otherwise)
    a = __builtin_bit_cast(float, u0);
    b = __builtin_bit_cast(float, u1);
}
DEV cpx tw1(const Twiddles &t, int k, int lane) {
    const float2 w = t.t1[64 * k + lane];
    return {w.x, w.y};
}
DEV cpx tw2(const Twiddles &t, int q, int lane) {
    const float2 w = t.t2[4 * q + (lane & 3)];
    return {w.x, w.y};
}

// col: the residue this lane holds (residue_of; the twiddle table t1 is already in lane order)
DEV void forward(cpx (&v)[16], const Twiddles &tw, float2 *buf, int lane, int col) {
    const int k1 = lane >> 2, a = lane & 3;
    fft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw1(tw, k, lane));
#pragma unroll
    for (int k = 0; k < 16; ++k) buf[k * PITCH + col] = make_float2(v[k].x, v[k].y);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const float2 t = buf[k1 * PITCH + 4 * p + a];
        v[p] = {t.x, t.y};
    }
    __builtin_amdgcn_wave_barrier();
    fft16(v);
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], tw2(tw, q, lane));
#pragma unroll
    for (int q = 0; q < 16; ++q) buf[k1 * PITCH + 4 * q + a] = make_float2(v[q].x, v[q].y);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            const float2 t = buf[k1 * PITCH + 4 * (4 * qi + a) + aa];   // u == a: this lane's residue of q
            v[4 * qi + aa] = {t.x, t.y};
        }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) fft4(v[4 * qi], v[4 * qi + 1], v[4 * qi + 2], v[4 * qi + 3]);
}

// the same stages in reverse order: output layout -> input layout
DEV void transposed(cpx (&v)[16], const Twiddles &tw, float2 *buf, int lane, int col) {
    const int k1 = lane >> 2, a = lane & 3;
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) fft4(v[4 * qi], v[4 * qi + 1], v[4 * qi + 2], v[4 * qi + 3]);
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) buf[k1 * PITCH + 4 * (4 * qi + a) + aa] = make_float2(v[4 * qi + aa].x, v[4 * qi + aa].y);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float2 t = buf[k1 * PITCH + 4 * q + a];
        v[q] = {t.x, t.y};
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], tw2(tw, q, lane));
    fft16(v);
#pragma unroll
    for (int p = 0; p < 16; ++p) buf[k1 * PITCH + 4 * p + a] = make_float2(v[p].x, v[p].y);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float2 t = buf[k * PITCH + col];
        v[k] = {t.x, t.y};
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw1(tw, k, lane));
    fft16(v);
}

// bit (4 qi + r) of word b: band b contains the bin this lane holds in register 4 qi + r
DEV void band_bits(unsigned (&bits)[3], const uint8_t *__restrict__ band, int lane) {
    const int k1 = lane >> 2, u = lane & 3;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        bits[b] = 0u;
#pragma unroll
        for (int qi = 0; qi < 4; ++qi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k1 + 16 * (4 * qi + u) + 256 * r;
                bits[b] |= (band[b * 1024 + k] ? 1u : 0u) << (4 * qi + r);
            }
    }
}

// Waits for every load issued so far, once, in front of a loop that prefetches the next row: the compiler sets the wait at
// the loop's head from BOTH ways into it, and on the way in from the prologue the prefetched loads are the youngest
// operations, so without this it would wait there, in every iteration, for the iteration's own stores as well
// (vmcnt counts stores, and a store's acknowledgement takes a memory round trip).  s_waitcnt vmcnt(0) alone.
#define DRAIN_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)

// A row as the loads deliver it: 16 sample slots per lane.  8-byte loads: slot m = x[64 m + lane].  16-byte loads (WIDE,
// rows 16-byte aligned): slots 2 m', 2 m' + 1 = the pair x[128 m' + 2 lane], x[128 m' + 2 lane + 1] -- half the memory
// instructions for the same bytes.  Everything that works sample by sample (error sums, the sign / difference terms of
// the gradient, the stores) uses the slots as they are; only the transform needs the residue layout (to_residues).
template <bool WIDE>
struct RowIO {
    static DEV void load(float2 (&r)[16], const float2 *__restrict__ row, int lane, bool live) {
        if (WIDE) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float4 q = live ? reinterpret_cast<const float4 *>(row)[64 * m + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
                r[2 * m] = make_float2(q.x, q.y);
                r[2 * m + 1] = make_float2(q.z, q.w);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) r[m] = live ? row[64 * m + lane] : make_float2(0.f, 0.f);
        }
    }
    static DEV void store(float2 *__restrict__ row, const float2 (&r)[16], int lane) {
        if (WIDE) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
                reinterpret_cast<float4 *>(row)[64 * m + lane] = make_float4(r[2 * m].x, r[2 * m].y, r[2 * m + 1].x, r[2 * m + 1].y);
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) row[64 * m + lane] = r[m];
        }
    }
    // slots -> v[m] = x[64 m + residue_of(lane)], and back (the exchange is its own inverse)
    static DEV void to_residues(cpx (&v)[16]) {
        if (!WIDE) return;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            exchange_halves(v[2 * m].x, v[2 * m + 1].x);
            exchange_halves(v[2 * m].y, v[2 * m + 1].y);
        }
    }
};

// TGT: a target tensor is given (a template flag, not a test of the pointer inside the loop: loads inside a conditional
// block make the compiler wait for every load in flight at the block's end, the prefetched row included)
template <bool WIDE, bool TGT>
__global__ __launch_bounds__(ROWS_PER_BLOCK * 64) void k_stats(const float2 *__restrict__ pred, const float2 *__restrict__ target,
                                                              int64_t B, const uint8_t *__restrict__ band,
                                                              float *__restrict__ stats) {
    __shared__ float2 lds[ROWS_PER_BLOCK][BUF];
    __shared__ float2 s_t1[1024], s_t2[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = residue_of<WIDE>(lane);
    make_tw<WIDE>(s_t1, s_t2);
    const Twiddles tw{s_t1, s_t2};
    unsigned bits[3];
    band_bits(bits, band, lane);
    // the next row's samples are loaded while the current row is transformed
    const int64_t stride = (int64_t)gridDim.x * ROWS_PER_BLOCK;
    float2 nx[16], nt[16];
    auto fetch = [&](int64_t b) {          // (index clamped, no branch around the loads; rows past the end are not used)
        const int64_t bc = b < B ? b : B - 1;
        RowIO<WIDE>::load(nx, pred + bc * 1024, lane, true);
        if (TGT) RowIO<WIDE>::load(nt, target + bc * 1024, lane, true);
    };
    fetch((int64_t)blockIdx.x * ROWS_PER_BLOCK + wave);
    DRAIN_VMEM();
    for (int64_t b = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave; b < B; b += stride) {
        cpx v[16];
        float s_abs = 0.f, s_err = 0.f, s_ref = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 x = nx[m];
            v[m] = {x.x, x.y};
            if (TGT) {
                const float2 t = nt[m];
                const float dx = x.x - t.x, dy = x.y - t.y;
                s_abs += fabsf(dx) + fabsf(dy);
                s_err += dx * dx + dy * dy;
                s_ref += t.x * t.x + t.y * t.y;
            }
        }
        fetch(b + stride);
#ifndef ABL_NO_FFT      // experiment (timing only): the row transform left out
        RowIO<WIDE>::to_residues(v);
        forward(v, tw, lds[wave], lane, col);
#endif
        float pm = 0.f, pl = 0.f, pu = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pw = (v[i].x * v[i].x + v[i].y * v[i].y) * (1.0f / 1024.0f);
            pm += ((bits[0] >> i) & 1u) ? pw : 0.f;
            pl += ((bits[1] >> i) & 1u) ? pw : 0.f;
            pu += ((bits[2] >> i) & 1u) ? pw : 0.f;
        }
        s_abs = wave_sum(s_abs);
        s_err = wave_sum(s_err);
        s_ref = wave_sum(s_ref);
        pm = wave_sum(pm);
        pl = wave_sum(pl);
        pu = wave_sum(pu);
        if (lane == 0) {
            float *o = stats + b * 8;
            o[0] = s_abs; o[1] = s_err; o[2] = s_ref; o[3] = pm; o[4] = pl; o[5] = pu; o[6] = 0.f; o[7] = 0.f;
        }
    }
}

template <bool WIDE, bool TGT>
__global__ __launch_bounds__(ROWS_PER_BLOCK * 64, 2) void k_bwd(const float2 *__restrict__ pred, const float2 *__restrict__ target,
                                                            int64_t B, const uint8_t *__restrict__ band,
                                                            const float *__restrict__ gstats, float2 *__restrict__ dpred) {
    __shared__ float2 lds[ROWS_PER_BLOCK][BUF];
    __shared__ float2 s_t1[1024], s_t2[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = residue_of<WIDE>(lane);
    make_tw<WIDE>(s_t1, s_t2);
    const Twiddles tw{s_t1, s_t2};
    unsigned bits[3];
    band_bits(bits, band, lane);
    // Two resident waves per SIMD (register budget 256): the row's target and the NEXT row's samples and gradient
    // coefficients are requested before the transforms, so their latency lies under the arithmetic of this row (the
    // wait at the top of an iteration then covers only what was issued a whole row ago, the stores included).
    const int64_t stride = (int64_t)gridDim.x * ROWS_PER_BLOCK;
    const int64_t b0 = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave;
    float2 nx[16];
    float ng[6];
    auto fetch_g = [&](int64_t b) {
        const float *gs = gstats + b * 8;
#pragma unroll
        for (int k = 0; k < 6; ++k) ng[k] = gs[k];
    };
    {
        const int64_t bc = b0 < B ? b0 : B - 1;            // (the launcher never passes an empty batch)
        RowIO<WIDE>::load(nx, pred + bc * 1024, lane, true);
        fetch_g(bc);
    }
    DRAIN_VMEM();
    for (int64_t b = b0; b < B; b += stride) {
        const float g_abs = ng[0], g_err = ng[1], g_m = ng[3], g_l = ng[4], g_u = ng[5];
        const bool need_fft = (g_m != 0.f) || (g_l != 0.f) || (g_u != 0.f);   // wave-uniform
        float2 x[16], tg[16];
        cpx v[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            x[m] = nx[m];
            v[m] = {x[m].x, x[m].y};
        }
        if (TGT) RowIO<WIDE>::load(tg, target + b * 1024, lane, true);
        {
            const int64_t bn = b + stride < B ? b + stride : b;                     // (clamped: no branch around the loads)
            RowIO<WIDE>::load(nx, pred + bn * 1024, lane, true);
            fetch_g(bn);
        }
#ifdef ABL_NO_FFT
        if (false) {
#else
        if (need_fft) {
#endif
            RowIO<WIDE>::to_residues(v);
            forward(v, tw, lds[wave], lane, col);
            // Y = conj(w X), then the transposed transform: IDFT_unnormalised(w X) = conj(F conj(w X))
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float w = (((bits[0] >> i) & 1u) ? g_m : 0.f) + (((bits[1] >> i) & 1u) ? g_l : 0.f) +
                                (((bits[2] >> i) & 1u) ? g_u : 0.f);
                v[i] = {w * v[i].x, -w * v[i].y};
            }
            transposed(v, tw, lds[wave], lane, col);
            RowIO<WIDE>::to_residues(v);          // back to the slots of the loads
        }
        const float c2 = 2.0f / 1024.0f;
        float2 g[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            float gx = 0.f, gy = 0.f;
            if (need_fft) {
                gx = c2 * v[m].x;
                gy = -c2 * v[m].y;
            }
            if (TGT) {
                const float dx = x[m].x - tg[m].x, dy = x[m].y - tg[m].y;
                gx += g_abs * (dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f)) + 2.f * g_err * dx;
                gy += g_abs * (dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f)) + 2.f * g_err * dy;
            }
            g[m] = make_float2(gx, gy);
        }
        RowIO<WIDE>::store(dpred + b * 1024, g, lane);
    }
}

#ifndef SPECTRAL_WIDE
#define SPECTRAL_WIDE 1            // experiment switch: 0 = 8-byte loads for every tensor
#endif
bool wide_ok(const void *a, const void *b, const void *c) {
    const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
    return SPECTRAL_WIDE && (bits & 15) == 0;
}
// resident blocks per CU: three by LDS (43 KB per block) for k_stats, two by registers for k_bwd
int grid_for_rows(int64_t B, int per_cu = 3) {
    int64_t blocks = (B + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace w1024


// ------------------------------------------------------------------ L = 256: four rows per wave
// 256 = 16 x 16.  A row occupies 16 lanes (lane group r = lane >> 4 of the wave, l = lane & 15), lane l
// holds x[16 m + l]: 16-point DFT over m -> A[k1]; A[k1] *= W256^(l k1); one transposition through LDS
// (lane k1' takes A_p[k1'], p = 0..15); 16-point DFT over p -> X[k1' + 16 k2] in register k2.  The
// reverse order of the same stages is the second transform of the backward, as for 1,024 points.
namespace w256 {

using w1024::cpx;
using w1024::cmul;
using w1024::fft16;
constexpr int WAVES = 4, ROWS_PER_WAVE = 4, ROWS_PER_BLOCK = WAVES * ROWS_PER_WAVE;
constexpr int PITCH = 17;                         // complex words per k1 row (34 words: 16 lanes hit 16 bank pairs)
constexpr int ROW_BUF = 16 * PITCH, BUF = ROWS_PER_WAVE * ROW_BUF;

DEV void make_tw(float2 *t) {                     // t[k][l] = W256^(l k)
    for (int i = threadIdx.x; i < 256; i += WAVES * 64) {
        float s, c;
        sincospif(-2.0f * (float)((i & 15) * (i >> 4)) / 256.0f, &s, &c);
        t[i] = make_float2(c, s);
    }
    __syncthreads();
}

DEV void forward(cpx (&v)[16], const float2 *tw, float2 *buf, int l) {
    fft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const float2 w = tw[16 * k + l];
        v[k] = cmul(v[k], cpx{w.x, w.y});
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) buf[k * PITCH + l] = make_float2(v[k].x, v[k].y);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const float2 t = buf[l * PITCH + p];
        v[p] = {t.x, t.y};
    }
    __builtin_amdgcn_wave_barrier();
    fft16(v);
}

DEV void transposed(cpx (&v)[16], const float2 *tw, float2 *buf, int l) {
    fft16(v);
#pragma unroll
    for (int p = 0; p < 16; ++p) buf[l * PITCH + p] = make_float2(v[p].x, v[p].y);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float2 t = buf[k * PITCH + l];
        v[k] = {t.x, t.y};
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const float2 w = tw[16 * k + l];
        v[k] = cmul(v[k], cpx{w.x, w.y});
    }
    fft16(v);
}

DEV float sum16(float v) {     // over the 16 lanes of a row
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

DEV void band_bits(unsigned (&bits)[3], const uint8_t *__restrict__ band, int l) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        bits[b] = 0u;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) bits[b] |= (band[b * 256 + l + 16 * k2] ? 1u : 0u) << k2;
    }
}

__global__ __launch_bounds__(WAVES * 64) void k_stats(const float2 *__restrict__ pred, const float2 *__restrict__ target,
                                                     int64_t B, const uint8_t *__restrict__ band, float *__restrict__ stats) {
    __shared__ float2 lds[WAVES][BUF];
    __shared__ float2 s_tw[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane >> 4, l = lane & 15;
    make_tw(s_tw);
    unsigned bits[3];
    band_bits(bits, band, l);
    float2 *buf = lds[wave] + r * ROW_BUF;
    const int64_t n_groups = (B + ROWS_PER_WAVE - 1) / ROWS_PER_WAVE;
    for (int64_t grp = (int64_t)blockIdx.x * WAVES + wave; grp < n_groups; grp += (int64_t)gridDim.x * WAVES) {
        const int64_t b = grp * ROWS_PER_WAVE + r;
        const bool live = b < B;
        cpx v[16];
        float s_abs = 0.f, s_err = 0.f, s_ref = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 x = live ? pred[b * 256 + 16 * m + l] : make_float2(0.f, 0.f);
            v[m] = {x.x, x.y};
            if (target && live) {
                const float2 t = target[b * 256 + 16 * m + l];
                const float dx = x.x - t.x, dy = x.y - t.y;
                s_abs += fabsf(dx) + fabsf(dy);
                s_err += dx * dx + dy * dy;
                s_ref += t.x * t.x + t.y * t.y;
            }
        }
        forward(v, s_tw, buf, l);
        float pm = 0.f, pl = 0.f, pu = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pw = (v[i].x * v[i].x + v[i].y * v[i].y) * (1.0f / 256.0f);
            pm += ((bits[0] >> i) & 1u) ? pw : 0.f;
            pl += ((bits[1] >> i) & 1u) ? pw : 0.f;
            pu += ((bits[2] >> i) & 1u) ? pw : 0.f;
        }
        s_abs = sum16(s_abs); s_err = sum16(s_err); s_ref = sum16(s_ref);
        pm = sum16(pm); pl = sum16(pl); pu = sum16(pu);
        if (l == 0 && live) {
            float *o = stats + b * 8;
            o[0] = s_abs; o[1] = s_err; o[2] = s_ref; o[3] = pm; o[4] = pl; o[5] = pu; o[6] = 0.f; o[7] = 0.f;
        }
    }
}

__global__ __launch_bounds__(WAVES * 64) void k_bwd(const float2 *__restrict__ pred, const float2 *__restrict__ target, int64_t B,
                                                   const uint8_t *__restrict__ band, const float *__restrict__ gstats,
                                                   float2 *__restrict__ dpred) {
    __shared__ float2 lds[WAVES][BUF];
    __shared__ float2 s_tw[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane >> 4, l = lane & 15;
    make_tw(s_tw);
    unsigned bits[3];
    band_bits(bits, band, l);
    float2 *buf = lds[wave] + r * ROW_BUF;
    const int64_t n_groups = (B + ROWS_PER_WAVE - 1) / ROWS_PER_WAVE;
    for (int64_t grp = (int64_t)blockIdx.x * WAVES + wave; grp < n_groups; grp += (int64_t)gridDim.x * WAVES) {
        const int64_t b = grp * ROWS_PER_WAVE + r;
        const bool live = b < B;
        float g_abs = 0.f, g_err = 0.f, g_m = 0.f, g_l = 0.f, g_u = 0.f;
        if (live) {
            const float *gs = gstats + b * 8;
            g_abs = gs[0]; g_err = gs[1]; g_m = gs[3]; g_l = gs[4]; g_u = gs[5];
        }
        cpx x[16], v[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 t = live ? pred[b * 256 + 16 * m + l] : make_float2(0.f, 0.f);
            x[m] = {t.x, t.y};
            v[m] = x[m];
        }
        // the four rows of a wave share the instruction stream: the transforms always run (zero weights give zero)
        forward(v, s_tw, buf, l);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float w = (((bits[0] >> i) & 1u) ? g_m : 0.f) + (((bits[1] >> i) & 1u) ? g_l : 0.f) +
                            (((bits[2] >> i) & 1u) ? g_u : 0.f);
            v[i] = {w * v[i].x, -w * v[i].y};
        }
        transposed(v, s_tw, buf, l);
        const float c2 = 2.0f / 256.0f;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            float gx = c2 * v[m].x, gy = -c2 * v[m].y;
            if (target && live) {
                const float2 t = target[b * 256 + 16 * m + l];
                const float dx = x[m].x - t.x, dy = x[m].y - t.y;
                gx += g_abs * (dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f)) + 2.f * g_err * dx;
                gy += g_abs * (dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f)) + 2.f * g_err * dy;
            }
            if (live) dpred[b * 256 + 16 * m + l] = make_float2(gx, gy);
        }
    }
}

int grid_for_rows(int64_t B) {
    int64_t blocks = (B + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    if (blocks > 256 * 4) blocks = 256 * 4;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace w256

int check_len(int64_t L, int *log2n) {
    int l2 = 0;
    while ((1ll << l2) < L) ++l2;
    *log2n = l2;
    return (L >= 2 && L <= MAXL && (1ll << l2) == L) ? 1 : 0;
}

}  // namespace

// ------------------------------------------------------------------ loss arithmetic on the row statistics
// SpectralLoss.forward (utils/spectral_loss.py:209-257) after the per-row statistics, in three tiny kernels so
// that a data-parallel caller can all-reduce the five sums in between:
//   rowdb:  per row (EVM_dB, ACPR_lower_dB, ACPR_upper_dB, NMSE_dB); sums = (sum|p-t|, sum EVM_dB, sum lower,
//           sum upper, sum NMSE_dB, rows) over the rows, in a fixed order (one workgroup, fp64 partials)
//   final:  total = w_l1 L1 + w_evm clamp((mean EVM_dB + 40)/40, 0, 2) + w_acpr clamp((max(mean lo, mean up) + 50)/50, 0, 2)
//           and the derivative of total w.r.t. the first four sums
//   terms_backward: d total / d statistics of every row (chain rule through the dB conversions)
namespace terms {

constexpr float LN10 = 2.302585092994046f;

struct RowDb {
    float evm, lo, up, nmse;
};
DEV RowDb row_db(const float *st, float n) {
    RowDb r;
    const float ref_power = fmaxf(st[2] / n, 1e-10f);
    const float evm = sqrtf((st[1] / n) / ref_power);
    r.evm = 20.f * log10f(evm + 1e-10f);
    const float mainp = fmaxf(st[3], 1e-10f);
    r.lo = 10.f * log10f(st[4] / mainp + 1e-10f);
    r.up = 10.f * log10f(st[5] / mainp + 1e-10f);
    r.nmse = 10.f * log10f(st[1] / fmaxf(st[2], 1e-10f) + 1e-10f);
    return r;
}

__global__ __launch_bounds__(1024) void k_rowdb(const float *__restrict__ stats, int64_t B, float n,
                                                float *__restrict__ rowdb, float *__restrict__ sums) {
    __shared__ double red[16][5];
    double a[5] = {0, 0, 0, 0, 0};
    for (int64_t b = threadIdx.x; b < B; b += 1024) {
        const float *st = stats + b * 8;
        const RowDb r = row_db(st, n);
        if (rowdb) {
            float *o = rowdb + b * 4;
            o[0] = r.evm; o[1] = r.lo; o[2] = r.up; o[3] = r.nmse;
        }
        a[0] += st[0]; a[1] += r.evm; a[2] += r.lo; a[3] += r.up; a[4] += r.nmse;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) a[i] = wave_sum_d(a[i]);
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 5; ++i) red[threadIdx.x >> 6][i] = a[i];
    __syncthreads();
    if (threadIdx.x < 5) {
        double t = 0;
        for (int w = 0; w < 16; ++w) t += red[w][threadIdx.x];
        sums[threadIdx.x] = (float)t;
    }
    if (threadIdx.x == 5) sums[5] = (float)B;
    if (threadIdx.x == 6) sums[6] = 0.f;
    if (threadIdx.x == 7) sums[7] = 0.f;
}

// out[0..3] = total, L1, evm term, acpr term; out[4..7] = d total / d sums[0..3]
__global__ void k_final(const float *__restrict__ sums, float n, float w_l1, float w_evm, float w_acpr,
                        float *__restrict__ out) {
    if (threadIdx.x != 0) return;
    const float Bg = sums[5];
    const float l1 = sums[0] / (Bg * n * 2.f);
    const float evm_mean = sums[1] / Bg, lo = sums[2] / Bg, up = sums[3] / Bg;
    const float ev = (evm_mean + 40.f) / 40.f, evm_term = fminf(fmaxf(ev, 0.f), 2.f);
    const float mx = fmaxf(lo, up), ac = (mx + 50.f) / 50.f, acpr_term = fminf(fmaxf(ac, 0.f), 2.f);
    out[0] = w_l1 * l1 + w_evm * evm_term + w_acpr * acpr_term;
    out[1] = l1;
    out[2] = evm_term;
    out[3] = acpr_term;
    out[4] = w_l1 / (Bg * n * 2.f);
    out[5] = (ev >= 0.f && ev <= 2.f) ? w_evm / (40.f * Bg) : 0.f;          // clamp passes gradient on [min, max]
    const float ga = (ac >= 0.f && ac <= 2.f) ? w_acpr / (50.f * Bg) : 0.f;
    out[6] = lo > up ? ga : (lo < up ? 0.f : 0.5f * ga);                     // a tie splits evenly, as torch.maximum
    out[7] = up > lo ? ga : (up < lo ? 0.f : 0.5f * ga);
}

__global__ __launch_bounds__(256) void k_terms_bwd(const float *__restrict__ stats, const float *__restrict__ coef,
                                                   const float *__restrict__ gscale, int64_t B, float n,
                                                   float *__restrict__ gstats) {
    const float g = gscale ? gscale[0] : 1.f;
    const float c0 = g * coef[0], c1 = g * coef[1], c2 = g * coef[2], c3 = g * coef[3];
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (int64_t)gridDim.x * 256) {
        const float *st = stats + b * 8;
        float *o = gstats + b * 8;
        // EVM_dB = 20 log10(e + 1e-10), e = sqrt(r), r = (s1/n) / max(s2/n, 1e-10)
        const float ref_power = fmaxf(st[2] / n, 1e-10f);
        const float r = (st[1] / n) / ref_power, e = sqrtf(r);
        const float d_evm = e > 0.f ? 20.f / (LN10 * (e + 1e-10f)) * (0.5f / e) / (n * ref_power) : 0.f;
        const float mainp = fmaxf(st[3], 1e-10f);
        const float rl = st[4] / mainp, ru = st[5] / mainp;
        const float dl = 10.f / (LN10 * (rl + 1e-10f)), du = 10.f / (LN10 * (ru + 1e-10f));
        const float pass = st[3] >= 1e-10f ? 1.f : 0.f;
        o[0] = c0;
        o[1] = c1 * d_evm;
        o[2] = 0.f;                                   // the reference signal carries no gradient
        o[3] = -pass * (c2 * dl * rl + c3 * du * ru) / mainp;
        o[4] = c2 * dl / mainp;
        o[5] = c3 * du / mainp;
        o[6] = 0.f;
        o[7] = 0.f;
    }
}

}  // namespace terms

extern "C" int dpd_spectral_rowdb(const float *stats, int64_t B, int64_t L, float *rowdb, float *sums, void *stream) {
    DPD_REQUIRE((stats || B == 0) && sums && B >= 0 && L >= 1, DPD_ERR_INVALID_ARG, "spectral_rowdb: bad argument");
    hipLaunchKernelGGL(terms::k_rowdb, dim3(1), dim3(1024), 0, as_stream(stream), stats, B, (float)L, rowdb, sums);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_spectral_loss_final(const float *sums, int64_t L, float w_l1, float w_evm, float w_acpr, float *out,
                                       void *stream) {
    DPD_REQUIRE(sums && out && L >= 1, DPD_ERR_INVALID_ARG, "spectral_loss_final: bad argument");
    hipLaunchKernelGGL(terms::k_final, dim3(1), dim3(64), 0, as_stream(stream), sums, (float)L, w_l1, w_evm, w_acpr, out);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_spectral_terms_backward(const float *stats, const float *coef, const float *gscale, int64_t B,
                                           int64_t L, float *gstats, void *stream) {
    DPD_REQUIRE(((stats && gstats) || B == 0) && coef && B >= 0 && L >= 1, DPD_ERR_INVALID_ARG, "spectral_terms_backward: bad argument");
    if (B == 0) return DPD_OK;
    hipLaunchKernelGGL(terms::k_terms_bwd, dim3(grid_for(B, 256)), dim3(256), 0, as_stream(stream), stats, coef, gscale, B,
                       (float)L, gstats);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_spectral_stats(const float *pred, const float *target, int64_t B, int64_t L, const uint8_t *band,
                                  float *stats, void *stream) {
    DPD_REQUIRE(((pred && stats && (band || target)) || B <= 0), DPD_ERR_INVALID_ARG, "spectral_stats: null pointer");
    DPD_REQUIRE(L >= 1, DPD_ERR_INVALID_ARG, "spectral_stats: empty rows");
    int log2n = 0;
    const bool pow2 = check_len(L, &log2n);
    DPD_REQUIRE(!band || pow2 || (L >= 2 && L <= 8192), DPD_ERR_INVALID_ARG,
                "spectral_stats: row length %lld is neither a power of two in 2..4096 nor within the direct-DFT limit of 8192", (long long)L);
    if (B <= 0) return DPD_OK;
    if (band && !pow2) {
        // the row sits in dynamic LDS next to the kernel's static words: the longest rows need more than the
        // 64 KiB a launch gets without the attribute (set once per device)
        static PerDeviceOnce attr_once;
        int attr_dev;
        if (attr_once.first_use(&attr_dev)) {
            DPD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_spectral_stats_dft),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(8192 * sizeof(float2))));
            attr_once.mark(attr_dev);
        }
        hipLaunchKernelGGL(k_spectral_stats_dft, dim3((unsigned)B), dim3(TPB), (size_t)L * sizeof(float2), as_stream(stream),
                           reinterpret_cast<const float2 *>(pred), reinterpret_cast<const float2 *>(target), L, band, stats);
        DPD_LAUNCH_CHECK();
        return DPD_OK;
    }
    if (band && L == 1024) {
        // rows of 8 KiB: 16-byte loads whenever the tensors start on a 16-byte boundary
        const bool wide = w1024::wide_ok(pred, target, nullptr);
#define LAUNCH_STATS(W, T)                                                                                              \
    hipLaunchKernelGGL((w1024::k_stats<W, T>), dim3(w1024::grid_for_rows(B)), dim3(w1024::ROWS_PER_BLOCK * 64), 0,          \
                       as_stream(stream), reinterpret_cast<const float2 *>(pred), reinterpret_cast<const float2 *>(target), \
                       B, band, stats)
        if (wide && target) LAUNCH_STATS(true, true);
        else if (wide) LAUNCH_STATS(true, false);
        else if (target) LAUNCH_STATS(false, true);
        else LAUNCH_STATS(false, false);
#undef LAUNCH_STATS
        DPD_LAUNCH_CHECK();
        return DPD_OK;
    }
    if (band && L == 256) {
        hipLaunchKernelGGL(w256::k_stats, dim3(w256::grid_for_rows(B)), dim3(w256::WAVES * 64), 0, as_stream(stream),
                           reinterpret_cast<const float2 *>(pred), reinterpret_cast<const float2 *>(target), B, band, stats);
        DPD_LAUNCH_CHECK();
        return DPD_OK;
    }
    const size_t lds = band ? (size_t)(L + L / 2) * sizeof(float2) : 0;
    hipLaunchKernelGGL(k_spectral_stats, dim3((unsigned)B), dim3(TPB), lds, as_stream(stream),
                       reinterpret_cast<const float2 *>(pred), reinterpret_cast<const float2 *>(target), L, log2n,
                       band, stats);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_spectral_backward(const float *pred, const float *target, int64_t B, int64_t L,
                                     const uint8_t *band, const float *gstats, float *dpred, void *stream) {
    DPD_REQUIRE(((pred && gstats && dpred && (band || target)) || B <= 0), DPD_ERR_INVALID_ARG, "spectral_backward: null pointer");
    DPD_REQUIRE(L >= 1, DPD_ERR_INVALID_ARG, "spectral_backward: empty rows");
    int log2n = 0;
    const bool pow2 = check_len(L, &log2n);
    DPD_REQUIRE(!band || pow2 || (L >= 2 && L <= 8192), DPD_ERR_INVALID_ARG,
                "spectral_backward: row length %lld is neither a power of two in 2..4096 nor within the direct-DFT limit of 8192", (long long)L);
    if (B <= 0) return DPD_OK;
    if (band && !pow2) {
        // twiddles and the weighted spectrum in dynamic LDS: up to 128 KiB, above what a launch gets without the attribute
        static PerDeviceOnce attr_once;
        int attr_dev;
        if (attr_once.first_use(&attr_dev)) {
            DPD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_spectral_bwd_dft),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 8192 * sizeof(float2))));
            attr_once.mark(attr_dev);
        }
        hipLaunchKernelGGL(k_spectral_bwd_dft, dim3((unsigned)B), dim3(TPB), (size_t)2 * L * sizeof(float2), as_stream(stream),
                           reinterpret_cast<const float2 *>(pred), reinterpret_cast<const float2 *>(target), L, band, gstats,
                           reinterpret_cast<float2 *>(dpred));
        DPD_LAUNCH_CHECK();
        return DPD_OK;
    }
    if (band && L == 1024) {
        const bool wide = w1024::wide_ok(pred, target, dpred);
#define LAUNCH_BWD(W, T)                                                                                                 \
    hipLaunchKernelGGL((w1024::k_bwd<W, T>), dim3(w1024::grid_for_rows(B, 2)), dim3(w1024::ROWS_PER_BLOCK * 64), 0,          \
                       as_stream(stream), reinterpret_cast<const float2 *>(pred), reinterpret_cast<const float2 *>(target), \
                       B, band, gstats, reinterpret_cast<float2 *>(dpred))
        if (wide && target) LAUNCH_BWD(true, true);
        else if (wide) LAUNCH_BWD(true, false);
        else if (target) LAUNCH_BWD(false, true);
        else LAUNCH_BWD(false, false);
#undef LAUNCH_BWD
        DPD_LAUNCH_CHECK();
        return DPD_OK;
    }
    if (band && L == 256) {
        hipLaunchKernelGGL(w256::k_bwd, dim3(w256::grid_for_rows(B)), dim3(w256::WAVES * 64), 0, as_stream(stream),
                           reinterpret_cast<const float2 *>(pred), reinterpret_cast<const float2 *>(target), B, band, gstats,
                           reinterpret_cast<float2 *>(dpred));
        DPD_LAUNCH_CHECK();
        return DPD_OK;
    }
    const size_t lds = band ? (size_t)(L + L / 2) * sizeof(float2) : 0;
    hipLaunchKernelGGL(k_spectral_bwd, dim3((unsigned)B), dim3(TPB), lds, as_stream(stream),
                       reinterpret_cast<const float2 *>(pred), reinterpret_cast<const float2 *>(target), L, log2n,
                       band, gstats, reinterpret_cast<float2 *>(dpred));
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}
