// Generator kernels, VALU family: one sample per lane, weights fetched as wave-uniform scalars.
// Used for small batches, for pre-assembled feature input, and as the on-device cross-check of the
// MFMA family.  Reference semantics: models/tdnn_generator.py:158-195 (taps), :257-287 (float),
// :394-456 (QAT), :46-103 (fake quantisation, straight-through gradient).
#include "gen_internal.h"

namespace {

constexpr int FWD_TPB = 256;
constexpr int BWD_TPB = 64;         // one wave per block: the block's LDS panel is wave-private
constexpr int ACT_PITCH = 79;       // xq[30] | h1q[32] | h2q[16] | 1.0
constexpr int GZ_PITCH = 51;        // gz1[32] | gz2[16] | gz3[2] | pad
constexpr int BWD_EPT = 25;         // gradient entries per thread: ceil(1554 / 64)

template <bool PRE>
DEV void load_features(const float *__restrict__ x, int64_t n, int64_t S, int64_t L, float (&f)[30]) {
    if (PRE) {
        const float *p = x + n * 30;
#pragma unroll
        for (int k = 0; k < 30; ++k) f[k] = p[k];
    } else {
        int64_t b = n / L, t = n - b * L;
        const float2 *p = reinterpret_cast<const float2 *>(x) + b * S + t;
        // p[5] is the current sample, p[5-g] is tap g
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            float2 v = p[5 - g];
            float e = sqrtf(v.x * v.x + v.y * v.y);
            float e2 = e * e;
            f[2 + g] = e;
            f[8 + g] = e2;
            f[14 + g] = e2 * e2;
            if (g == 0) {
                f[0] = v.x;
                f[1] = v.y;
            } else {
                f[20 + 2 * (g - 1)] = v.x;
                f[21 + 2 * (g - 1)] = v.y;
            }
        }
    }
}

template <int O, int K>
DEV void dense(const float *__restrict__ w, const float *__restrict__ b, const float (&in)[K], float (&out)[O]) {
#pragma unroll
    for (int j = 0; j < O; ++j) {
        float a = b[j];
#pragma unroll
        for (int k = 0; k < K; ++k) a = __builtin_fmaf(w[j * K + k], in[k], a);
        out[j] = a;
    }
}

template <int K>
DEV float absmax_of(const float (&v)[K], float m) {
#pragma unroll
    for (int k = 0; k < K; ++k) m = fmaxf(m, fabsf(v[k]));
    return m;
}

// DEPTH 0..3: QAT abs-max passes (3 also writes the unquantised output); DEPTH 4: float forward.
template <bool PRE, bool QAT, int DEPTH>
__global__ __launch_bounds__(FWD_TPB) void k_gen_fwd(const float *__restrict__ x, int64_t N, int64_t S, int64_t L,
                                                     const float *__restrict__ wq, float slope,
                                                     float *__restrict__ absmax, float *__restrict__ out) {
    // weights are read from LDS at wave-uniform addresses (broadcast reads); 1554 values do not fit SGPRs
    __shared__ float sw[G_NP + 2];
    for (int i = threadIdx.x; i < G_NP; i += FWD_TPB) sw[i] = wq[i];
    __syncthreads();
    QS s0 = qs_one(), s1 = qs_one(), s2 = qs_one();
    if (QAT) {
        if (DEPTH >= 1) s0 = qs_of(absmax[0]);
        if (DEPTH >= 2) s1 = qs_of(absmax[1]);
        if (DEPTH >= 3) s2 = qs_of(absmax[2]);
    }
    float amax = 0.f;
    for (int64_t n = (int64_t)blockIdx.x * FWD_TPB + threadIdx.x; n < N; n += (int64_t)gridDim.x * FWD_TPB) {
        float f[30];
        load_features<PRE>(x, n, S, L, f);
        if (QAT && DEPTH == 0) {
            amax = absmax_of(f, amax);
            continue;
        }
        if (QAT) {
#pragma unroll
            for (int k = 0; k < 30; ++k) f[k] = fq16(f[k], s0);
        }
        float h1[32];
        dense<32, 30>(sw + G_W1, sw + G_B1, f, h1);
#pragma unroll
        for (int j = 0; j < 32; ++j) h1[j] = lrelu(h1[j], slope);
        if (QAT && DEPTH == 1) {
            amax = absmax_of(h1, amax);
            continue;
        }
        if (QAT) {
#pragma unroll
            for (int j = 0; j < 32; ++j) h1[j] = fq16(h1[j], s1);
        }
        float h2[16];
        dense<16, 32>(sw + G_W2, sw + G_B2, h1, h2);
#pragma unroll
        for (int j = 0; j < 16; ++j) h2[j] = lrelu(h2[j], slope);
        if (QAT && DEPTH == 2) {
            amax = absmax_of(h2, amax);
            continue;
        }
        if (QAT) {
#pragma unroll
            for (int j = 0; j < 16; ++j) h2[j] = fq16(h2[j], s2);
        }
        float z3[2];
        dense<2, 16>(sw + G_W3, sw + G_B3, h2, z3);
        float2 y = make_float2(tanhf(z3[0]), tanhf(z3[1]));
        if (QAT) amax = fmaxf(amax, fmaxf(fabsf(y.x), fabsf(y.y)));
        reinterpret_cast<float2 *>(out)[n] = y;
    }
    if (QAT) {
        amax = wave_max(amax);
        if ((threadIdx.x & 63) == 0) atomic_max_nonneg(absmax + DEPTH, amax);
    }
}

__global__ __launch_bounds__(256) void k_quantize_output(float *__restrict__ out, int64_t n,
                                                         const float *__restrict__ absmax) {
    const QS s = qs_of(absmax[3]);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = fq16(out[i], s);
}

// Recomputes the forward for one sample per lane, back-propagates, and reduces the parameter
// gradient of the block's 64 samples through an LDS panel: lane e owns entries e, e+64, ...
template <bool PRE, bool QAT, bool MSE>
__global__ __launch_bounds__(BWD_TPB) void k_gen_bwd(const float *__restrict__ x, int64_t N, int64_t S, int64_t L,
                                                     const float *__restrict__ wq, float slope,
                                                     const float *__restrict__ absmax,
                                                     const float *__restrict__ grad_out,
                                                     const float *__restrict__ target, float mse_scale,
                                                     float *__restrict__ grad_feat, float *__restrict__ slab) {
    __shared__ float act[BWD_TPB * ACT_PITCH];
    __shared__ float gz[BWD_TPB * GZ_PITCH];
    __shared__ float sw[G_NP + 2];
    const int tid = threadIdx.x;
    for (int i = tid; i < G_NP; i += BWD_TPB) sw[i] = wq[i];
    __syncthreads();
    QS s0 = qs_one(), s1 = qs_one(), s2 = qs_one(), s3 = qs_one();
    if (QAT) {
        s0 = qs_of(absmax[0]);
        s1 = qs_of(absmax[1]);
        s2 = qs_of(absmax[2]);
        s3 = qs_of(absmax[3]);
    }
    // entry ownership
    int gcol[BWD_EPT], acol[BWD_EPT];
#pragma unroll
    for (int i = 0; i < BWD_EPT; ++i) {
        int e = tid + 64 * i;
        int gc = 50, ac = 78;  // gz pad column holds 0, act column 78 holds 1
        if (e < G_B1) {
            gc = e / 30;
            ac = e % 30;
        } else if (e < G_W2) {
            gc = e - G_B1;
        } else if (e < G_B2) {
            gc = 32 + (e - G_W2) / 32;
            ac = 30 + (e - G_W2) % 32;
        } else if (e < G_W3) {
            gc = 32 + (e - G_B2);
        } else if (e < G_B3) {
            gc = 48 + (e - G_W3) / 16;
            ac = 62 + (e - G_W3) % 16;
        } else if (e < G_NP) {
            gc = 48 + (e - G_B3);
        }
        gcol[i] = gc;
        acol[i] = ac;
    }
    float acc[BWD_EPT];
#pragma unroll
    for (int i = 0; i < BWD_EPT; ++i) acc[i] = 0.f;
    float loss = 0.f;

    const int64_t ntiles = (N + BWD_TPB - 1) / BWD_TPB;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n = tile * BWD_TPB + tid;
        float *arow = act + tid * ACT_PITCH;
        float *grow = gz + tid * GZ_PITCH;
        if (n < N) {
            float f[30];
            load_features<PRE>(x, n, S, L, f);
            if (QAT) {
#pragma unroll
                for (int k = 0; k < 30; ++k) f[k] = fq16(f[k], s0);
            }
            float h1[32], m1[32];
            dense<32, 30>(sw + G_W1, sw + G_B1, f, h1);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                m1[j] = lrelu_mask(h1[j], slope);
                h1[j] = lrelu(h1[j], slope);
                if (QAT) h1[j] = fq16(h1[j], s1);
            }
            float h2[16], m2[16];
            dense<16, 32>(sw + G_W2, sw + G_B2, h1, h2);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                m2[j] = lrelu_mask(h2[j], slope);
                h2[j] = lrelu(h2[j], slope);
                if (QAT) h2[j] = fq16(h2[j], s2);
            }
            float z3[2];
            dense<2, 16>(sw + G_W3, sw + G_B3, h2, z3);
            float y0 = tanhf(z3[0]), y1 = tanhf(z3[1]);
            float g0, g1;
            if (MSE) {
                float o0 = QAT ? fq16(y0, s3) : y0, o1 = QAT ? fq16(y1, s3) : y1;
                float2 t = reinterpret_cast<const float2 *>(target)[n];
                float d0 = o0 - t.x, d1 = o1 - t.y;
                loss += d0 * d0 + d1 * d1;
                g0 = d0 * mse_scale;
                g1 = d1 * mse_scale;
            } else {
                float2 g = reinterpret_cast<const float2 *>(grad_out)[n];
                g0 = g.x;
                g1 = g.y;
            }
            float gz3_0 = g0 * (1.f - y0 * y0), gz3_1 = g1 * (1.f - y1 * y1);
            float gz2[16];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                gz2[k] = __builtin_fmaf(gz3_1, sw[G_W3 + 16 + k], gz3_0 * sw[G_W3 + k]) * m2[k];
            float gz1[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) a = __builtin_fmaf(gz2[j], sw[G_W2 + j * 32 + k], a);
                gz1[k] = a * m1[k];
            }
            if (PRE && grad_feat != nullptr) {
#pragma unroll
                for (int k = 0; k < 30; ++k) {
                    float a = 0.f;
#pragma unroll
                    for (int j = 0; j < 32; ++j) a = __builtin_fmaf(gz1[j], sw[G_W1 + j * 30 + k], a);
                    grad_feat[n * 30 + k] = a;
                }
            }
#pragma unroll
            for (int k = 0; k < 30; ++k) arow[k] = f[k];
#pragma unroll
            for (int k = 0; k < 32; ++k) arow[30 + k] = h1[k];
#pragma unroll
            for (int k = 0; k < 16; ++k) arow[62 + k] = h2[k];
            arow[78] = 1.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) grow[k] = gz1[k];
#pragma unroll
            for (int k = 0; k < 16; ++k) grow[32 + k] = gz2[k];
            grow[48] = gz3_0;
            grow[49] = gz3_1;
            grow[50] = 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < GZ_PITCH; ++k) grow[k] = 0.f;
#pragma unroll
            for (int k = 0; k < ACT_PITCH; ++k) arow[k] = 0.f;
        }
        __syncthreads();
        for (int s = 0; s < BWD_TPB; ++s) {
            const float *ar = act + s * ACT_PITCH;
            const float *gr = gz + s * GZ_PITCH;
#pragma unroll
            for (int i = 0; i < BWD_EPT; ++i) acc[i] = __builtin_fmaf(gr[gcol[i]], ar[acol[i]], acc[i]);
        }
        __syncthreads();
    }
    float *row = slab + (int64_t)blockIdx.x * G_SLAB;
#pragma unroll
    for (int i = 0; i < BWD_EPT; ++i) {
        int e = tid + 64 * i;
        if (e < G_NP) row[e] = acc[i];
    }
    loss = wave_sum(loss);
    if (tid == 0) row[G_NP] = loss;
}

// out[e] = sum over partial rows of slab[p][e] for e < n_out, and extra[0] = extra_scale * sum of column
// n_out.  64 columns per workgroup; the four waves take every fourth row (coalesced 256-byte reads),
// accumulate in double and are combined through LDS in a fixed order (bitwise reproducible).
__global__ __launch_bounds__(256) void k_reduce_slab(const float *__restrict__ slab, int n_partials, int row_len,
                                                     int n_out, float *__restrict__ grads, float *__restrict__ loss,
                                                     float loss_scale) {
    __shared__ double part[4][64];
    const int c = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + c;
    double a = 0.0;
    if (e <= n_out) {
        const float *p = slab + e;
#pragma unroll 8
        for (int r = s; r < n_partials; r += 4) a += (double)p[(int64_t)r * row_len];
    }
    part[s][c] = a;
    __syncthreads();
    if (s == 0 && e <= n_out) {
        a = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        if (e < n_out)
            grads[e] = (float)a;
        else if (loss != nullptr)
            loss[0] = (float)(a * (double)loss_scale);
    }
}

// One wave: row r < 50 of the three weight matrices per lane.
__global__ __launch_bounds__(64) void k_prepare_weights(const float *__restrict__ w1, const float *__restrict__ b1,
                                                        const float *__restrict__ w2, const float *__restrict__ b2,
                                                        const float *__restrict__ w3, const float *__restrict__ b3,
                                                        int qat, float *__restrict__ wq, float *__restrict__ absmax) {
    const int r = threadIdx.x;
    const float *src = nullptr;
    float *dst = nullptr;
    int len = 0, which = 3;
    if (r < 32) {
        src = w1 + r * 30; dst = wq + G_W1 + r * 30; len = 30; which = 0;
    } else if (r < 48) {
        src = w2 + (r - 32) * 32; dst = wq + G_W2 + (r - 32) * 32; len = 32; which = 1;
    } else if (r < 50) {
        src = w3 + (r - 48) * 16; dst = wq + G_W3 + (r - 48) * 16; len = 16; which = 2;
    }
    float am = 0.f;
    for (int k = 0; k < len; ++k) am = fmaxf(am, fabsf(src[k]));
    const QS s = qs_of(am);
    for (int k = 0; k < len; ++k) dst[k] = qat ? fq16(src[k], s) : src[k];
    float a0 = wave_max(which == 0 ? am : 0.f);
    float a1 = wave_max(which == 1 ? am : 0.f);
    float a2 = wave_max(which == 2 ? am : 0.f);
    if (r == 0) {
        absmax[4] = a0; absmax[5] = a1; absmax[6] = a2; absmax[7] = 0.f;
    }
    if (r < 32) wq[G_B1 + r] = b1[r];
    if (r < 16) wq[G_B2 + r] = b2[r];
    if (r < 2) wq[G_B3 + r] = b3[r];
}

__global__ __launch_bounds__(256) void k_tap_assembly(const float *__restrict__ iq, int64_t N, int64_t S, int64_t L,
                                                      float *__restrict__ feat) {
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
        float f[30];
        load_features<false>(iq, n, S, L, f);
#pragma unroll
        for (int k = 0; k < 30; ++k) feat[n * 30 + k] = f[k];
    }
}

// Adjoint of the tap assembly (the reference's MemoryTapAssembly is differentiable torch, models/tdnn_generator.py:158-195):
// dx[b, s] = sum over the outputs n = s - 5 + g (g = 0..5, 0 <= n < L) that read point s as their tap g of
//   the gradient of the IQ columns of that tap + (gf_e + 2 e gf_e2 + 4 e^3 gf_e4) x / e,   e = |x[b, s]|
// (columns: 0, 1 current IQ; 2 + g, 8 + g, 14 + g the envelope and its powers of tap g; 20 + 2 (g - 1), + 1 the IQ of
// tap g >= 1).  One thread per point, a gather: no atomics, bitwise reproducible.  At e = 0 the envelope terms are
// dropped (torch's sqrt has no finite derivative there).
__global__ __launch_bounds__(256) void k_tap_adjoint(const float2 *__restrict__ iq, int64_t B, int64_t S, int64_t L,
                                                     const float *__restrict__ gfeat, float2 *__restrict__ dx) {
    const int64_t n_pts = B * S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pts; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / S, s = i - b * S;
        const float2 v = iq[i];
        const float e = sqrtf(v.x * v.x + v.y * v.y), e2 = e * e;
        float gi = 0.f, gq = 0.f, ge = 0.f;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int64_t n = s - 5 + g;
            if (n < 0 || n >= L) continue;
            const float *f = gfeat + (b * L + n) * 30;
            gi += g == 0 ? f[0] : f[20 + 2 * (g - 1)];
            gq += g == 0 ? f[1] : f[21 + 2 * (g - 1)];
            ge += f[2 + g] + 2.f * e * f[8 + g] + 4.f * e2 * e * f[14 + g];
        }
        const float r = e > 0.f ? ge / e : 0.f;
        dx[i] = make_float2(gi + r * v.x, gq + r * v.y);
    }
}

template <bool PRE, bool QAT>
int launch_fwd(const GenProblem &p, int depth, float *out, hipStream_t st) {
    const int grid = grid_for(p.N, FWD_TPB);
#define GO(D) hipLaunchKernelGGL((k_gen_fwd<PRE, QAT, D>), dim3(grid), dim3(FWD_TPB), 0, st, p.x, p.N, p.S, p.L, p.wq, \
                                 p.slope, p.absmax, out)
    if constexpr (!QAT) {
        GO(4);
    } else {
        switch (depth) {
            case 0: GO(0); break;
            case 1: GO(1); break;
            case 2: GO(2); break;
            default: GO(3); break;
        }
    }
#undef GO
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

template <bool PRE, bool QAT, bool MSE>
int launch_bwd(const GenProblem &p, const float *grad_out, const float *target, float *grad_feat, float *slab,
               int grid, hipStream_t st) {
    float mse_scale = MSE ? (float)(2.0 / (2.0 * (double)p.N)) : 0.f;
    hipLaunchKernelGGL((k_gen_bwd<PRE, QAT, MSE>), dim3(grid), dim3(BWD_TPB), 0, st, p.x, p.N, p.S, p.L, p.wq, p.slope,
                       p.absmax, grad_out, target, mse_scale, grad_feat, slab);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

}  // namespace

int gen_valu_forward(const GenProblem &p, int depth, float *out, hipStream_t st) {
    if (p.pre) return p.qat ? launch_fwd<true, true>(p, depth, out, st) : launch_fwd<true, false>(p, depth, out, st);
    return p.qat ? launch_fwd<false, true>(p, depth, out, st) : launch_fwd<false, false>(p, depth, out, st);
}

int gen_valu_backward(const GenProblem &p, const float *grad_out, const float *target, float *grad_feat, float *slab,
                      int *n_partials, hipStream_t st) {
    const int grid = grid_for(p.N, BWD_TPB, 256 * 4);
    *n_partials = grid;
    const bool mse = (grad_out == nullptr);
#define GO(PRE, QAT, MSE) return launch_bwd<PRE, QAT, MSE>(p, grad_out, target, grad_feat, slab, grid, st)
    if (p.pre) {
        if (p.qat) { if (mse) GO(true, true, true); else GO(true, true, false); }
        else       { if (mse) GO(true, false, true); else GO(true, false, false); }
    } else {
        if (p.qat) { if (mse) GO(false, true, true); else GO(false, true, false); }
        else       { if (mse) GO(false, false, true); else GO(false, false, false); }
    }
#undef GO
}

int gen_reduce_slab(const float *slab, int n_partials, float *grads, float *loss, float loss_scale, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_slab, dim3((G_NP + 1 + 63) / 64), dim3(256), 0, st, slab, n_partials, G_SLAB, G_NP,
                       grads, loss, loss_scale);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

int gen_reduce_slab_generic(const float *slab, int n_partials, int row_len, int n_out, float *grads, float *extra,
                            float extra_scale, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_slab, dim3((n_out + 1 + 63) / 64), dim3(256), 0, st, slab, n_partials, row_len, n_out,
                       grads, extra, extra_scale);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

int gen_launch_prepare(const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                       const float *b3, int qat, float *wq, float *absmax, hipStream_t st) {
    hipLaunchKernelGGL(k_prepare_weights, dim3(1), dim3(64), 0, st, w1, b1, w2, b2, w3, b3, qat, wq, absmax);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

int gen_launch_taps(const float *iq, int64_t B, int64_t S, float *feat, hipStream_t st) {
    const int64_t L = S - 5, N = B * L;
    hipLaunchKernelGGL(k_tap_assembly, dim3(grid_for(N, 256)), dim3(256), 0, st, iq, N, S, L, feat);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

int gen_launch_tap_adjoint(const float *iq, int64_t B, int64_t S, const float *gfeat, float *dx, hipStream_t st) {
    const int64_t L = S - 5;
    hipLaunchKernelGGL(k_tap_adjoint, dim3(grid_for(B * S, 256)), dim3(256), 0, st, reinterpret_cast<const float2 *>(iq), B, S, L,
                       gfeat, reinterpret_cast<float2 *>(dx));
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

int gen_launch_quantize_output(float *out, int64_t n, const float *absmax, hipStream_t st) {
    hipLaunchKernelGGL(k_quantize_output, dim3(grid_for(n, 256)), dim3(256), 0, st, out, n, absmax);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}
