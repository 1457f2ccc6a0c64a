// Conditional critic 4 -> 64 -> 32 -> 16 -> 1 (reference models/discriminator.py:103-136) and the
// WGAN-GP gradient penalty with its double backward (:139-191).  The weights arrive already
// spectrally normalised (flat wb[2945]); LeakyReLU sub-gradient at 0 is the slope.
//
// VALU family: one sample per lane, weights broadcast from LDS.  Parameter gradients are rank-1
// updates summed over samples; each wave parks its 64 samples' left/right factors in an LDS panel
// and every lane accumulates its share of the 2945 entries over the panel.
//
// Gradient penalty, with masks D_i = (z_i > 0 ? 1 : slope) of x_hat = a r + (1 - a) f:
//   v3 = W4 * D3,  v2 = (v3 W3) * D2,  v1 = (v2 W2) * D1,  g = v1 W1[:, 0:2]
//   gp = mean_b (||g_b|| - 1)^2 over rows b of L samples (norm over all 2L entries).
// The net is piecewise linear, so d gp / d W is the backward of that linear chain with the masks
// held constant: with gh = c_b g (c_b = 2 (||g_b|| - 1) / (||g_b|| B)),
//   dW1[:, 0:2] += v1^T gh;  dv1 = W1[:, 0:2] gh;  du1 = dv1 * D1;  dW2 += v2^T du1;  dv2 = W2 du1;
//   du2 = dv2 * D2;  dW3 += v3^T du2;  dv3 = W3 du2;  dW4 += dv3 * D3.   Bias gradients are zero.
#include "common.h"

int gen_reduce_slab_generic(const float *slab, int n_partials, int row_len, int n_out, float *grads, float *extra,
                            float extra_scale, hipStream_t st);

int critic_mfma_forward(const float *x, const float *cond, int64_t N, const float *wb, float slope, float *score,
                        float *score_sum, hipStream_t st);
int critic_mfma_backward(const float *x, const float *cond, int64_t N, const float *wb, float slope, const float *gscore,
                         float gconst, float *grads, float *gx, float *score_sum, float *ws, hipStream_t st);
size_t critic_mfma_gp_ws_bytes(int64_t B, int64_t L, size_t ws_bytes);
int critic_mfma_gp(const float *real, const float *fake, const float *cond, const float *alpha, int64_t B, int64_t L,
                   const float *wb, float slope, float *gp, float *rownorm, float *grads, void *ws, size_t ws_bytes,
                   hipStream_t st);

// kernel family: 0 = by size (MFMA from 1024 samples), 1 = VALU, 2 = MFMA
static int g_critic_impl = 0;
extern int g_gen_deterministic;
extern int g_critic_deterministic;
extern int g_chain_carry;
extern int g_gp_fused;
static bool critic_use_mfma(int64_t N) { return g_critic_impl == 2 || (g_critic_impl == 0 && N >= 1024); }

extern "C" int dpd_set_option(const char *name, int value) {
    DPD_REQUIRE(name, DPD_ERR_INVALID_ARG, "set_option: null name");
    if (__builtin_strcmp(name, "critic_impl") == 0 && value >= 0 && value <= 2) {
        g_critic_impl = value;
        return DPD_OK;
    }
    if (__builtin_strcmp(name, "gen_deterministic") == 0 && (value == 0 || value == 1)) {
        g_gen_deterministic = value;
        return DPD_OK;
    }
    if (__builtin_strcmp(name, "critic_deterministic") == 0 && (value == 0 || value == 1)) {
        g_critic_deterministic = value;
        return DPD_OK;
    }
    if (__builtin_strcmp(name, "chain_carry") == 0 && value >= 0 && value <= 2) {
        g_chain_carry = value;
        return DPD_OK;
    }
    if (__builtin_strcmp(name, "gp_fused") == 0 && (value == 0 || value == 1)) {
        g_gp_fused = value;
        return DPD_OK;
    }
    dpd_set_error("set_option: unknown option %s or bad value %d", name, value);
    return DPD_ERR_INVALID_ARG;
}

namespace {

constexpr int TPB = 64;               // one wave per block: the LDS panel is wave-private
constexpr int NL = 113, NR = 117;     // panel widths: left factors / right factors
constexpr int LP = 115, RP = 119;     // pitches (odd)
constexpr int EPT = 47;               // entries per lane: ceil(2945 / 64)

#define MBIT(h) ((h) > 0.f ? 1u : 0u)
// keeps the scheduler from hoisting every weight load of an unrolled layer to its top (register spills)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

DEV void load_wb(float *sw, const float *__restrict__ wb) {
    for (int i = threadIdx.x; i < C_NP; i += blockDim.x) sw[i] = wb[i];
    __syncthreads();
}

struct Fwd {
    float in[4];
    float h1[64], h2[32], h3[16];   // post-activation; sign(h) == sign(z), so the mask is read off h
    float score;
};

DEV void critic_fwd(Fwd &f, const float *sw, float slope) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        float a = sw[C_B1 + j];
#pragma unroll
        for (int k = 0; k < 4; ++k) a = __builtin_fmaf(sw[C_W1 + j * 4 + k], f.in[k], a);
        f.h1[j] = lrelu(a, slope);
        FENCE();
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        float a = sw[C_B2 + j];
#pragma unroll
        for (int k = 0; k < 64; ++k) a = __builtin_fmaf(sw[C_W2 + j * 64 + k], f.h1[k], a);
        f.h2[j] = lrelu(a, slope);
        FENCE();
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float a = sw[C_B3 + j];
#pragma unroll
        for (int k = 0; k < 32; ++k) a = __builtin_fmaf(sw[C_W3 + j * 32 + k], f.h2[k], a);
        f.h3[j] = lrelu(a, slope);
        FENCE();
    }
    float a = sw[C_B4];
#pragma unroll
    for (int k = 0; k < 16; ++k) a = __builtin_fmaf(sw[C_W4 + k], f.h3[k], a);
    f.score = a;
}

__global__ __launch_bounds__(256) void k_critic_fwd(const float2 *__restrict__ x, const float2 *__restrict__ cond,
                                                    int64_t N, const float *__restrict__ wb, float slope,
                                                    float *__restrict__ score, float *__restrict__ score_sum) {
    __shared__ float sw[C_NP + 3];
    load_wb(sw, wb);
    float acc = 0.f;
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
        Fwd f;
        float2 a = x[n], c = cond[n];
        f.in[0] = a.x; f.in[1] = a.y; f.in[2] = c.x; f.in[3] = c.y;
        critic_fwd(f, sw, slope);
        if (score) score[n] = f.score;
        acc += f.score;
    }
    if (score_sum) {
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) atomicAdd(score_sum, acc);
    }
}

// (left column, right column) of flat entry e for the ordinary backward:
//   left  panel: gz1[64] | gz2[32] | gz3[16] | gz4[1]            (+ a zero column 113 is not needed)
//   right panel: in[4] | h1[64] | h2[32] | h3[16] | 1.0
DEV void entry_cols_bwd(int e, int &lc, int &rc) {
    lc = 0; rc = 116;  // default: harmless (entries past the end are never stored)
    if (e < C_B1) { lc = e / 4; rc = e % 4; }
    else if (e < C_W2) { lc = e - C_B1; rc = 116; }
    else if (e < C_B2) { lc = 64 + (e - C_W2) / 64; rc = 4 + (e - C_W2) % 64; }
    else if (e < C_W3) { lc = 64 + (e - C_B2); rc = 116; }
    else if (e < C_B3) { lc = 96 + (e - C_W3) / 32; rc = 68 + (e - C_W3) % 32; }
    else if (e < C_W4) { lc = 96 + (e - C_B3); rc = 116; }
    else if (e < C_B4) { lc = 112; rc = 100 + (e - C_W4); }
    else if (e < C_NP) { lc = 112; rc = 116; }
}

// acc[i] += sum over the panel's 64 samples of left[s][lc] * right[s][rc] for this lane's entry i.
// Entry-major order keeps one accumulator live at a time; the column pair is recomputed per tile.
template <int LPITCH, int RPITCH, typename ColFn>
DEV void panel_accumulate(const float *lp, const float *rp, int tid, float (&acc)[EPT], ColFn cols) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        int lc, rc;
        cols(tid + 64 * i, lc, rc);
        const float *l = lp + lc, *r = rp + rc;
        float a = acc[i];
#pragma unroll 8
        for (int s = 0; s < TPB; ++s) a = __builtin_fmaf(l[s * LPITCH], r[s * RPITCH], a);
        acc[i] = a;
    }
}

template <bool WGRAD>
__global__ __launch_bounds__(TPB) void k_critic_bwd(const float2 *__restrict__ x, const float2 *__restrict__ cond,
                                                    int64_t N, const float *__restrict__ wb, float slope,
                                                    const float *__restrict__ gscore, float gconst,
                                                    float2 *__restrict__ gx, float *__restrict__ slab,
                                                    float *__restrict__ score_sum) {
    extern __shared__ float dyn_lds[];   // > 64 KiB in total: launched with an explicit dynamic size
    float *sw = dyn_lds, *lp = sw + (C_NP + 3), *rp = lp + TPB * LP;
    load_wb(sw, wb);
    const int tid = threadIdx.x;
    float acc[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) acc[i] = 0.f;
    float ssum = 0.f;
    const int64_t ntiles = (N + TPB - 1) / TPB;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n = tile * TPB + tid;
        float *l = lp + tid * LP, *r = rp + tid * RP;
        if (n < N) {
            Fwd f;
            float2 a = x[n], c = cond[n];
            f.in[0] = a.x; f.in[1] = a.y; f.in[2] = c.x; f.in[3] = c.y;
            critic_fwd(f, sw, slope);
            ssum += f.score;
            const float gs = gscore ? gscore[n] : gconst;
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = f.in[k];
#pragma unroll
            for (int k = 0; k < 64; ++k) r[4 + k] = f.h1[k];
#pragma unroll
            for (int k = 0; k < 32; ++k) r[68 + k] = f.h2[k];
#pragma unroll
            for (int k = 0; k < 16; ++k) r[100 + k] = f.h3[k];
            r[116] = 1.f;
            float gz3[16], gz2[32], gz1[64];
#pragma unroll
            for (int k = 0; k < 16; ++k) gz3[k] = gs * sw[C_W4 + k] * lrelu_mask(f.h3[k], slope);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) s = __builtin_fmaf(gz3[j], sw[C_W3 + j * 32 + k], s);
                gz2[k] = s * lrelu_mask(f.h2[k], slope);
                FENCE();
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 32; ++j) s = __builtin_fmaf(gz2[j], sw[C_W2 + j * 64 + k], s);
                gz1[k] = s * lrelu_mask(f.h1[k], slope);
                FENCE();
            }
            if (gx) {
                float g0 = 0.f, g1 = 0.f;
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    g0 = __builtin_fmaf(gz1[j], sw[C_W1 + j * 4 + 0], g0);
                    g1 = __builtin_fmaf(gz1[j], sw[C_W1 + j * 4 + 1], g1);
                }
                gx[n] = make_float2(g0, g1);
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) l[k] = gz1[k];
#pragma unroll
            for (int k = 0; k < 32; ++k) l[64 + k] = gz2[k];
#pragma unroll
            for (int k = 0; k < 16; ++k) l[96 + k] = gz3[k];
            l[112] = gs;
        } else {
            for (int k = 0; k < NL; ++k) l[k] = 0.f;
            for (int k = 0; k < NR; ++k) r[k] = 0.f;
        }
        if (WGRAD) {
            __syncthreads();
            panel_accumulate<LP, RP>(lp, rp, tid, acc, [](int e, int &lc, int &rc) { entry_cols_bwd(e, lc, rc); });
            __syncthreads();
        }
    }
    if (score_sum) {   // optional sum of scores: one atomic per wave (order-dependent in the last bits)
        ssum = wave_sum(ssum);
        if (tid == 0) atomicAdd(score_sum, ssum);
    }
    if (!WGRAD) return;
    float *row = slab + (int64_t)blockIdx.x * C_SLAB;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        int e = tid + 64 * i;
        if (e < C_NP) row[e] = acc[i];
    }
}

// ---------------------------------------------------------------------------- gradient penalty
// pass A: per sample g (2 floats) and the 112 mask bits
__global__ __launch_bounds__(256) void k_gp_grad(const float2 *__restrict__ real, const float2 *__restrict__ fake,
                                                 const float2 *__restrict__ cond, const float *__restrict__ alpha,
                                                 int64_t N, int64_t L, const float *__restrict__ wb, float slope,
                                                 float2 *__restrict__ g, uint4 *__restrict__ masks) {
    __shared__ float sw[C_NP + 3];
    load_wb(sw, wb);
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
        const float a = alpha[n / L];
        const float2 r = real[n], fk = fake[n], c = cond[n];
        Fwd f;
        f.in[0] = a * r.x + (1.f - a) * fk.x;
        f.in[1] = a * r.y + (1.f - a) * fk.y;
        f.in[2] = c.x; f.in[3] = c.y;
        critic_fwd(f, sw, slope);
        float v3[16], v2[32], v1[64];
#pragma unroll
        for (int k = 0; k < 16; ++k) v3[k] = sw[C_W4 + k] * lrelu_mask(f.h3[k], slope);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) s = __builtin_fmaf(v3[j], sw[C_W3 + j * 32 + k], s);
            v2[k] = s * lrelu_mask(f.h2[k], slope);
                FENCE();
        }
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 32; ++j) s = __builtin_fmaf(v2[j], sw[C_W2 + j * 64 + k], s);
            v1[k] = s * lrelu_mask(f.h1[k], slope);
                FENCE();
        }
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            g0 = __builtin_fmaf(v1[j], sw[C_W1 + j * 4 + 0], g0);
            g1 = __builtin_fmaf(v1[j], sw[C_W1 + j * 4 + 1], g1);
        }
        g[n] = make_float2(g0, g1);
        uint4 mb = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            mb.x |= (MBIT(f.h1[k])) << k;
            mb.y |= (MBIT(f.h1[32 + k])) << k;
            mb.z |= (MBIT(f.h2[k])) << k;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) mb.w |= (MBIT(f.h3[k])) << k;
        masks[n] = mb;
    }
}

// pass B: one wave per batch row: ||g_b||, the per-row coefficient c_b, and the penalty sum
__global__ __launch_bounds__(64) void k_gp_rows(const float2 *__restrict__ g, int64_t B, int64_t L,
                                                float *__restrict__ rownorm, float *__restrict__ rowc,
                                                float *__restrict__ gp_sum) {
    float pen = 0.f;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        float s = 0.f;
        for (int64_t i = threadIdx.x; i < L; i += 64) {
            float2 v = g[b * L + i];
            s += v.x * v.x + v.y * v.y;
        }
        s = wave_sum(s);
        const float nrm = sqrtf(s);
        if (threadIdx.x == 0) {
            rownorm[b] = nrm;
            // d (nrm - 1)^2 / d g = 2 (nrm - 1) g / nrm; torch's norm backward gives 0 at nrm == 0
            rowc[b] = nrm > 0.f ? 2.f * (nrm - 1.f) / (nrm * (float)B) : 0.f;
            pen += (nrm - 1.f) * (nrm - 1.f);
        }
    }
    if (threadIdx.x == 0) atomicAdd(gp_sum, pen / (float)B);
}

// (left, right) columns for the penalty backward:
//   left  panel: v1[64] | v2[32] | v3[16] | dv3*D3 [16] -> 128 wide (pitch LPG)
//   right panel: gh[2] | du1[64] | du2[32] | 1.0        -> 99 wide (pitch RPG)
constexpr int LPG = 129, RPG = 101;

DEV void entry_cols_gp(int e, int &lc, int &rc) {
    lc = 0; rc = 99;  // column 99 of the right panel holds 0: bias entries and padding come out as 0
    if (e < C_B1) { int k = e % 4; if (k < 2) { lc = e / 4; rc = k; } }
    else if (e >= C_W2 && e < C_B2) { lc = 64 + (e - C_W2) / 64; rc = 2 + (e - C_W2) % 64; }
    else if (e >= C_W3 && e < C_B3) { lc = 96 + (e - C_W3) / 32; rc = 66 + (e - C_W3) % 32; }
    else if (e >= C_W4 && e < C_B4) { lc = 112 + (e - C_W4); rc = 98; }
}

__global__ __launch_bounds__(TPB) void k_gp_bwd(const float2 *__restrict__ g, const uint4 *__restrict__ masks,
                                                const float *__restrict__ rowc, int64_t N, int64_t L,
                                                const float *__restrict__ wb, float slope, float *__restrict__ slab) {
    extern __shared__ float dyn_lds[];
    float *sw = dyn_lds, *lp = sw + (C_NP + 3), *rp = lp + TPB * LPG;
    load_wb(sw, wb);
    const int tid = threadIdx.x;
    float acc[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) acc[i] = 0.f;
    const int64_t ntiles = (N + TPB - 1) / TPB;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t n = tile * TPB + tid;
        float *l = lp + tid * LPG, *r = rp + tid * RPG;
        if (n < N) {
            const uint4 mb = masks[n];
            const float c = rowc[n / L];
            const float2 gv = g[n];
            const float gh0 = c * gv.x, gh1 = c * gv.y;
            float v3[16], v2[32], v1[64];
#pragma unroll
            for (int k = 0; k < 16; ++k) v3[k] = sw[C_W4 + k] * (((mb.w >> k) & 1u) ? 1.f : slope);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) s = __builtin_fmaf(v3[j], sw[C_W3 + j * 32 + k], s);
                v2[k] = s * (((mb.z >> k) & 1u) ? 1.f : slope);
                FENCE();
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 32; ++j) s = __builtin_fmaf(v2[j], sw[C_W2 + j * 64 + k], s);
                const unsigned bit = k < 32 ? (mb.x >> k) & 1u : (mb.y >> (k - 32)) & 1u;
                v1[k] = s * (bit ? 1.f : slope);
                FENCE();
            }
            float du1[64], du2[32];
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const unsigned bit = k < 32 ? (mb.x >> k) & 1u : (mb.y >> (k - 32)) & 1u;
                du1[k] = __builtin_fmaf(sw[C_W1 + k * 4 + 1], gh1, sw[C_W1 + k * 4] * gh0) * (bit ? 1.f : slope);
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 64; ++k) s = __builtin_fmaf(sw[C_W2 + j * 64 + k], du1[k], s);
                du2[j] = s * (((mb.z >> j) & 1u) ? 1.f : slope);
                FENCE();
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 32; ++j) s = __builtin_fmaf(sw[C_W3 + k * 32 + j], du2[j], s);
                l[112 + k] = s * (((mb.w >> k) & 1u) ? 1.f : slope);
                FENCE();
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) l[k] = v1[k];
#pragma unroll
            for (int k = 0; k < 32; ++k) l[64 + k] = v2[k];
#pragma unroll
            for (int k = 0; k < 16; ++k) l[96 + k] = v3[k];
            r[0] = gh0; r[1] = gh1;
#pragma unroll
            for (int k = 0; k < 64; ++k) r[2 + k] = du1[k];
#pragma unroll
            for (int k = 0; k < 32; ++k) r[66 + k] = du2[k];
            r[98] = 1.f;
            r[99] = 0.f;
        } else {
            for (int k = 0; k < 128; ++k) l[k] = 0.f;
            for (int k = 0; k < 100; ++k) r[k] = 0.f;
        }
        __syncthreads();
        panel_accumulate<LPG, RPG>(lp, rp, tid, acc, [](int e, int &lc, int &rc) { entry_cols_gp(e, lc, rc); });
        __syncthreads();
    }
    float *row = slab + (int64_t)blockIdx.x * C_SLAB;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        int e = tid + 64 * i;
        if (e < C_NP) row[e] = acc[i];
    }
}

}  // namespace

extern "C" int dpd_critic_forward(const float *x, const float *cond, int64_t N, const float *wb, float slope,
                                  float *score, float *score_sum, void *stream) {
    DPD_REQUIRE(((x && cond) || N == 0) && wb && (score || score_sum || N == 0), DPD_ERR_INVALID_ARG, "critic_forward: null pointer");
    DPD_REQUIRE(N >= 0, DPD_ERR_INVALID_ARG, "critic_forward: negative N");
    hipStream_t st = as_stream(stream);
    if (score_sum) DPD_HIP(hipMemsetAsync(score_sum, 0, sizeof(float), st));
    if (N == 0) return DPD_OK;
    if (critic_use_mfma(N)) return critic_mfma_forward(x, cond, N, wb, slope, score, score_sum, st);
    hipLaunchKernelGGL(k_critic_fwd, dim3(grid_for(N, 256)), dim3(256), 0, st, reinterpret_cast<const float2 *>(x),
                       reinterpret_cast<const float2 *>(cond), N, wb, slope, score, score_sum);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_critic_backward(const float *x, const float *cond, int64_t N, const float *wb, float slope,
                                   const float *gscore, float gconst, float *grads, float *gx, float *score_sum,
                                   void *ws, size_t ws_bytes, void *stream) {
    DPD_REQUIRE(((x && cond) || N == 0) && wb && (grads || gx || N == 0), DPD_ERR_INVALID_ARG, "critic_backward: null pointer");
    DPD_REQUIRE(!grads || (ws && ws_bytes >= dpd_workspace_bytes(0)), DPD_ERR_WORKSPACE, "critic_backward: workspace too small");
    hipStream_t st = as_stream(stream);
    // (with weight gradients the MFMA family writes the sum from its slab reduction: nothing to clear)
    if (score_sum && !(N > 0 && grads && critic_use_mfma(N))) DPD_HIP(hipMemsetAsync(score_sum, 0, sizeof(float), st));
    if (N <= 0) {
        if (grads) DPD_HIP(hipMemsetAsync(grads, 0, C_NP * sizeof(float), st));
        return DPD_OK;
    }
    if (critic_use_mfma(N))
        return critic_mfma_backward(x, cond, N, wb, slope, gscore, gconst, grads, gx, score_sum,
                                    reinterpret_cast<float *>(grads || ws_bytes >= 64 ? ws : nullptr), st);
    const int grid = grid_for(N, TPB, 1024);
    float *slab = reinterpret_cast<float *>(ws);
    constexpr size_t lds_bytes = (size_t)(C_NP + 3 + TPB * LP + TPB * RP) * sizeof(float);
    static PerDeviceOnce attr_once;   // per device, not per process
    int attr_dev;
    if (attr_once.first_use(&attr_dev)) {
        DPD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_critic_bwd<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        DPD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_critic_bwd<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_once.mark(attr_dev);
    }
    if (!grads) {   // input gradient only (generator update): no panel reduction, no slab
        hipLaunchKernelGGL(k_critic_bwd<false>, dim3(grid_for(N, TPB, 4096)), dim3(TPB), lds_bytes, st,
                           reinterpret_cast<const float2 *>(x), reinterpret_cast<const float2 *>(cond), N, wb, slope,
                           gscore, gconst, reinterpret_cast<float2 *>(gx), slab, score_sum);
        DPD_LAUNCH_CHECK();
        return DPD_OK;
    }
    hipLaunchKernelGGL(k_critic_bwd<true>, dim3(grid), dim3(TPB), lds_bytes, st, reinterpret_cast<const float2 *>(x),
                       reinterpret_cast<const float2 *>(cond), N, wb, slope, gscore, gconst,
                       reinterpret_cast<float2 *>(gx), slab, score_sum);
    DPD_LAUNCH_CHECK();
    return gen_reduce_slab_generic(slab, grid, C_SLAB, C_NP, grads, nullptr, 1.f, st);
}

extern "C" int dpd_critic_gp(const float *real, const float *fake, const float *cond, const float *alpha, int64_t B,
                             int64_t L, const float *wb, float slope, float *gp, float *rownorm, float *grads,
                             void *ws, size_t ws_bytes, void *stream) {
    DPD_REQUIRE(((real && fake && cond && alpha && rownorm) || B == 0) && wb && gp && grads, DPD_ERR_INVALID_ARG, "critic_gp: null pointer");
    DPD_REQUIRE(B >= 0 && L >= 1, DPD_ERR_INVALID_ARG, "critic_gp: bad shape B=%lld L=%lld", (long long)B, (long long)L);
    const int64_t N = B * L;
    // workspace: slab | g [N,2] | masks [N] uint4 | rowc [B]
    const size_t slab_bytes = (size_t)1024 * C_SLAB * sizeof(float);
    const size_t need = slab_bytes + (size_t)N * (8 + 16) + (size_t)B * 4 + 256;
    DPD_REQUIRE(ws && ws_bytes >= need, DPD_ERR_WORKSPACE, "critic_gp: workspace of %zu bytes needed, %zu given", need, ws_bytes);
    hipStream_t st = as_stream(stream);
    const bool mfma_family = N > 0 && critic_use_mfma(N) && (L == 1 || (L & 15) == 0);
    if (!mfma_family) DPD_HIP(hipMemsetAsync(gp, 0, sizeof(float), st));      // (the MFMA family writes gp, it does not add to it)
    if (N == 0) {
        DPD_HIP(hipMemsetAsync(grads, 0, C_NP * sizeof(float), st));
        return DPD_OK;
    }
    if (mfma_family) {
        DPD_REQUIRE(ws_bytes >= critic_mfma_gp_ws_bytes(B, L, ws_bytes), DPD_ERR_WORKSPACE, "critic_gp: workspace too small");
        return critic_mfma_gp(real, fake, cond, alpha, B, L, wb, slope, gp, rownorm, grads, ws, ws_bytes, st);
    }
    char *base = reinterpret_cast<char *>(ws);
    float *slab = reinterpret_cast<float *>(base);
    uint4 *masks = reinterpret_cast<uint4 *>(base + slab_bytes);
    float2 *g = reinterpret_cast<float2 *>(base + slab_bytes + (size_t)N * 16);
    float *rowc = reinterpret_cast<float *>(base + slab_bytes + (size_t)N * 24);
    hipLaunchKernelGGL(k_gp_grad, dim3(grid_for(N, 256)), dim3(256), 0, st, reinterpret_cast<const float2 *>(real),
                       reinterpret_cast<const float2 *>(fake), reinterpret_cast<const float2 *>(cond), alpha, N, L, wb,
                       slope, g, masks);
    DPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gp_rows, dim3(grid_for(B, 1, 2048)), dim3(64), 0, st, g, B, L, rownorm, rowc, gp);
    DPD_LAUNCH_CHECK();
    const int grid = grid_for(N, TPB, 1024);
    constexpr size_t lds_bytes = (size_t)(C_NP + 3 + TPB * LPG + TPB * RPG) * sizeof(float);
    static PerDeviceOnce attr_once;
    int attr_dev;
    if (attr_once.first_use(&attr_dev)) {
        DPD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gp_bwd),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_once.mark(attr_dev);
    }
    hipLaunchKernelGGL(k_gp_bwd, dim3(grid), dim3(TPB), lds_bytes, st, g, masks, rowc, N, L, wb, slope, slab);
    DPD_LAUNCH_CHECK();
    return gen_reduce_slab_generic(slab, grid, C_SLAB, C_NP, grads, nullptr, 1.f, st);
}
