// Critic 4 -> 64 -> 32 -> 16 -> 1 on fp32 MFMA (v_mfma_f32_16x16x4_f32): forward, backward and the
// gradient penalty with its double backward.  Semantics: reference models/discriminator.py:103-191
// (see critic.hip for the formulas).
//
// Geometry as in the generator kernels: tiles of 16 samples, lane l = (g = l >> 4, j = l & 15),
// every product computed transposed (D = W * X^T) so that a 16x16 result leaves lane (g, j) holding
// neurons 4g..4g+3 of sample j, which is the B-operand layout of the next layer.  The A operands
// (weights in MFMA lane order) do not fit registers next to the gradient accumulators, so the
// workgroup keeps them in an LDS image, four k-steps per ds_read_b128.  The single-row products
// (score, input gradient) use a replicated-row image so that the result lands in register 0 (0, 1)
// of every lane group.  Weight gradients contract over samples: left and right factor tiles are
// transposed through XOR-swizzled wave-private LDS tiles in three phases that share one buffer.
#include "common.h"

int gen_reduce_slab_generic(const float *slab, int n_partials, int row_len, int n_out, float *grads, float *extra,
                            float extra_scale, hipStream_t st);

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// experiment switches (timing only, results are wrong): ABL_NO_MFMA drops every matrix instruction but keeps its
// data dependencies, ABL_NO_SPLIT leaves one instruction of the three-way split, ABL_NO_LDSOPS takes the weight
// operands from registers instead of the LDS image
#ifdef ABL_NO_MFMA
DEV f32x4 mfma4(float a, float b, f32x4 c) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return c; }
#else
DEV f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
#endif
// sixteen independent 4x4 outer products: lane 4b + i of `a` times lane 4b + j of `b` -> register i of lane 4b + j
#ifdef ABL_NO_MFMA
DEV f32x4 mfma_blocks(float a, float b, f32x4 c) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return c; }
#else
DEV f32x4 mfma_blocks(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
#endif

constexpr int WAVES = 4;
// waves per workgroup of the weight-gradient kernels: one operand image serves all of them, so a large
// workgroup is what lets a third wave per SIMD fit into the LDS (12: 146 KiB, one workgroup per CU)
#ifndef CRITIC_BF16X3
#define CRITIC_BF16X3 1
#endif
#ifndef WG_WAVES
#define WG_WAVES (CRITIC_BF16X3 ? 8 : 4)   // the bf16 image is 48 KiB: eight waves share one per CU
#endif
constexpr int WG_PER_CU = WG_WAVES == 4 ? 2 : 1;
constexpr int NOGRAD_PER_CU = CRITIC_BF16X3 ? 3 : 4;   // workgroups per CU of the kernels without a transposition buffer (LDS)
// operand image: quads of four consecutive k-steps per lane
#if CRITIC_BF16X3
// The hidden-layer products run on the bf16 matrix cores with both operands split three ways into bf16
// (x = hi + mid + lo, each the bf16 rounding of what the previous terms leave; six products hh, hm, mh, hl,
// lh, mm accumulated in fp32, small terms first): the error against fp64 is that of the fp32 MFMA, six
// v_mfma_f32_16x16x32_bf16 replace eight v_mfma_f32_16x16x4_f32 at 8 ns instead of 14.5 ns each, and VALU
// work issues in their shadow, which it does not for the fp32 MFMA.  A 16-byte image entry holds eight
// bf16 of one term: k = 8 * lane group + e, where e < 4 is neuron 4g + e of one 16-row result tile and
// e >= 4 neuron 4g + e - 4 of the next tile (exactly what a lane holds after the previous layer).
constexpr int Q_A1 = 0;     // [t]            e = t : W1[16t + i][g]  (FC1, K = 4, stays on the fp32 MFMA)
constexpr int Q_W1C = 1;    // [c][t]         e = r : W1[16t + 4g + r][c], c = 0, 1 (the two x columns; VALU operand)
constexpr int Q_B1 = 9;     // [t]            e = r : b1[16t + 4g + r]
constexpr int Q_B2 = 13;    // [u]            e = r : b2[16u + 4g + r]
constexpr int Q_B3 = 15;    //                e = r : b3[4g + r]
constexpr int Q_B4 = 16;    //                e = 0 : b4, else 0
constexpr int Q_W4 = 17;    //                e = r : W4[4g + r] (VALU operand)
constexpr int QB_A2 = 18;   // [u][q][term]   W2[16u + i][neuron(q, g, e)], neuron = 16(2q + (e >> 2)) + 4g + (e & 3)
constexpr int QB_A3 = 30;   // [term]         W3[i][16(e >> 2) + 4g + (e & 3)]
constexpr int QB_A2T = 33;  // [t][term]      W2[16(e >> 2) + 4g + (e & 3)][16t + i]
constexpr int QB_A3T = 45;  // [u][pair]      W3[4g + (e & 3)][16u + i], term h for e < 4, l (pair 0) or m (pair 1) for e >= 4:
                            //                K = 32 = the 16 layer-3 neurons in two terms (see chain_top)
// FC1 on the bf16 matrix cores as well (CRITIC_FC1_BF16): K = 4 inputs, one per lane group, so the eight k-slots of a lane
// group hold the six products of its input -- weight terms [h, h, m, h, l, m] against input terms [h, m, h, l, h, m] -- and
// the two spare slots of groups 0 and 1 carry the three terms of the bias against constant ones: one MFMA per 16-row tile
// with a zero start replaces one fp32 MFMA (32 cycles during which the SIMD issues nothing else) and the bias quad.
// Measured: weight-gradient pass -0.8 %, input-gradient kernel +5 %, one-pass penalty +2.7 % (four spilled registers): off.
#ifndef CRITIC_FC1_BF16
#define CRITIC_FC1_BF16 0
#endif
constexpr int QB_A1B = 49;  // [t]            FC1 weights and bias of row 16t + i in the slot order above
constexpr int NQUAD = CRITIC_FC1_BF16 ? 53 : 49;
#else
constexpr int Q_A1 = 0;     // [t]            e = t : W1[16t + i][g]
constexpr int Q_A2 = 1;     // [u][t]         e = r : W2[16u + i][16t + 4g + r]
constexpr int Q_A3 = 9;     // [u]            e = r : W3[i][16u + 4g + r]
constexpr int Q_A3T = 11;   // [u]            e = r : W3[4g + r][16u + i]
constexpr int Q_A2T = 13;   // [t][u]         e = r : W2[16u + 4g + r][16t + i]
constexpr int Q_W1C = 21;   // [c][t]         e = r : W1[16t + 4g + r][c], c = 0, 1 (the two x columns; VALU operand)
constexpr int Q_B1 = 29;    // [t]            e = r : b1[16t + 4g + r]
constexpr int Q_B2 = 33;    // [u]            e = r : b2[16u + 4g + r]
constexpr int Q_B3 = 35;    //                e = r : b3[4g + r]
constexpr int Q_B4 = 36;    //                e = 0 : b4, else 0
constexpr int Q_W4 = 37;    //                e = r : W4[4g + r] (VALU operand)
constexpr int NQUAD = 38;
#endif
constexpr int IMG_WORDS = NQUAD * 64 * 4;          // 9728 (10752 with the FC1 quads)
// The weight gradients of the two inner layers (sums over the samples of a tile of h x d) on the bf16 matrix
// cores as well: the three bf16 terms of both factors exist anyway (the forward and the chain split them), so
// they are parked in LDS as [sample][neuron] slots and read back with the transposing ds_read_b64_tr_b16,
// which hands a lane four consecutive samples of one neuron -- the operand layout of a product over samples.
#ifndef CRITIC_OUTER_BF16
#define CRITIC_OUTER_BF16 CRITIC_BF16X3
#endif
static_assert(!CRITIC_OUTER_BF16 || CRITIC_BF16X3, "the bf16 outer products take the split terms of the bf16 hidden layers");
// per-wave transposition buffer
constexpr int P64 = 80, P32 = 48, P16 = 16;        // row pitches: 16 mod 32 -> rows s, s+1 hit different bank halves
#if CRITIC_OUTER_BF16
// [p1 tile: 16 x P64 words, which the term slots of h2 / d2 (3) and d3 (2) share in time] [input tile] [h1 term slots (6)]
constexpr int SLOT_B = 1024;                        // bytes of a slot: 16 samples x (two 16-neuron tiles x 4 neurons of a lane group) bf16
constexpr int LEFT_W = 0, CONST_W = 16 * P64, H1_W = CONST_W + 16 * P16;
constexpr int WAVE_WORDS = H1_W + 6 * SLOT_B / 4;  // 3072
constexpr int S_H2 = 0, S_D2 = 0, S_D3 = 3 * SLOT_B, S_H1 = 4 * H1_W;   // byte offsets of the slot groups in the wave's buffer
static_assert(5 * SLOT_B <= 4 * 16 * P64, "the h2 / d2 / d3 slots lie inside the p1 tile");
#else
constexpr int LEFT_W = 0, RIGHT_W = 16 * P64, CONST_W = RIGHT_W + 16 * P32;
constexpr int WAVE_WORDS = CONST_W + 16 * P16;     // 2304
#endif
constexpr int LDS_WORDS = IMG_WORDS + WG_WAVES * WAVE_WORDS;
static_assert(LDS_WORDS >= WG_WAVES * C_SLAB, "gradient rows are combined in the same LDS");

DEV int swz(int s) { return 4 * ((s >> 1) & 3); }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// x = h + m + l to 24 bits: each term is the bf16 rounding of what the previous terms leave
DEV void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

#if CRITIC_BF16X3
DEV void build_image(float *img, const float *__restrict__ wb) {
    for (int idx = threadIdx.x; idx < NQUAD * 64; idx += blockDim.x) {
        const int q = idx >> 6, lane = idx & 63, i = lane & 15, g = lane >> 4;
        if (q < QB_A2) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x;
                if (q == Q_A1) x = wb[C_W1 + (16 * e + i) * 4 + g];
                else if (q < Q_B1) { int c = (q - Q_W1C) >> 2, t = (q - Q_W1C) & 3; x = wb[C_W1 + (16 * t + 4 * g + e) * 4 + c]; }
                else if (q < Q_B2) { int t = q - Q_B1; x = wb[C_B1 + 16 * t + 4 * g + e]; }
                else if (q < Q_B3) { int u = q - Q_B2; x = wb[C_B2 + 16 * u + 4 * g + e]; }
                else if (q == Q_B3) x = wb[C_B3 + 4 * g + e];
                else if (q == Q_B4) x = e == 0 ? wb[C_B4] : 0.f;
                else x = wb[C_W4 + 4 * g + e];
                v[e] = x;
            }
            reinterpret_cast<f32x4 *>(img)[idx] = v;
        } else {
            int term;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int n = 16 * (e >> 2) + 4 * g + (e & 3);     // neuron inside a pair of 16-row tiles
                float x;
                if (q < QB_A3) { int k = q - QB_A2; term = k % 3; int u = k / 6, qq = (k / 3) & 1; x = wb[C_W2 + (16 * u + i) * 64 + 32 * qq + n]; }
                else if (q < QB_A2T) { term = q - QB_A3; x = wb[C_W3 + i * 32 + n]; }
                else if (q < QB_A3T) { int k = q - QB_A2T; term = k % 3; int t = k / 3; x = wb[C_W2 + n * 64 + 16 * t + i]; }
                else if (q < QB_A1B) { int k = q - QB_A3T; term = e < 4 ? 0 : ((k & 1) ? 1 : 2); x = wb[C_W3 + (4 * g + (e & 3)) * 32 + 16 * (k >> 1) + i]; }
                else {                                             // FC1: weight terms h, h, m, h, l, m; bias terms in the spare slots
                    const int t = q - QB_A1B;
                    if (e < 6) { term = e == 2 || e == 5 ? 1 : (e == 4 ? 2 : 0); x = wb[C_W1 + (16 * t + i) * 4 + g]; }
                    else { term = g == 0 ? e - 6 : 2; x = (g == 0 || (g == 1 && e == 6)) ? wb[C_B1 + 16 * t + i] : 0.f; }
                }
                __bf16 h, m, l;
                split3(x, h, m, l);
                v[e] = term == 0 ? h : (term == 1 ? m : l);
            }
            reinterpret_cast<bf16x8 *>(img)[idx] = v;
        }
    }
    __syncthreads();
}
#else
DEV void build_image(float *img, const float *__restrict__ wb) {
    for (int idx = threadIdx.x; idx < NQUAD * 64; idx += blockDim.x) {
        const int q = idx >> 6, lane = idx & 63, i = lane & 15, g = lane >> 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = 0.f;
            if (q == Q_A1) x = wb[C_W1 + (16 * e + i) * 4 + g];
            else if (q < Q_A3) { int u = (q - Q_A2) >> 2, t = (q - Q_A2) & 3; x = wb[C_W2 + (16 * u + i) * 64 + 16 * t + 4 * g + e]; }
            else if (q < Q_A3T) { int u = q - Q_A3; x = wb[C_W3 + i * 32 + 16 * u + 4 * g + e]; }
            else if (q < Q_A2T) { int u = q - Q_A3T; x = wb[C_W3 + (4 * g + e) * 32 + 16 * u + i]; }
            else if (q < Q_W1C) { int t = (q - Q_A2T) >> 1, u = (q - Q_A2T) & 1; x = wb[C_W2 + (16 * u + 4 * g + e) * 64 + 16 * t + i]; }
            else if (q < Q_B1) { int c = (q - Q_W1C) >> 2, t = (q - Q_W1C) & 3; x = wb[C_W1 + (16 * t + 4 * g + e) * 4 + c]; }
            else if (q < Q_B2) { int t = q - Q_B1; x = wb[C_B1 + 16 * t + 4 * g + e]; }
            else if (q < Q_B3) { int u = q - Q_B2; x = wb[C_B2 + 16 * u + 4 * g + e]; }
            else if (q == Q_B3) x = wb[C_B3 + 4 * g + e];
            else if (q == Q_B4) x = e == 0 ? wb[C_B4] : 0.f;
            else x = wb[C_W4 + 4 * g + e];
            v[e] = x;
        }
        reinterpret_cast<f32x4 *>(img)[idx] = v;
    }
    __syncthreads();
}
#endif

DEV f32x4 quad(const float *img, int q, int lane) { return reinterpret_cast<const f32x4 *>(img)[q * 64 + lane]; }

struct Acts {
    f32x4 h1[4], h2[2], h3;   // post-activation
    f32x4 m1[4], m2[2], m3;   // LeakyReLU derivative (1 or slope), from the same comparison as the activation
    unsigned bits;            // BITS: the comparisons as one word (h1: bits 0..15, h2: 16..23, h3: 24..27)
    float score;
};

DEV f32x4 lrelu4(f32x4 z, float slope) {
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = z[r] > 0.f ? z[r] : z[r] * slope;
    return h;
}
// activation and its derivative from one comparison per element: h = z * m is z itself or z * slope
template <bool BITS>
DEV void act4(f32x4 z, float slope, f32x4 &h, f32x4 &m, unsigned &bits, int k0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool pos = z[r] > 0.f;
        m[r] = pos ? 1.f : slope;
        if (BITS) bits |= pos ? (1u << (k0 + r)) : 0u;
    }
    h = z * m;
}
DEV f32x4 mask4(f32x4 h, float slope) {
    f32x4 m;
#pragma unroll
    for (int r = 0; r < 4; ++r) m[r] = h[r] > 0.f ? 1.f : slope;
    return m;
}

// sum of a per-lane value over the four lane groups (same j), left in every lane: two row swaps.  The
// swap results pass through an empty asm: this compiler otherwise folds the second result of a swap
// into the first once they are reinterpreted as floats.
DEV float sum_over_groups(float v) {
    const unsigned b = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);       // rows (0,0,2,2), (1,1,3,3)
    unsigned u0 = r[0], u1 = r[1];
    asm volatile("" : "+v"(u0), "+v"(u1));
    const float s = __builtin_bit_cast(float, u0) + __builtin_bit_cast(float, u1);
    const unsigned c = __builtin_bit_cast(unsigned, s);
    const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);       // lower half, upper half
    unsigned w0 = q[0], w1 = q[1];
    asm volatile("" : "+v"(w0), "+v"(w1));
    return __builtin_bit_cast(float, w0) + __builtin_bit_cast(float, w1);
}

// ---- operand prefetch ---------------------------------------------------------------------------------
// The weights live in the LDS image and are read per tile.  Every MFMA group takes its operand quads as
// arguments that the caller (or the previous group, through the `pre` callable) has read one group
// earlier, so the LDS latency of a read is covered by the MFMAs of the group before it.  The fences keep
// the compiler from sinking those reads down to their first use.
#define FENCE() __builtin_amdgcn_sched_barrier(0)
// Issue priority of the wave by the kind of code it is in.  Two waves share a SIMD, and the arbiter serves the older
// one first: a wave with a matrix instruction waiting for the pipe keeps the vector issue port, so the younger wave's
// VALU work does not issue beside the older wave's burst of MFMAs (scratch/ubench/cross_wave_nop.hip: a wave of
// back-to-back bf16 MFMAs and a younger wave of VALU work take the sum of their times, 1.25 + 1.45 -> 2.66 ms; with
// the VALU wave at a higher priority 1.94 ms whichever is older, which is what the issue port allows).  So a wave runs
// its stretches without matrix instructions at priority CRITIC_VPRIO and its stretches with them at 0.  s_setprio is
// a scheduling boundary for the compiler, hence every switch sits next to a fence that is there anyway.
#ifndef CRITIC_VPRIO
#define CRITIC_VPRIO 2
#endif
#if CRITIC_VPRIO
#define PRIO_V() __builtin_amdgcn_s_setprio(CRITIC_VPRIO)
#define PRIO_M() __builtin_amdgcn_s_setprio(0)
#else
#define PRIO_V()
#define PRIO_M()
#endif
// experiment hook: ts == NULL in every product build (the code folds away); a timing build passes an
// array whose slot 15 holds the last stamp
#define TS(k)                                                      \
    if (ts) {                                                      \
        FENCE();                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        ts[k] += now_ - ts[15];                                    \
        ts[15] = now_;                                             \
        FENCE();                                                   \
    }
struct NoPre {
    DEV void operator()() const {}
};
struct InOps {         // W1[:, 0:2] per neuron, VALU operands
    f32x4 w0[4], w1[4];
};
DEV void load_ops(InOps &o, const float *img, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        o.w0[t] = quad(img, Q_W1C + t, lane);
        o.w1[t] = quad(img, Q_W1C + 4 + t, lane);
    }
}

#if CRITIC_BF16X3
#ifdef ABL_NO_LDSOPS
DEV bf16x8 bquad(const float *img, int q, int lane) {
    f32x4 v = {(float)q, (float)lane, 1.f, 2.f};
    asm volatile("" : "+v"(v));
    return __builtin_bit_cast(bf16x8, v);
}
#else
DEV bf16x8 bquad(const float *img, int q, int lane) { return reinterpret_cast<const bf16x8 *>(img)[q * 64 + lane]; }
#endif
struct B3 {            // eight values per lane, split three ways
    bf16x8 h, m, l;
};
DEV void load_b3(B3 &o, const float *img, int q, int lane) {
    o.h = bquad(img, q, lane);
#ifdef ABL_ONE_TERM_READ      // experiment (timing only, results are wrong): one of the three operand reads
    o.m = o.h;
    o.l = o.h;
    asm volatile("" : "+v"(o.m), "+v"(o.l));
    return;
#endif
    o.m = bquad(img, q + 1, lane);
    o.l = bquad(img, q + 2, lane);
}
// Two fp32 values -> their three bf16 terms, packed in pairs.  hi and mid are the upper 16 bits of the
// value and of what it leaves (truncation: the mask gives the term as a float at once, the subtraction is
// exact, and a byte permute packs the pair), lo is the bf16 rounding of the rest: 9 VALU instructions per
// pair where round-to-nearest terms need 11.  x = hi + mid + lo to 24 bits either way.
DEV void split_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
#ifdef ABL_NO_SPLIT
    m = h;
    l = h;
    return;
#endif
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u), q1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
    l = __builtin_bit_cast(unsigned, bf16x2{(__bf16)q0, (__bf16)q1});
}
// the eight values a lane holds of a pair of 16-row result tiles -> three bf16 operands
DEV void split8(B3 &o, const f32x4 &x0, const f32x4 &x1) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned h[4], m[4], l[4];
    split_pair(x0[0], x0[1], h[0], m[0], l[0]);
    split_pair(x0[2], x0[3], h[1], m[1], l[1]);
    split_pair(x1[0], x1[1], h[2], m[2], l[2]);
    split_pair(x1[2], x1[3], h[3], m[3], l[3]);
    o.h = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
    o.m = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
    o.l = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
}
#ifdef ABL_NO_MFMA
DEV f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return c; }
#else
DEV f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif
// c + a x b to fp32 accuracy, small terms first.  CRITIC_ONE_CHAIN (default): the six products of a tile go into one
// accumulator, in rising order of magnitude -- the matrix pipe runs a dependent accumulation chain at the rate of
// independent ones, so a second accumulator per tile only costs its registers and the additions that merge it.
// CRITIC_ONE_CHAIN = 0: two chains of three products each (lo: mm, lh, hl; hi: mh, hm, hh), merged by the caller.
#ifndef CRITIC_ONE_CHAIN
#define CRITIC_ONE_CHAIN 1
#endif
DEV void mfma6(f32x4 &lo, f32x4 &hi, const B3 &a, const B3 &b) {
#if CRITIC_ONE_CHAIN
    hi = mfma_bf(a.m, b.m, hi);
    hi = mfma_bf(a.l, b.h, hi);
    hi = mfma_bf(a.h, b.l, hi);
    hi = mfma_bf(a.m, b.h, hi);
    hi = mfma_bf(a.h, b.m, hi);
    hi = mfma_bf(a.h, b.h, hi);
#else
    lo = mfma_bf(a.m, b.m, lo);
    hi = mfma_bf(a.m, b.h, hi);
    lo = mfma_bf(a.l, b.h, lo);
    hi = mfma_bf(a.h, b.m, hi);
    lo = mfma_bf(a.h, b.l, lo);
    hi = mfma_bf(a.h, b.h, hi);
#endif
}
// two tiles that share the right operand, their chains interleaved
DEV void mfma6x2(f32x4 &lo0, f32x4 &hi0, f32x4 &lo1, f32x4 &hi1, const B3 &a0, const B3 &a1, const B3 &b) {
#if CRITIC_ONE_CHAIN
    hi0 = mfma_bf(a0.m, b.m, hi0);
    hi1 = mfma_bf(a1.m, b.m, hi1);
    hi0 = mfma_bf(a0.l, b.h, hi0);
    hi1 = mfma_bf(a1.l, b.h, hi1);
    hi0 = mfma_bf(a0.h, b.l, hi0);
    hi1 = mfma_bf(a1.h, b.l, hi1);
    hi0 = mfma_bf(a0.m, b.h, hi0);
    hi1 = mfma_bf(a1.m, b.h, hi1);
    hi0 = mfma_bf(a0.h, b.m, hi0);
    hi1 = mfma_bf(a1.h, b.m, hi1);
    hi0 = mfma_bf(a0.h, b.h, hi0);
    hi1 = mfma_bf(a1.h, b.h, hi1);
#else
    mfma6(lo0, hi0, a0, b);
    mfma6(lo1, hi1, a1, b);
#endif
}

struct FrontOps {      // FC1 weights and accumulator start (b1), FC2 accumulator start (b2), FC2 weights of the first k-step
    f32x4 a1, b1[4], b2[2];
    bf16x8 a1b[4];     // CRITIC_FC1_BF16: FC1 weights and bias as one bf16 operand per row tile (a1, b1 unused then)
    B3 w[2];
};
// one fp32 value per lane -> the B operand of the FC1 product: terms [h, m, h, l, h, m] and `ones` in the two spare slots
// (bf16 ones where the A operand carries bias terms: 0x3f803f80 in lane group 0, 0x00003f80 in group 1; 0 for a product
// without bias)
DEV bf16x8 fc1_operand(float x, unsigned ones) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const float r1 = x - __builtin_bit_cast(float, u & 0xffff0000u);
    const unsigned v = __builtin_bit_cast(unsigned, r1);
    const float r2 = r1 - __builtin_bit_cast(float, v & 0xffff0000u);
    const unsigned hm = __builtin_amdgcn_perm(v, u, 0x07060302u), hl = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r2), u, 0x07060302u);
    return __builtin_bit_cast(bf16x8, u32x4{hm, hl, hm, ones});
}
DEV unsigned fc1_ones(int lane) { return (lane >> 4) == 0 ? 0x3f803f80u : ((lane >> 4) == 1 ? 0x00003f80u : 0u); }
struct Fc3Ops {        // FC3 accumulator start (b3) and weights; W4 and b4 for the VALU last layer
    f32x4 b3, w4;
    float b4;          // read with the other operands, one MFMA group ahead: a read at its use would expose the LDS latency
    B3 w;
};
struct TopOps {        // W3^T, 16-row tile u: [h|l] in t[2u], [h|m] in t[2u + 1]
    bf16x8 t[4];
};
struct BotOps {        // W2^T, row tiles 0 and 1
    B3 w[2];
};
// The fp32 operands of a tile that are not weights of a matrix product with split terms -- the FC1 weights, the three
// bias quads that start the accumulators, W4 and b4 -- stay in registers for the life of the wave (CRITIC_RESIDENT, 44
// registers): an LDS read of 16 bytes per lane costs the SIMD about 15 cycles whoever issues it (measured with extra
// reads per tile), and these were nine of the forty operand reads of a tile.  load_resident() once, load_ops() per tile.
#ifndef CRITIC_RESIDENT
#define CRITIC_RESIDENT 1
#endif
DEV void load_resident(FrontOps &o, const float *img, int lane) {
    if (!CRITIC_FC1_BF16) {
        o.a1 = quad(img, Q_A1, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) o.b1[t] = quad(img, Q_B1 + t, lane);
    }
    o.b2[0] = quad(img, Q_B2, lane);
    o.b2[1] = quad(img, Q_B2 + 1, lane);
}
DEV void load_ops(FrontOps &o, const float *img, int lane) {
    if (!CRITIC_RESIDENT) load_resident(o, img, lane);
    if (CRITIC_FC1_BF16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) o.a1b[t] = bquad(img, QB_A1B + t, lane);
    }
    load_b3(o.w[0], img, QB_A2, lane);
    load_b3(o.w[1], img, QB_A2 + 6, lane);
}
DEV void load_resident(Fc3Ops &o, const float *img, int lane) {
    o.b3 = quad(img, Q_B3, lane);
    o.w4 = quad(img, Q_W4, lane);
    o.b4 = img[(Q_B4 * 64 + lane) * 4];
}
DEV void load_ops(Fc3Ops &o, const float *img, int lane) {
    if (!CRITIC_RESIDENT) load_resident(o, img, lane);
    load_b3(o.w, img, QB_A3, lane);
}
DEV void load_ops(TopOps &o, const float *img, int lane) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o.t[k] = bquad(img, QB_A3T + k, lane);
}
DEV void load_ops(BotOps &o, const float *img, int lane) {
    load_b3(o.w[0], img, QB_A2T, lane);
    load_b3(o.w[1], img, QB_A2T + 3, lane);
}

// z0, z1 += W2 (rows 0..15 | 16..31) x src, src = 64 values per sample in four quads, two k-steps of 32.
// ACT: src holds FC1 pre-activations; LeakyReLU is applied on the way in.  pre() runs before the second
// k-step.  The splits and activations of the second step sit in the shadow of the first step's MFMAs.
struct NoKeep {        // the split terms of a factor are not wanted
    DEV void operator()(int, const B3 &) const {}
    DEV void operator()(const B3 &) const {}
    DEV void operator()(const unsigned (&)[2], const unsigned (&)[2], const unsigned (&)[2]) const {}
};
template <bool ACT, bool BITS, class Pre, class Keep = NoKeep>
DEV void fc2_block(f32x4 &z0, f32x4 &z1, f32x4 (&src)[4], const B3 (&w)[2], f32x4 (&m)[4], unsigned &bits,
                   const float *img, int lane, float slope, Pre pre, unsigned long long *ts = nullptr, Keep keep = Keep()) {
    if (ACT) {
        act4<BITS>(src[0], slope, src[0], m[0], bits, 0);
        act4<BITS>(src[1], slope, src[1], m[1], bits, 4);
    }
    B3 n[2], b;
    load_b3(n[0], img, QB_A2 + 3, lane);
    load_b3(n[1], img, QB_A2 + 9, lane);
    split8(b, src[0], src[1]);
    keep(0, b);
    FENCE();
    PRIO_M();
    f32x4 y0 = {0.f, 0.f, 0.f, 0.f}, y1 = {0.f, 0.f, 0.f, 0.f};
    mfma6x2(y0, z0, y1, z1, w[0], w[1], b);
    if (ACT) {
        act4<BITS>(src[2], slope, src[2], m[2], bits, 8);
        act4<BITS>(src[3], slope, src[3], m[3], bits, 12);
    }
    B3 b2;
    split8(b2, src[2], src[3]);
    keep(1, b2);
    TS(2)
    FENCE();
    pre();
    FENCE();
    mfma6x2(y0, z0, y1, z1, n[0], n[1], b2);
    PRIO_V();
    FENCE();
#if !CRITIC_ONE_CHAIN
    z0 += y0;
    z1 += y1;
#endif
    TS(3)
}

// forward of one tile; `in` = input component g of sample j (x0, x1, c0, c1 by lane group).
template <bool BITS = false, class Pre, class Keep = NoKeep>
DEV void forward_front(Acts &a, const float *img, int lane, float in, float slope, const FrontOps &o, Pre pre,
                       unsigned long long *ts = nullptr, Keep keep = Keep()) {
    a.bits = 0u;
    TS(0)
    if (CRITIC_FC1_BF16) {
        const bf16x8 b = fc1_operand(in, fc1_ones(lane));
#pragma unroll
        for (int t = 0; t < 4; ++t) a.h1[t] = mfma_bf(o.a1b[t], b, f32x4{0.f, 0.f, 0.f, 0.f});
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) a.h1[t] = mfma4(o.a1[t], in, o.b1[t]);
    }
    TS(1)
    f32x4 z0 = o.b2[0], z1 = o.b2[1];
    fc2_block<true, BITS>(z0, z1, a.h1, o.w, a.m1, a.bits, img, lane, slope, pre, ts, keep);
    act4<BITS>(z0, slope, a.h2[0], a.m2[0], a.bits, 16);
    act4<BITS>(z1, slope, a.h2[1], a.m2[1], a.bits, 20);
    TS(6)
}
// y = W3 x (16 outputs from the 32 values of a tile pair), on top of `start`
template <class Keep = NoKeep>
DEV f32x4 fc3_product(const f32x4 &x0, const f32x4 &x1, const B3 &w, f32x4 start, Keep keep = Keep()) {
    B3 b;
    split8(b, x0, x1);
    keep(b);
    f32x4 lo = {0.f, 0.f, 0.f, 0.f};
    PRIO_M();
    mfma6(lo, start, w, b);
    PRIO_V();
#if CRITIC_ONE_CHAIN
    return start;
#else
    return start + lo;
#endif
}
// FC3 and the last layer (16 -> 1), which runs on the VALU: four FMAs per lane and a sum over the lane groups
// SCORE = false: the caller wants the activations and masks only (the gradient penalty)
template <bool BITS = false, bool SCORE = true, class Pre, class Keep = NoKeep>
DEV void forward_back(Acts &a, const float *img, int lane, float slope, const Fc3Ops &o, Pre pre,
                      unsigned long long *ts = nullptr, Keep keep = Keep()) {
    TS(7)
    pre();
    FENCE();
    const f32x4 y = fc3_product(a.h2[0], a.h2[1], o.w, o.b3, keep);
    FENCE();
    TS(8)
    act4<BITS>(y, slope, a.h3, a.m3, a.bits, 24);
    if (!SCORE) return;
    float s = o.w4[0] * a.h3[0];
#pragma unroll
    for (int r = 1; r < 4; ++r) s = __builtin_fmaf(o.w4[r], a.h3[r], s);
    a.score = sum_over_groups(s) + o.b4;
    TS(9)
}

// back-propagation of an upstream vector through the three hidden layers, in two steps:
//   top[r] (16 values on neurons 4g + r of layer 3, already masked) -> d2[2] (layer 2, masked) -> d1[4] (layer 1, masked)
// The first step contracts over the 16 layer-3 neurons, four per lane group (k = 4g + e is what a lane holds).
// The MFMAs of one wave and the VALU instructions of another wave of the same SIMD do not overlap on this part
// (scratch/ubench/cross_wave_overlap.hip: their times add up), so a burst of MFMAs hides nothing unless the
// wave itself has independent VALU work in the same stretch of code.  Hence the order here: the first 16-row
// tile's chain of six MFMAs, then the second tile's six with the masking and the three-way split of the first
// tile's result beside them; the split of the second tile's result sits beside over() -- the caller's next
// MFMAs that do not depend on it (the FC3 weight gradient) -- and sd2 hands the terms on to chain_bottom.
struct NoOp {
    DEV void operator()() const {}
};
template <class Pre, class Keep = NoKeep, class Over = NoOp>
DEV void chain_top(f32x4 top, const f32x4 (&m2)[2], f32x4 (&d2)[2], B3 &sd2, const TopOps &o, Pre pre, Keep keep = Keep(),
                   Over over = Over()) {
    pre();
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    unsigned bh[2], bm[2], bl[2];
    split_pair(top[0], top[1], bh[0], bm[0], bl[0]);
    split_pair(top[2], top[3], bh[1], bm[1], bl[1]);
    keep(bh, bm, bl);
    // The contraction runs over the 16 layer-3 neurons only, so a K = 32 instruction takes two terms per operand
    // (k-slots 0..3 of a lane group: one term of neurons 4g..4g+3, slots 4..7: another term of the same neurons)
    // and three instructions give the six products: [h|l] x [l|h], [h|m] x [m|h], [h|m] x [h|m] (weights first).
    const bf16x8 b_lh = __builtin_bit_cast(bf16x8, u32x4{bl[0], bl[1], bh[0], bh[1]});
    const bf16x8 b_mh = __builtin_bit_cast(bf16x8, u32x4{bm[0], bm[1], bh[0], bh[1]});
    const bf16x8 b_hm = __builtin_bit_cast(bf16x8, u32x4{bh[0], bh[1], bm[0], bm[1]});
    FENCE();
    PRIO_M();
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
    e0 = mfma_bf(o.t[0], b_lh, e0);
    e0 = mfma_bf(o.t[1], b_mh, e0);
    e0 = mfma_bf(o.t[1], b_hm, e0);
    FENCE();
    unsigned h[4], m[4], l[4];
    e1 = mfma_bf(o.t[2], b_lh, e1);
    e1 = mfma_bf(o.t[3], b_mh, e1);
    e1 = mfma_bf(o.t[3], b_hm, e1);
    d2[0] = e0 * m2[0];
    split_pair(d2[0][0], d2[0][1], h[0], m[0], l[0]);
    split_pair(d2[0][2], d2[0][3], h[1], m[1], l[1]);
    FENCE();
    over();
    d2[1] = e1 * m2[1];
    split_pair(d2[1][0], d2[1][1], h[2], m[2], l[2]);
    split_pair(d2[1][2], d2[1][3], h[3], m[3], l[3]);
    sd2.h = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
    sd2.m = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
    sd2.l = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
    PRIO_V();
}
// b: the three-way split of d2 (from chain_top).  pre() runs before the second half, whose MFMAs have the masking
// of the first half's result beside them.
template <class Pre, class Keep = NoKeep>
DEV void chain_bottom(const float *img, int lane, const f32x4 (&)[2], const B3 &b, const f32x4 (&m1)[4], f32x4 (&d1)[4],
                      const BotOps &o, Pre pre, Keep keep = Keep()) {
    B3 w2[2];
    load_b3(w2[0], img, QB_A2T + 6, lane);
    load_b3(w2[1], img, QB_A2T + 9, lane);
    keep(b);
    FENCE();
    PRIO_M();          // stays: the caller's outer products follow (or it switches back itself)
    f32x4 f[4], flo[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        flo[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    mfma6x2(flo[0], f[0], flo[1], f[1], o.w[0], o.w[1], b);
    FENCE();
    pre();
    FENCE();
    mfma6x2(flo[2], f[2], flo[3], f[3], w2[0], w2[1], b);
#if CRITIC_ONE_CHAIN
    d1[0] = f[0] * m1[0];
    d1[1] = f[1] * m1[1];
    FENCE();
    d1[2] = f[2] * m1[2];
    d1[3] = f[3] * m1[3];
#else
    d1[0] = (f[0] + flo[0]) * m1[0];
    d1[1] = (f[1] + flo[1]) * m1[1];
    FENCE();
    d1[2] = (f[2] + flo[2]) * m1[2];
    d1[3] = (f[3] + flo[3]) * m1[3];
#endif
}
#else
struct B3 {};          // (the split terms exist only in the bf16 build)
struct NoKeep {
    DEV void operator()(int, const B3 &) const {}
    DEV void operator()(const B3 &) const {}
};
struct NoOp {
    DEV void operator()() const {}
};
struct FrontOps {      // FC1 weights and accumulator start (b1), FC2 accumulator start (b2), first FC2 weight pair
    f32x4 a1, b1[4], b2[2], w0, w1;
};
struct Fc3Ops {        // FC3 accumulator start (b3) and weights; W4 and b4 for the VALU last layer
    f32x4 b3, w30, w31, w4;
    float b4;
};
struct TopOps {        // W3^T
    f32x4 t0, t1;
};
struct BotOps {        // W2^T, first half (u = 0)
    f32x4 w[4];
};
DEV void load_resident(FrontOps &, const float *, int) {}      // (the all-fp32 build reads every operand per tile)
DEV void load_ops(FrontOps &o, const float *img, int lane) {
    o.a1 = quad(img, Q_A1, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) o.b1[t] = quad(img, Q_B1 + t, lane);
    o.b2[0] = quad(img, Q_B2, lane);
    o.b2[1] = quad(img, Q_B2 + 1, lane);
    o.w0 = quad(img, Q_A2, lane);
    o.w1 = quad(img, Q_A2 + 4, lane);
}
DEV void load_ops(Fc3Ops &o, const float *img, int lane) {
    o.b3 = quad(img, Q_B3, lane);
    o.w30 = quad(img, Q_A3, lane);
    o.w31 = quad(img, Q_A3 + 1, lane);
    o.w4 = quad(img, Q_W4, lane);
    o.b4 = img[(Q_B4 * 64 + lane) * 4];
}
DEV void load_resident(Fc3Ops &, const float *, int) {}
DEV void load_ops(TopOps &o, const float *img, int lane) {
    o.t0 = quad(img, Q_A3T, lane);
    o.t1 = quad(img, Q_A3T + 1, lane);
}
DEV void load_ops(BotOps &o, const float *img, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) o.w[t] = quad(img, Q_A2T + 2 * t, lane);
}
// z0, z1 += W2 (rows 0..15 | 16..31) x src, src = 64 values per sample in four quads.  ACT: src holds FC1
// pre-activations; LeakyReLU is applied on the way in.  pre() runs before the last group.
template <bool ACT, bool BITS, class Pre>
DEV void fc2_block(f32x4 &z0, f32x4 &z1, f32x4 (&src)[4], f32x4 w0, f32x4 w1, f32x4 (&m)[4], unsigned &bits,
                   const float *img, int lane, float slope, Pre pre, unsigned long long *ts = nullptr) {
    if (ACT) act4<BITS>(src[0], slope, src[0], m[0], bits, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 n0 = w0, n1 = w1;
        if (t < 3) {
            n0 = quad(img, Q_A2 + t + 1, lane);
            n1 = quad(img, Q_A2 + 4 + t + 1, lane);
        } else {
            pre();
        }
        FENCE();
        // the activation of the next quad sits in the issue shadow of this group's MFMAs
        if (ACT && t < 3) act4<BITS>(src[t + 1], slope, src[t + 1], m[t + 1], bits, 4 * (t + 1));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z0 = mfma4(w0[r], src[t][r], z0);
            z1 = mfma4(w1[r], src[t][r], z1);
        }
        FENCE();
        w0 = n0;
        w1 = n1;
        TS(2 + t)
    }
}

// forward of one tile; `in` = input component g of sample j (x0, x1, c0, c1 by lane group).
template <bool BITS = false, class Pre>
DEV void forward_front(Acts &a, const float *img, int lane, float in, float slope, const FrontOps &o, Pre pre,
                       unsigned long long *ts = nullptr) {
    a.bits = 0u;
    TS(0)
#pragma unroll
    for (int t = 0; t < 4; ++t) a.h1[t] = mfma4(o.a1[t], in, o.b1[t]);
    TS(1)
    f32x4 z0 = o.b2[0], z1 = o.b2[1];
    fc2_block<true, BITS>(z0, z1, a.h1, o.w0, o.w1, a.m1, a.bits, img, lane, slope, pre, ts);
    act4<BITS>(z0, slope, a.h2[0], a.m2[0], a.bits, 16);
    act4<BITS>(z1, slope, a.h2[1], a.m2[1], a.bits, 20);
    TS(6)
}
// FC3 and the last layer (16 -> 1), which runs on the VALU: four FMAs per lane and a sum over the lane groups
template <bool BITS = false, bool SCORE = true, class Pre>
DEV void forward_back(Acts &a, const float *img, int lane, float slope, const Fc3Ops &o, Pre pre,
                      unsigned long long *ts = nullptr) {
    TS(7)
    pre();
    FENCE();
    f32x4 y0 = o.b3, y1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        y0 = mfma4(o.w30[r], a.h2[0][r], y0);
        y1 = mfma4(o.w31[r], a.h2[1][r], y1);
    }
    FENCE();
    TS(8)
    act4<BITS>(y0 + y1, slope, a.h3, a.m3, a.bits, 24);
    if (!SCORE) return;
    float s = o.w4[0] * a.h3[0];
#pragma unroll
    for (int r = 1; r < 4; ++r) s = __builtin_fmaf(o.w4[r], a.h3[r], s);
    a.score = sum_over_groups(s) + o.b4;
    TS(9)
}

// back-propagation of an upstream vector through the three hidden layers, in two steps:
//   top[r] (16 values on neurons 4g + r of layer 3, already masked) -> d2[2] (layer 2, masked) -> d1[4] (layer 1, masked)
template <class Pre>
DEV void chain_top(f32x4 top, const f32x4 (&m2)[2], f32x4 (&d2)[2], const TopOps &o, Pre pre) {
    pre();
    FENCE();
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        e0 = mfma4(o.t0[r], top[r], e0);
        e1 = mfma4(o.t1[r], top[r], e1);
    }
    FENCE();
    d2[0] = e0 * m2[0];
    d2[1] = e1 * m2[1];
}
// pre() runs before the second half
template <class Pre>
DEV void chain_bottom(const float *img, int lane, const f32x4 (&d2)[2], const f32x4 (&m1)[4], f32x4 (&d1)[4], const BotOps &o,
                      Pre pre) {
    f32x4 f[4], w2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        w2[t] = quad(img, Q_A2T + 2 * t + 1, lane);
    }
    FENCE();
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) f[t] = mfma4(o.w[t][r], d2[0][r], f[t]);
    FENCE();
    pre();
    FENCE();
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < 4; ++t) f[t] = mfma4(w2[t][r], d2[1][r], f[t]);
    FENCE();
#pragma unroll
    for (int t = 0; t < 4; ++t) d1[t] = f[t] * m1[t];
}

// the call shape of the bf16 build (split terms handed from chain_top to chain_bottom): nothing to hand over here
template <class Pre, class Keep = NoKeep, class Over = NoOp>
DEV void chain_top(f32x4 top, const f32x4 (&m2)[2], f32x4 (&d2)[2], B3 &, const TopOps &o, Pre pre, Keep = Keep(), Over over = Over()) {
    chain_top(top, m2, d2, o, pre);
    over();
}
template <class Pre, class Keep = NoKeep>
DEV void chain_bottom(const float *img, int lane, const f32x4 (&d2)[2], const B3 &, const f32x4 (&m1)[4], f32x4 (&d1)[4],
                      const BotOps &o, Pre pre, Keep = Keep()) {
    chain_bottom(img, lane, d2, m1, d1, o, pre);
}
#endif

// (W1[:, 0:2])^T d1 -> the two input-gradient components, in every lane of sample j.  Two of sixteen
// rows of an MFMA tile would be used, so this runs on the VALU: 32 FMAs per lane over its 16 neurons,
// then the sum over the four lane groups for both components with three row swaps.
DEV void to_input(const f32x4 (&d1)[4], float &g0, float &g1, const InOps &o) {
    float p0 = 0.f, p1 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p0 = __builtin_fmaf(o.w0[t][r], d1[t][r], p0);
            p1 = __builtin_fmaf(o.w1[t][r], d1[t][r], p1);
        }
    // halves: lanes 0..31 collect component 0 (rows g and g + 2), lanes 32..63 component 1
    const auto h = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), false, false);
    unsigned x = h[0], y = h[1];
    asm volatile("" : "+v"(x), "+v"(y));
    const unsigned s = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y));
    const auto r = __builtin_amdgcn_permlane16_swap(s, s, false, false);
    unsigned u = r[0], v = r[1];
    asm volatile("" : "+v"(u), "+v"(v));
    const unsigned tot = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, u) + __builtin_bit_cast(float, v));
    const auto q = __builtin_amdgcn_permlane32_swap(tot, tot, false, false);   // component 0 | component 1 everywhere
    unsigned a = q[0], b = q[1];
    asm volatile("" : "+v"(a), "+v"(b));
    g0 = __builtin_bit_cast(float, a);
    g1 = __builtin_bit_cast(float, b);
}

// ---- outer-product accumulation over the 16 samples of a tile --------------------------------------
struct Accum {
    f32x4 p1, p1b;     // lane n: sum over (even, odd) samples of left64[n] * (columns 0..3 of the right tile)
    f32x2 p1_sum;      // lane n: sum over (even, odd) samples of left64[n]
    f32x4 p2[4][2];    // [left64 tile t][right32 u]
    f32x4 p3[2];       // [left32 tile u]            x right16
};

// LDS addresses of this lane inside its wave's transposition buffer, wave base included: every access is
// one register plus an immediate offset (no address arithmetic in the tile loop)
typedef __attribute__((address_space(3))) char lds_char;
struct Addr {
    char *wl64;                          // p1: this lane's float4 stores (row j, columns 4g..)
#if CRITIC_OUTER_BF16
    char *tw;                            // term slots: this lane's 16 bytes of row j (chunk g, swizzled)
    lds_char *tr;                        // term slots: the 8 bytes this lane points the transposing read at
    lds_char *tr2, *tr3;                 // the same address again, opaque to the compiler (see read_op)
#else
    char *wl32, *wr32, *wr16;            // this lane's float4 stores (row j, columns 4g..)
    char *rl64[4], *rl32[4], *rr32[4], *rr16[4];   // operand reads for k-step q (row 4q + g, column j)
#endif
    char *row64[4];                      // p1: word lane ^ 4k of a left64 row (k = swizzle class of the row)
    char *col4;                          // p1: column lane & 3 of the constant region
    char *cst;                           // this lane's word of the constant region: column g of row j
};

DEV void make_addr(Addr &ad, char *wb, int j, int g) {
    ad.wl64 = wb + 4 * (LEFT_W + j * P64 + ((4 * g) ^ swz(j)));
#if CRITIC_OUTER_BF16
    // A slot is 16 rows (samples) of 64 bytes; chunk c of a row holds neurons 4c..4c+3 of an even tile and of the
    // next one, 8 bytes each.  The chunk index is XORed with bits 1..2 of the row: the 8 lanes of a 16-byte
    // store group then cover all 32 banks.  The transposing read of lane group kg takes rows 4kg..4kg+3:
    // lane 4q + p of the group points at chunk p of row 4kg + q (two-way bank conflict between rows r, r + 4).
    ad.tw = wb + 64 * j + 16 * (g ^ ((j >> 1) & 3));
    {
        const int row = 4 * g + (j >> 2), p = j & 3;
        ad.tr = (lds_char *)(wb + 64 * row + 16 * (p ^ ((row >> 1) & 3)));
        ad.tr2 = ad.tr;
        ad.tr3 = ad.tr;
        // (volatile: two identical plain asm statements on the same input are one value to the optimiser, and the
        // reads through tr2 and tr3 would merge again)
        asm volatile("" : "+v"(ad.tr2));
        asm volatile("" : "+v"(ad.tr3));
    }
#else
    ad.wl32 = wb + 4 * (LEFT_W + j * P32 + ((4 * g) ^ swz(j)));
    ad.wr32 = wb + 4 * (RIGHT_W + j * P32 + ((4 * g) ^ swz(j)));
    ad.wr16 = wb + 4 * (RIGHT_W + j * P16 + ((4 * g) ^ swz(j)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int s = 4 * q + g, c = j ^ swz(s);
        ad.rl64[q] = wb + 4 * (LEFT_W + s * P64 + c);
        ad.rl32[q] = wb + 4 * (LEFT_W + s * P32 + c);
        ad.rr32[q] = wb + 4 * (RIGHT_W + s * P32 + c);
        ad.rr16[q] = wb + 4 * (RIGHT_W + s * P16 + c);
    }
#endif
    const int lane = 16 * g + j;
#pragma unroll
    for (int k = 0; k < 4; ++k) ad.row64[k] = wb + 4 * (LEFT_W + (lane ^ (4 * k)));
    ad.col4 = wb + 4 * (CONST_W + (lane & 3));
    ad.cst = wb + 4 * (CONST_W + j * P16 + (g ^ swz(j)));
}

#define ST4(base, off, v) (*reinterpret_cast<f32x4 *>((base) + (off)) = (v))
#define LD1(base, off) (*reinterpret_cast<const float *>((base) + (off)))

// Each outer-product phase is split in two: park_* stores this lane's rows of the factor tiles, run_* reads
// them back transposed and issues the MFMAs.  The three phases share one buffer, so a phase is parked only
// after the previous one has run; between a park and its run the caller places other MFMA work, which
// covers the store -> read round trip through LDS (LDS operations of a wave execute in order).
// p1 += left64^T (x) cst16 where the 16-column right tile lives in the wave's constant region
DEV void park_p1(const Addr &ad, const f32x4 (&left)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) ST4(ad.wl64, 64 * t, left[t]);
    __builtin_amdgcn_wave_barrier();
}
// One 4x4-block MFMA per sample: lane n = 4b + c multiplies left64[s][n] with columns 0..3 of sample s of the
// constant region, so every product is a wanted one (a 16x16x4 tile would spend 16 columns on these 4).
// Every run_* comes in two parts, read_* (LDS -> registers) and mfma_*, so that the caller can issue the
// reads one MFMA group ahead.
struct P1R {
    float l[16], r[16];
};
DEV void read_p1(P1R &o, const Addr &ad) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        o.l[s] = LD1(ad.row64[(s >> 1) & 3], 4 * s * P64);
        o.r[s] = LD1(ad.col4, 4 * (s * P16 + swz(s)));
    }
    __builtin_amdgcn_wave_barrier();
}
DEV void mfma_p1(Accum &ac, const P1R &o) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s & 1) ac.p1b = mfma_blocks(o.r[s], o.l[s], ac.p1b);
        else ac.p1 = mfma_blocks(o.r[s], o.l[s], ac.p1);
        if (s & 1) ac.p1_sum += f32x2{o.l[s - 1], o.l[s]};
    }
}
#if CRITIC_OUTER_BF16
// ---- p2, p3 on the bf16 matrix cores ------------------------------------------------------------------
// x = h + m + l on both sides; the six products a K = 32 instruction needs three MFMAs for, two products
// each: the 8 k-slots of a lane group are four samples of one term and the same four samples of another.
//   [h|l] x [l|h] = hl + lh,  [h|m] x [m|h] = hm + mh,  [h|m] x [h|m] = hh + mm      (left term first)
struct OpL { bf16x8 hm, hl; };          // left factor (rows of the result): [h|m], [h|l]
struct OpR { bf16x8 hm, mh, lh; };      // right factor (columns): [h|m], [m|h], [l|h]
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
DEV s16x4 tr_read(lds_char *p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p); }
DEV bf16x8 cat(s16x4 a, s16x4 b) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
// h, m, l: byte offsets of the 8-byte pieces of the three terms.  A term that goes into several operands is read
// once per operand, through a copy of the address the compiler cannot see through: merged into one read, the
// value would have to be copied into place with VALU moves, and an LDS read is the cheaper instruction here.
DEV void read_op(OpL &o, const Addr &ad, int h, int m, int l) {
    o.hm = cat(tr_read(ad.tr + h), tr_read(ad.tr + m));
    o.hl = cat(tr_read(ad.tr2 + h), tr_read(ad.tr + l));
}
DEV void read_op(OpR &o, const Addr &ad, int h, int m, int l) {
    o.hm = cat(tr_read(ad.tr + h), tr_read(ad.tr + m));
    o.mh = cat(tr_read(ad.tr2 + m), tr_read(ad.tr2 + h));
    o.lh = cat(tr_read(ad.tr + l), tr_read(ad.tr3 + h));
}
// tile (0 or 1) of a parked pair of tiles: slot group at byte offset `base`, one slot per term
template <class Op>
DEV void read_pair_op(Op &o, const Addr &ad, int base, int tile) {
    read_op(o, ad, base + 8 * tile, base + SLOT_B + 8 * tile, base + 2 * SLOT_B + 8 * tile);
}
// (stored as float vectors like everything else in this buffer: the slots share memory with the p1 tile, and
// accesses of one type keep their program order)
DEV void park_b3(const Addr &ad, int base, const B3 &b) {
    ST4(ad.tw, base, __builtin_bit_cast(f32x4, b.h));
    ST4(ad.tw, base + SLOT_B, __builtin_bit_cast(f32x4, b.m));
    ST4(ad.tw, base + 2 * SLOT_B, __builtin_bit_cast(f32x4, b.l));
    __builtin_amdgcn_wave_barrier();
}
// one 16-neuron tile: [h|m] in the first slot, l in the first half of the second
DEV void park_single(const Addr &ad, int base, const unsigned (&h)[2], const unsigned (&m)[2], const unsigned (&l)[2]) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    ST4(ad.tw, base, __builtin_bit_cast(f32x4, u32x4{h[0], h[1], m[0], m[1]}));
    *reinterpret_cast<f32x2 *>(ad.tw + base + SLOT_B) = __builtin_bit_cast(f32x2, u32x2{l[0], l[1]});
    __builtin_amdgcn_wave_barrier();
}
DEV void read_single_op(OpR &o, const Addr &ad, int base) { read_op(o, ad, base, base + 8, base + SLOT_B); }
DEV void outer3(f32x4 &acc, const OpL &a, const OpR &b) {
    acc = mfma_bf(a.hl, b.lh, acc);
    acc = mfma_bf(a.hm, b.mh, acc);
    acc = mfma_bf(a.hm, b.hm, acc);
}
// both right tiles against one left tile, the two accumulator chains interleaved
DEV void outer3x2(f32x4 &acc0, f32x4 &acc1, const OpL &a, const OpR &b0, const OpR &b1) {
    acc0 = mfma_bf(a.hl, b0.lh, acc0);
    acc1 = mfma_bf(a.hl, b1.lh, acc1);
    acc0 = mfma_bf(a.hm, b0.mh, acc0);
    acc1 = mfma_bf(a.hm, b1.mh, acc1);
    acc0 = mfma_bf(a.hm, b0.hm, acc0);
    acc1 = mfma_bf(a.hm, b1.hm, acc1);
}
#else
DEV void park_p2(const Addr &ad, const f32x4 (&left)[4], const f32x4 (&right)[2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) ST4(ad.wl64, 64 * t, left[t]);
    ST4(ad.wr32, 0, right[0]);
    ST4(ad.wr32, 64, right[1]);
    __builtin_amdgcn_wave_barrier();
}
struct P2R {
    float r0[4], r1[4], l[4][4];
};
DEV void read_p2(P2R &o, const Addr &ad) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        o.r0[q] = LD1(ad.rr32[q], 0);
        o.r1[q] = LD1(ad.rr32[q], 64);
#pragma unroll
        for (int t = 0; t < 4; ++t) o.l[q][t] = LD1(ad.rl64[q], 64 * t);
    }
    __builtin_amdgcn_wave_barrier();
}
DEV void mfma_p2(Accum &ac, const P2R &o) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ac.p2[t][0] = mfma4(o.l[q][t], o.r0[q], ac.p2[t][0]);
            ac.p2[t][1] = mfma4(o.l[q][t], o.r1[q], ac.p2[t][1]);
        }
}
DEV void park_p3(const Addr &ad, const f32x4 (&left)[2], f32x4 right) {
    ST4(ad.wl32, 0, left[0]);
    ST4(ad.wl32, 64, left[1]);
    ST4(ad.wr16, 0, right);
    __builtin_amdgcn_wave_barrier();
}
struct P3R {
    float r[4], l0[4], l1[4];
};
DEV void read_p3(P3R &o, const Addr &ad) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        o.r[q] = LD1(ad.rr16[q], 0);
        o.l0[q] = LD1(ad.rl32[q], 0);
        o.l1[q] = LD1(ad.rl32[q], 64);
    }
    __builtin_amdgcn_wave_barrier();
}
DEV void mfma_p3(Accum &ac, const P3R &o) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ac.p3[0] = mfma4(o.l0[q], o.r[q], ac.p3[0]);
        ac.p3[1] = mfma4(o.l1[q], o.r[q], ac.p3[1]);
    }
}
#endif

DEV void zero_accum(Accum &ac) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    ac.p1 = z;
    ac.p1b = z;
    ac.p1_sum = f32x2{0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ac.p2[t][0] = z;
        ac.p2[t][1] = z;
    }
    ac.p3[0] = z;
    ac.p3[1] = z;
}

// p1/p2/p3 and the per-lane vectors -> this wave's row of flat critic gradients (in LDS).
// ncol1: columns of p1 that are W1 columns (4 for the backward, 2 for the penalty); bias1: p1_sum is the
// layer-1 bias gradient.
DEV void store_row(float *row, const Accum &ac, int j, int g, int ncol1, bool bias1) {
    {
        const int n = 16 * g + j;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < ncol1) row[C_W1 + n * 4 + c] = ac.p1[c] + ac.p1b[c];
        if (bias1) row[C_B1 + n] = ac.p1_sum[0] + ac.p1_sum[1];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 16 * t + 4 * g + r;
#pragma unroll
            for (int u = 0; u < 2; ++u) row[C_W2 + (16 * u + j) * 64 + n] = ac.p2[t][u][r];
        }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) row[C_W3 + j * 32 + 16 * u + 4 * g + r] = ac.p3[u][r];
}

DEV f32x4 sum_over_j(f32x4 v) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], off);
    return v;
}
// sum over the 16 lanes of a lane group, left in every lane: four adds whose second operand is the lane's row
// rotated by 8, 4, 2, 1 (DPP row_ror; no trip through the LDS crossbar, which a shuffle would take)
DEV float sum_over_j1(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

DEV void combine_rows(float *lds, float *slab_row) {
    __syncthreads();
    for (int e = threadIdx.x; e <= C_NP; e += blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < WG_WAVES; w += 4)
            acc += (lds[w * C_SLAB + e] + lds[(w + 1) * C_SLAB + e]) + (lds[(w + 2) * C_SLAB + e] + lds[(w + 3) * C_SLAB + e]);
        slab_row[e] = acc;
    }
}

// ---- tile hand-out ----------------------------------------------------------------------------------
// Units of consecutive tiles are handed out through a device counter, first come, first served: CUs
// run these kernels at visibly different speeds (per-wave run times of one launch spread by +-13% under
// a fixed assignment), and the launch ends with its slowest wave.  Every hand-out is an atomic on one
// address, and those take their turn in the L2 at about 12 ns each: with units of 8 tiles, 2^17 of them per
// launch at 2^24 samples were a floor of 1.5 ms under every kernel of this file (measured: the kernels took
// that long with their matrix instructions removed).  So a unit is BIG tiles long over the first 7/8 of the
// tiles and SMALL over the rest, where the grain decides how evenly the waves finish: 36,864 atomics.
// The counter is bumped one unit ahead with a hand-issued atomic (lane 0 only: the asm narrows exec itself,
// so no branch and no merge of register values surrounds it) whose result stays in next_raw until the unit
// is started; the compiler's own atomic sequence waits for the round trip at once.  The compiler does not
// count this load, hence the explicit wait before the value is read (wait and read are one asm statement,
// and the atomic takes next_raw as an in/out operand: no copy of the register between the issue and the
// wait).  queue == NULL: units in a fixed round-robin order (results then do not depend on timing).  The
// first unit of a wave is its own index.  The loops that use this are wave-uniform (all lanes active).
#ifndef TILE_UNIT_BIG
#define TILE_UNIT_BIG 32
#endif
#ifndef TILE_UNIT_SMALL
#define TILE_UNIT_SMALL 16
#endif
struct TileQueue {
    static constexpr unsigned BIG = TILE_UNIT_BIG, SMALL = TILE_UNIT_SMALL;
    unsigned *queue;
    unsigned n_waves, static_next, n_big, big, len, in_unit, next_raw, unit;
    int64_t base;                                 // first tile of the current unit
    DEV void start(unsigned u) {                  // unit u: `big` tiles each below n_big, SMALL from there on
        unit = u;
        len = u < n_big ? big : SMALL;
        base = u < n_big ? (int64_t)u * big : (int64_t)n_big * big + (int64_t)(u - n_big) * SMALL;
        in_unit = 0;
    }
    // big_tiles: length of the long units (k_gp_fused_mfma picks 64, 32 or 16 so that no unit crosses a row)
    DEV void init(unsigned *q, unsigned first, unsigned waves, int64_t ntiles, unsigned big_tiles = BIG) {
        queue = q;
        n_waves = waves;
        static_next = first;
        big = big_tiles;
        n_big = (unsigned)((ntiles - (ntiles >> 3)) / big);
        next_raw = 0;
        start(first);
    }
    DEV int64_t tile() const { return base + in_unit; }
    DEV void advance(int lane) {
        if (in_unit == 1 && queue != nullptr) {
            const unsigned one = 1u;
            unsigned long long saved;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, off sc0\n\ts_mov_b64 exec, %1"
                         : "+v"(next_raw), "=&s"(saved)
                         : "v"(queue), "v"(one)
                         : "memory");
        }
        if (++in_unit == len) {
            if (queue != nullptr) {
                unsigned raw;   // wait and read in one statement: nothing can read the register in between
                asm volatile("s_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %0, %1" : "=s"(raw) : "v"(next_raw) : "memory");
                start(n_waves + raw);
            } else {
                start(n_waves + static_next);
                static_next += n_waves;
            }
        }
    }
};

// experiment hook: ABL_EXTRA_VALU / ABL_EXTRA_MFMA extra independent instructions per tile (what a kernel's time
// does with them tells which unit bounds it); nothing in a product build
DEV void abl_extra(float &d0, float &d1, float &d2, float &d3, f32x4 &dm) {
#ifdef ABL_EXTRA_VALU
#pragma unroll
    for (int i = 0; i < ABL_EXTRA_VALU; i += 4) {
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d0));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d1));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d2));
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d3));
    }
#endif
#ifdef ABL_EXTRA_MFMA
    bf16x8 a = __builtin_bit_cast(bf16x8, f32x4{d0, d1, d2, d3});
#pragma unroll
    for (int i = 0; i < ABL_EXTRA_MFMA; ++i) dm = mfma_bf(a, a, dm);
#endif
}
// experiment hook: ABL_EXTRA_LDS extra 16-byte reads of the weight image per tile (results kept alive, not used)
DEV void abl_extra_lds(const float *img, int lane, f32x4 &sink) {
#ifdef ABL_EXTRA_LDS
#pragma unroll
    for (int i = 0; i < ABL_EXTRA_LDS; ++i) {
        f32x4 v = quad(img, Q_B1 + (i & 3), lane);
        asm volatile("" : "+v"(v));
        sink[i & 3] = v[i & 3];
    }
#endif
}
// ------------------------------------------------------------------------------------- kernels
#ifndef NOGRAD_WAVES
#define NOGRAD_WAVES 2      // measured: 3 waves per SIMD (168 registers, ~35 spilled) is 7% slower on the Stage-2 step
#endif
__global__ __launch_bounds__(WAVES * 64) void k_critic_fwd_mfma(const float *__restrict__ x, const float *__restrict__ cond,
                                                                int64_t N, const float *__restrict__ wb, float slope,
                                                                float *__restrict__ score, float *__restrict__ part) {
    __shared__ __attribute__((aligned(16))) float lds[IMG_WORDS];
    build_image(lds, wb);
    PRIO_V();
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the tile walk then runs on the scalar unit
    const int64_t ntiles = (N + 15) >> 4;
    float acc = 0.f;
    // the tile's input is loaded one tile ahead (one value per lane): no HBM latency in front of the MFMAs
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    // (no branch around the load: the index is clamped and the value selected afterwards.  With the load inside
    // a conditional block the compiler loses count of the loads in flight across the loop edge and waits for
    // all of them, the one just issued for the next tile included, in front of the first MFMA of every tile.)
    // The loaded value is not touched before the next iteration either: an instruction that consumes it here
    // (a select, a product) would bring the wait for the load to this point, and the latency would be exposed.
    const float *src = g < 2 ? x + g : cond + (g - 2);
    auto fetch = [&](int64_t tile) {
        const int64_t n = tile * 16 + j;
        return src[2 * (n < N ? n : 0)];            // n >= N past the last tile as well
    };
    float raw_next = fetch((int64_t)blockIdx.x * WAVES + wave);
    FrontOps front;
    Fc3Ops o3;
    load_resident(front, lds, lane);
    load_resident(o3, lds, lane);
    load_ops(front, lds, lane);
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += stride) {
        const int64_t n = tile * 16 + j;
        const bool valid = n < N;
        const float in = valid ? raw_next : 0.f;
        raw_next = fetch(tile + stride);
        Acts a;
        forward_front(a, lds, lane, in, slope, front, [&] { load_ops(o3, lds, lane); });
        forward_back(a, lds, lane, slope, o3, [&] { load_ops(front, lds, lane); });
        if (valid && g == 0) {
            if (score) score[n] = a.score;
            acc += a.score;
        }
    }
    if (part) {   // optional running sum: one atomic per wave (order-dependent in the last bits)
        acc = wave_sum(acc);
        if (lane == 0) atomicAdd(part, acc);
    }
}

__global__ __launch_bounds__(256) void k_sum_partials(const float *__restrict__ part, int n, float *__restrict__ out,
                                                      float scale) {
    __shared__ double red[4];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += (double)part[i];
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) * (double)scale);
}

template <bool WGRAD, bool DX>
__global__ __launch_bounds__((WGRAD ? WG_WAVES : WAVES) * 64, WGRAD ? WG_PER_CU : NOGRAD_WAVES) void k_critic_bwd_mfma(const float *__restrict__ x, const float *__restrict__ cond,
                                                                int64_t N, const float *__restrict__ wb, float slope,
                                                                const float *__restrict__ gscore, float gconst,
                                                                float2 *__restrict__ gx, float *__restrict__ slab,
                                                                float *__restrict__ score_sum, unsigned *__restrict__ queue) {
    // (dynamic: the size depends on WGRAD.  A static array above 64 KiB as such launches and completes on this
    // part without any attribute -- scratch/ubench/static_lds_probe.hip -- so the one variant of this kernel
    // that never finished with a static array failed for a reason of its own, which was not tracked down;
    // that variant is gone)
    extern __shared__ __attribute__((aligned(16))) float lds[];   // WGRAD: LDS_WORDS (> 64 KiB), else IMG_WORDS
    build_image(lds, wb);
    PRIO_V();
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the tile walk then runs on the scalar unit
    char *wbuf = reinterpret_cast<char *>(lds + IMG_WORDS + (WGRAD ? wave * WAVE_WORDS : 0));
    Addr ad;
    Accum ac;
    f32x4 db2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, db3 = {0.f, 0.f, 0.f, 0.f}, dw4 = {0.f, 0.f, 0.f, 0.f};
    float db4 = 0.f;
    if (WGRAD) {
        make_addr(ad, wbuf, j, g);
        zero_accum(ac);
        float *cst = lds + IMG_WORDS + wave * WAVE_WORDS + CONST_W;
        for (int k = lane; k < 256; k += 64) cst[k] = 0.f;            // the input tile [sample][in0..3]
        // the W1 gradient of a tile is accumulated during the next tile's forward (before the first: zeros)
        float *left = lds + IMG_WORDS + wave * WAVE_WORDS + LEFT_W;
        for (int k = lane; k < 16 * P64; k += 64) left[k] = 0.f;
        __builtin_amdgcn_wave_barrier();
    }
    const f32x4 w4 = quad(lds, Q_W4, lane);
    float ssum = 0.f;
#if defined(ABL_EXTRA_VALU) || defined(ABL_EXTRA_MFMA)
    float abl0 = slope, abl1 = slope + 1.f, abl2 = slope + 2.f, abl3 = slope + 3.f;
    f32x4 ablm = {0.f, 0.f, 0.f, 0.f};
#endif
#if defined(ABL_TIMING) || defined(ABL_CLOCK)
    // experiment: shader-clock cycles and 100 MHz ticks between here and the end of the wave's loop
    const unsigned long long clk0 = __builtin_readcyclecounter(), real0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef ABL_TIMING
    unsigned long long tacc[6] = {0}, tlast = clk0;
    unsigned long long fts[16] = {0};
#endif
#ifdef ABL_EXTRA_LDS
    f32x4 abl_sink = {0.f, 0.f, 0.f, 0.f};
#endif
    const int64_t ntiles = (N + 15) >> 4;
    // the tile's input (and upstream gradient) is loaded one tile ahead
    // (branch-free: clamped index, value selected afterwards -- a load inside a conditional block makes the
    // compiler wait for every load in flight, the next tile's included, in front of the tile's first MFMA)
    const float *src = g < 2 ? x + g : cond + (g - 2);
    const float *gsrc = gscore ? gscore : x;        // no upstream vector: any readable address, the value is not used
    // The loaded values are not touched before the next iteration either: an instruction that consumes one
    // here (a select, a product) would bring the wait for the load to this point and expose its latency.
    auto fetch_in = [&](int64_t tile) {
        const int64_t n = tile * 16 + j;
        return src[2 * (n < N ? n : 0)];            // n >= N past the last tile as well
    };
    auto fetch_gs = [&](int64_t tile) {
        const int64_t n = tile * 16 + j;
        return gsrc[(n < N && gscore) ? n : 0];
    };
    FrontOps front;
    Fc3Ops o3;
    load_resident(front, lds, lane);
    load_resident(o3, lds, lane);
    load_ops(front, lds, lane);
#ifdef ABL_PHASE_SHIFT     // experiment: the second wave of every SIMD starts a fraction of a tile late
    if (WGRAD && wave >= WG_WAVES / 2) __builtin_amdgcn_s_sleep(ABL_PHASE_SHIFT);
#endif
    TileQueue tq;
    constexpr int NW = WGRAD ? WG_WAVES : WAVES;
    tq.init(queue, blockIdx.x * NW + wave, gridDim.x * NW, ntiles);
    int64_t tile = tq.tile();
    float in_next = fetch_in(tile), gs_next = fetch_gs(tile);
    while (tile < ntiles) {
        const int64_t n = tile * 16 + j;
        const bool valid = n < N;
        const float in = valid ? in_next : 0.f, gs = valid ? (gscore ? gs_next : gconst) : 0.f;
        tq.advance(lane);
        tile = tq.tile();
        in_next = fetch_in(tile);
        gs_next = fetch_gs(tile);
        Acts a;
#ifdef ABL_STAGGER     // experiment (fixed tile order, equal tile counts only): the two waves of a SIMD half a tile apart
        if (WGRAD && wave < WG_WAVES / 2) __builtin_amdgcn_s_barrier();
#endif
#ifdef ABL_TIMING
#define STAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long now = __builtin_readcyclecounter(); tacc[k] += now - tlast; tlast = now; __builtin_amdgcn_sched_barrier(0); }
        STAMP(0)
#else
#define STAMP(k)
#endif
        TopOps top;
        BotOps bot;
        InOps ino;
        P1R p1r;
#ifdef ABL_TIMING
        fts[15] = __builtin_readcyclecounter();
        unsigned long long *const ftp = fts;
#else
        unsigned long long *const ftp = nullptr;
#endif
#if CRITIC_OUTER_BF16
        // The factors of the two inner weight gradients are parked as the bf16 terms the products split them
        // into anyway -- h1 in FC2, h2 in FC3, gz3 and gz2 in the chain -- and come back transposed right before
        // the chain step whose MFMAs cover the read: p3 (h2 x gz3) after the top of the chain, p2 (h1 x gz2)
        // after its bottom, one h1 tile ahead.
        OpL h2l[2], h1l[2];
        OpR d3r, d2r[2];
        forward_front(a, lds, lane, in, slope, front, [&] {
            load_ops(o3, lds, lane);
            if (WGRAD) read_p1(p1r, ad);   // previous tile's W1 gradient: parked a whole forward ago
        }, ftp, [&](int half, const B3 &b) {
            if (WGRAD) park_b3(ad, S_H1 + 3 * SLOT_B * half, b);
        });
        if (WGRAD) mfma_p1(ac, p1r);
        forward_back(a, lds, lane, slope, o3, [&] { load_ops(top, lds, lane); }, ftp, [&](const B3 &b) {
            if (WGRAD) park_b3(ad, S_H2, b);     // after read_p1: the slots lie inside the p1 tile
        });
        STAMP(1)
#ifdef ABL_STAGGER
        if (WGRAD && wave >= WG_WAVES / 2) __builtin_amdgcn_s_barrier();
#endif
        ssum += (valid && g == 0) ? a.score : 0.f;
        const f32x4 (&m1)[4] = a.m1;
        const f32x4 (&m2)[2] = a.m2;
        const f32x4 gz3 = w4 * gs * a.m3;
        f32x4 gz2[2], gz1[4];
        B3 sgz2;
        chain_top(gz3, m2, gz2, sgz2, top, [&] { load_ops(bot, lds, lane); },
                  [&](const unsigned (&bh)[2], const unsigned (&bm)[2], const unsigned (&bl)[2]) {
                      if (WGRAD) {
                          park_single(ad, S_D3, bh, bm, bl);
                          read_pair_op(h2l[0], ad, S_H2, 0);
                          read_pair_op(h2l[1], ad, S_H2, 1);
                          read_single_op(d3r, ad, S_D3);
                          __builtin_amdgcn_wave_barrier();
                      }
                  },
                  [&] {      // p3, beside the split of the second half of gz2
                      if (WGRAD) {
                          ac.p3[0] = mfma_bf(h2l[0].hl, d3r.lh, ac.p3[0]);
                          ac.p3[1] = mfma_bf(h2l[1].hl, d3r.lh, ac.p3[1]);
                          ac.p3[0] = mfma_bf(h2l[0].hm, d3r.mh, ac.p3[0]);
                          ac.p3[1] = mfma_bf(h2l[1].hm, d3r.mh, ac.p3[1]);
                          ac.p3[0] = mfma_bf(h2l[0].hm, d3r.hm, ac.p3[0]);
                          ac.p3[1] = mfma_bf(h2l[1].hm, d3r.hm, ac.p3[1]);
                      }
                  });
        if (WGRAD) {
            db2[0] += gz2[0];
            db2[1] += gz2[1];
        }
        chain_bottom(lds, lane, gz2, sgz2, m1, gz1, bot, [&] {
            if (WGRAD) {
                read_pair_op(d2r[0], ad, S_D2, 0);
                read_pair_op(d2r[1], ad, S_D2, 1);
                read_pair_op(h1l[0], ad, S_H1, 0);
                __builtin_amdgcn_wave_barrier();
            }
            if (DX) load_ops(ino, lds, lane);
        }, [&](const B3 &b) {
            if (WGRAD) park_b3(ad, S_D2, b);     // the h2 slots: their reads are issued
        });
        STAMP(2)
        if (!WGRAD) {
            PRIO_V();
            load_ops(front, lds, lane);      // next tile
            FENCE();
        }
        if (WGRAD) {
            // (the fences keep every read one MFMA group ahead of its use; the next tile's operands are read behind the
            // last of the transposing reads -- LDS reads return in order, and thirteen 16-byte reads in front of a
            // group's factors were a wait in front of its MFMAs)
            read_pair_op(h1l[1], ad, S_H1, 1);
            FENCE();
            outer3x2(ac.p2[0][0], ac.p2[0][1], h1l[0], d2r[0], d2r[1]);
            FENCE();
            read_pair_op(h1l[0], ad, S_H1 + 3 * SLOT_B, 0);
            FENCE();
            outer3x2(ac.p2[1][0], ac.p2[1][1], h1l[1], d2r[0], d2r[1]);
            FENCE();
            read_pair_op(h1l[1], ad, S_H1 + 3 * SLOT_B, 1);
            load_ops(front, lds, lane);      // next tile
            FENCE();
            outer3x2(ac.p2[2][0], ac.p2[2][1], h1l[0], d2r[0], d2r[1]);
            FENCE();
            outer3x2(ac.p2[3][0], ac.p2[3][1], h1l[1], d2r[0], d2r[1]);
            PRIO_V();
            __builtin_amdgcn_wave_barrier();
            STAMP(4)
        }
        if (DX) {
            float g0, g1;
            to_input(gz1, g0, g1, ino);
            if (valid && g == 0) gx[n] = make_float2(g0, g1);
        }
        if (WGRAD) {
            db3 += gz3;
            dw4 += a.h3 * gs;
            db4 += gs;
            // the input tile: column g of row j
            *reinterpret_cast<float *>(ad.cst) = in;
            park_p1(ad, gz1);
            STAMP(3)
        }
#else
        P2R p2r;
        P3R p3r;
        forward_front(a, lds, lane, in, slope, front, [&] {
            load_ops(o3, lds, lane);
            if (WGRAD) read_p1(p1r, ad);   // previous tile's W1 gradient: parked a whole forward ago
        }, ftp);
        if (WGRAD) mfma_p1(ac, p1r);
        forward_back(a, lds, lane, slope, o3, [&] { load_ops(top, lds, lane); }, ftp);
        STAMP(1)
        ssum += (valid && g == 0) ? a.score : 0.f;
        const f32x4 (&m1)[4] = a.m1;
        const f32x4 (&m2)[2] = a.m2;
        const f32x4 gz3 = w4 * gs * a.m3;
        f32x4 gz2[2], gz1[4];
        // The outer-product phases are parked as soon as their factors exist and run after the next piece
        // of the chain, whose MFMAs cover the LDS round trip: p3 (h2 x gz3) around the top of the chain,
        // p2 (h1 x gz2) around its bottom; only p1 (gz1 x input) follows its own park directly.
        if (WGRAD) park_p3(ad, a.h2, gz3);
        B3 sgz2;
        chain_top(gz3, m2, gz2, sgz2, top, [&] {
            load_ops(bot, lds, lane);
            if (WGRAD) read_p3(p3r, ad);
        });
        if (WGRAD) {
            mfma_p3(ac, p3r);
            park_p2(ad, a.h1, gz2);
        }
        chain_bottom(lds, lane, gz2, sgz2, m1, gz1, bot, [&] {
            if (WGRAD) read_p2(p2r, ad);
            if (DX) load_ops(ino, lds, lane);
        });
        STAMP(2)
        load_ops(front, lds, lane);          // next tile
        FENCE();
        if (WGRAD) {
            mfma_p2(ac, p2r);
            STAMP(4)
        }
        if (DX) {
            float g0, g1;
            to_input(gz1, g0, g1, ino);
            if (valid && g == 0) gx[n] = make_float2(g0, g1);
        }
        if (WGRAD) {
            db2[0] += gz2[0];
            db2[1] += gz2[1];
            db3 += gz3;
            dw4 += a.h3 * gs;
            db4 += gs;
            // the input tile: column g of row j
            *reinterpret_cast<float *>(ad.cst) = in;
            park_p1(ad, gz1);
            STAMP(3)
        }
#endif
#if defined(ABL_EXTRA_VALU) || defined(ABL_EXTRA_MFMA)
        abl_extra(abl0, abl1, abl2, abl3, ablm);
#endif
#ifdef ABL_EXTRA_LDS
        abl_extra_lds(lds, lane, abl_sink);
#endif
    }
#ifdef ABL_EXTRA_LDS
    if (abl_sink[0] + abl_sink[1] + abl_sink[2] + abl_sink[3] == 1.2345f) ssum += 1.f;
#endif
#if defined(ABL_EXTRA_VALU) || defined(ABL_EXTRA_MFMA)
    if (abl0 + abl1 + abl2 + abl3 + ablm[0] == 1.2345f) ssum += 1.f;
#endif
    if (WGRAD) {
        P1R p1r;
        read_p1(p1r, ad);
        mfma_p1(ac, p1r);
    }
#ifdef ABL_TIMING
    if (WGRAD && lane == 0)
        for (int k = 0; k < 6; ++k) atomicAdd(reinterpret_cast<unsigned long long *>(slab + (size_t)2048 * C_SLAB) + k, tacc[k]);
    if (WGRAD && lane == 0)
        for (int k = 0; k < 10; ++k) atomicAdd(reinterpret_cast<unsigned long long *>(slab + (size_t)2048 * C_SLAB) + 10 + k, fts[k]);
#endif
#if defined(ABL_TIMING) || defined(ABL_CLOCK)
    if (WGRAD && lane == 0) {   // slots 6, 7: their ratio is the shader clock in units of 100 MHz
        atomicAdd(reinterpret_cast<unsigned long long *>(slab + (size_t)2048 * C_SLAB) + 6, __builtin_readcyclecounter() - clk0);
        atomicAdd(reinterpret_cast<unsigned long long *>(slab + (size_t)2048 * C_SLAB) + 7, __builtin_amdgcn_s_memrealtime() - real0);
        atomicMax(reinterpret_cast<unsigned long long *>(slab + (size_t)2048 * C_SLAB) + 8, __builtin_amdgcn_s_memrealtime() - real0);
        atomicMin(reinterpret_cast<unsigned long long *>(slab + (size_t)2048 * C_SLAB) + 9, __builtin_amdgcn_s_memrealtime() - real0);
    }
#endif
    if (!WGRAD) {
        if (score_sum) {   // no slab in this variant: one atomic per wave (order-dependent in the last bits)
            ssum = wave_sum(ssum);
            if (lane == 0) atomicAdd(score_sum, ssum);
        }
        return;
    }
    __syncthreads();
    float *row = lds + wave * C_SLAB;
    store_row(row, ac, j, g, 4, true);
    ssum = wave_sum(ssum);
    if (lane == 0) row[C_NP] = ssum;            // extra slab column: sum of scores, reduced in a fixed order
    db2[0] = sum_over_j(db2[0]);
    db2[1] = sum_over_j(db2[1]);
    db3 = sum_over_j(db3);
    dw4 = sum_over_j(dw4);
    db4 = sum_over_j1(db4);
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            row[C_B2 + 4 * g + r] = db2[0][r];
            row[C_B2 + 16 + 4 * g + r] = db2[1][r];
            row[C_B3 + 4 * g + r] = db3[r];
            row[C_W4 + 4 * g + r] = dw4[r];
        }
        if (g == 0) row[C_B4] = db4;
    }
    combine_rows(lds, slab + (int64_t)blockIdx.x * C_SLAB);
}

// ---- gradient penalty, rows of L samples with L % 16 == 0 (or L == 1: every sample is its own row) -------
// pass A: forward of x_hat, mask bits (one word per lane and tile), g per sample, sum |g|^2 per tile
template <bool ROW1>
__global__ __launch_bounds__(WAVES * 64, NOGRAD_WAVES) void k_gp_a_mfma(const float *__restrict__ real, const float *__restrict__ fake,
                                                          const float *__restrict__ cond, const float *__restrict__ alpha,
                                                          int64_t N, int64_t L, const float *__restrict__ wb, float slope,
                                                          float2 *__restrict__ gout, unsigned *__restrict__ maskw,
                                                          float *__restrict__ tile_sq, unsigned *__restrict__ queue) {
    __shared__ __attribute__((aligned(16))) float lds[IMG_WORDS];
    build_image(lds, wb);
    PRIO_V();
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the tile walk then runs on the scalar unit
    const f32x4 w4 = quad(lds, Q_W4, lane);
    const int64_t ntiles = (N + 15) >> 4;
    const unsigned tpr = ROW1 ? 1u : (unsigned)(L >> 4);      // tiles per row
    // the tile's input is loaded one tile ahead
    // branch-free (clamped indices, values selected afterwards; see k_critic_bwd_mfma): lane groups 2, 3 load their
    // condition component twice and an interpolation coefficient they do not use
    // The loaded values are carried to the next iteration untouched (an instruction consuming them here would
    // bring the wait for the loads to this point).
    const float *src_a = g < 2 ? real + g : cond + (g - 2), *src_b = g < 2 ? fake + g : cond + (g - 2);
    struct Raw { float al, a, b; };
    auto fetch = [&](int64_t tile) {
        const int64_t n = tile * 16 + j;
        const int64_t m = n < N ? n : 0;            // n >= N past the last tile as well
        // rows of L samples are whole tiles (L % 16 == 0, checked by the launcher): the row of a tile is one
        // 32-bit division of wave-uniform values, not a 64-bit division per lane
        Raw r;
        r.al = alpha[ROW1 ? m : (int64_t)((unsigned)(tile < ntiles ? tile : 0) / tpr)];
        r.a = src_a[2 * m];
        r.b = src_b[2 * m];
        return r;
    };
    FrontOps front;
    Fc3Ops o3;
    load_resident(front, lds, lane);
    load_resident(o3, lds, lane);
    load_ops(front, lds, lane);
#if defined(ABL_EXTRA_VALU) || defined(ABL_EXTRA_MFMA)
    float abl0 = slope, abl1 = slope + 1.f, abl2 = slope + 2.f, abl3 = slope + 3.f;
    f32x4 ablm = {0.f, 0.f, 0.f, 0.f};
#endif
    TileQueue tq;
    tq.init(queue, blockIdx.x * WAVES + wave, gridDim.x * WAVES, ntiles);
    int64_t tile_next = tq.tile();
    Raw raw_next = fetch(tile_next);
    while (tile_next < ntiles) {
        const int64_t tile = tile_next;
        const int64_t n = tile * 16 + j;
        const bool valid = n < N;
        const float in = valid ? (g < 2 ? raw_next.al * raw_next.a + (1.f - raw_next.al) * raw_next.b : raw_next.a) : 0.f;
        tq.advance(lane);
        tile_next = tq.tile();
        raw_next = fetch(tile_next);
        Acts a;
        TopOps top;
        BotOps bot;
        InOps ino;
        forward_front<true>(a, lds, lane, in, slope, front, [&] { load_ops(o3, lds, lane); });
        forward_back<true, false>(a, lds, lane, slope, o3, [&] { load_ops(top, lds, lane); });
        const f32x4 (&m1)[4] = a.m1;
        const f32x4 (&m2)[2] = a.m2;
        const unsigned bits = a.bits;
        const f32x4 v3 = w4 * a.m3;
        f32x4 v2[2], v1[4];
        B3 sv2;
        chain_top(v3, m2, v2, sv2, top, [&] { load_ops(bot, lds, lane); });
        chain_bottom(lds, lane, v2, sv2, m1, v1, bot, [&] { load_ops(ino, lds, lane); });
        PRIO_V();
        load_ops(front, lds, lane);          // next tile
        FENCE();
        float g0, g1;
        to_input(v1, g0, g1, ino);
        if (!valid) { g0 = 0.f; g1 = 0.f; }
#ifdef ABL_GPA_NOSTORE     // experiment: no global stores in the loop (the condition never holds)
        if (bits != 0x12345678u || g0 != 1.2345f) continue;
#endif
        maskw[tile * 64 + lane] = bits;
        if (valid && g == 0) gout[n] = make_float2(g0, g1);
        if (!ROW1) {
            float sq = sum_over_j1(g0 * g0 + g1 * g1);
            if (lane == 0) tile_sq[tile] = sq;
        }
#if defined(ABL_EXTRA_VALU) || defined(ABL_EXTRA_MFMA)
        abl_extra(abl0, abl1, abl2, abl3, ablm);
#endif
    }
#if defined(ABL_EXTRA_VALU) || defined(ABL_EXTRA_MFMA)
    if (abl0 + abl1 + abl2 + abl3 + ablm[0] == 1.2345f) tile_sq[0] = 0.f;
#endif
}

// pass B (rows of L samples): per-row norm, coefficient and penalty term; one wave per row
__global__ __launch_bounds__(64) void k_gp_rows_tiles(const float *__restrict__ tile_sq, int64_t B, int64_t tiles_per_row,
                                                      float *__restrict__ rownorm, float *__restrict__ rowc,
                                                      float *__restrict__ pen) {
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        float s = 0.f;
        for (int64_t i = threadIdx.x; i < tiles_per_row; i += 64) s += tile_sq[b * tiles_per_row + i];
        s = wave_sum(s);
        if (threadIdx.x == 0) {
            const float nrm = sqrtf(s);
            rownorm[b] = nrm;
            rowc[b] = nrm > 0.f ? 2.f * (nrm - 1.f) / (nrm * (float)B) : 0.f;
            pen[b] = (nrm - 1.f) * (nrm - 1.f);
        }
    }
}
// pass B for rows of one sample
__global__ __launch_bounds__(256) void k_gp_rows_single(const float2 *__restrict__ g, int64_t B, float *__restrict__ rownorm,
                                                        float *__restrict__ rowc, float *__restrict__ pen) {
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < B; b += (int64_t)gridDim.x * 256) {
        const float2 v = g[b];
        const float nrm = sqrtf(v.x * v.x + v.y * v.y);
        rownorm[b] = nrm;
        rowc[b] = nrm > 0.f ? 2.f * (nrm - 1.f) / (nrm * (float)B) : 0.f;
        pen[b] = (nrm - 1.f) * (nrm - 1.f);
    }
}

// pass C: reverse chain with gh = c_row * g and the weight gradient
template <bool ROW1>
__global__ __launch_bounds__(WG_WAVES * 64, WG_PER_CU) void k_gp_c_mfma(const float2 *__restrict__ gin, const unsigned *__restrict__ maskw,
                                                          const float *__restrict__ rowc, int64_t N, int64_t L,
                                                          const float *__restrict__ wb, float slope,
                                                          float *__restrict__ slab, unsigned *__restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // LDS_WORDS
    build_image(lds, wb);
    PRIO_V();
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the tile walk then runs on the scalar unit
    char *wbuf = reinterpret_cast<char *>(lds + IMG_WORDS + wave * WAVE_WORDS);
    Addr ad;
    Accum ac;
    make_addr(ad, wbuf, j, g);
    zero_accum(ac);
    {
        float *cst = lds + IMG_WORDS + wave * WAVE_WORDS + CONST_W;
        for (int k = lane; k < 256; k += 64) cst[k] = 0.f;     // columns 2, 3 of the gh tile stay 0
        // the W1 gradient of a tile is accumulated during the next tile's first MFMA block (before the first: zeros)
        float *left = lds + IMG_WORDS + wave * WAVE_WORDS + LEFT_W;
        for (int k = lane; k < 16 * P64; k += 64) left[k] = 0.f;
        __builtin_amdgcn_wave_barrier();
    }
    f32x4 dw4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 w4 = quad(lds, Q_W4, lane);
    FrontOps front;
    Fc3Ops o3;
    load_resident(front, lds, lane);
    load_resident(o3, lds, lane);
    load_ops(front, lds, lane);
    const int64_t ntiles = (N + 15) >> 4;
    const unsigned tpr = ROW1 ? 1u : (unsigned)(L >> 4);      // tiles per row
    // the tile's mask word, input gradient and row coefficient are loaded one tile ahead
    // (branch-free, clamped indices, and carried to the next iteration untouched: see k_critic_bwd_mfma)
    unsigned bits_next = 0u;
    float2 gv_next = make_float2(0.f, 0.f);
    float c_next = 0.f;
    auto fetch = [&](int64_t tile) {
        const int64_t n = tile * 16 + j;
        const int64_t tc = tile < ntiles ? tile : 0, m = n < N ? n : 0;
        bits_next = maskw[tc * 64 + lane];
        gv_next = gin[m];
        c_next = rowc[ROW1 ? m : (int64_t)((unsigned)tc / tpr)];      // whole tiles per row, as in pass A
    };
    TileQueue tq;
    tq.init(queue, blockIdx.x * WG_WAVES + wave, gridDim.x * WG_WAVES, ntiles);
    int64_t tile = tq.tile();
    fetch(tile);
    while (tile < ntiles) {
        const unsigned bits = bits_next;
        const bool valid = tile * 16 + j < N;
        const float gh0 = valid ? c_next * gv_next.x : 0.f, gh1 = valid ? c_next * gv_next.y : 0.f;
        tq.advance(lane);
        tile = tq.tile();
        fetch(tile);
        f32x4 m1[4], m2[2], m3;
        // the mask of a neuron from its bit.  (A hand-written v_bfi_b32 on the sign-extended bit is one instruction
        // shorter, but an instruction inside an asm statement is invisible to the compiler's matrix-instruction hazard
        // handling, and the previous tile's last MFMAs may still be executing here.)
        auto unpack = [&](int k) { return (bits >> k) & 1u ? 1.f : slope; };
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) m1[t][r] = unpack(4 * t + r);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) m2[u][r] = unpack(16 + 4 * u + r);
#pragma unroll
        for (int r = 0; r < 4; ++r) m3[r] = unpack(24 + r);
        const f32x4 v3 = w4 * m3;
        f32x4 v2[2], v1[4];
        // du1 = (W1[:, 0:2] gh) * D1 : k = input column on the lane groups (groups 2, 3 carry zeros)
        const float ghb = g == 0 ? gh0 : (g == 1 ? gh1 : 0.f);
        f32x4 du1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            du1[t] = (CRITIC_FC1_BF16 ? mfma_bf(front.a1b[t], fc1_operand(ghb, 0u), f32x4{0.f, 0.f, 0.f, 0.f})
                                      : mfma4(front.a1[t], ghb, f32x4{0.f, 0.f, 0.f, 0.f})) * m1[t];
        // dv2 = W2 du1, du2 = dv2 * D2
        f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
        TopOps top;
        BotOps bot;
        P1R p1r;
        unsigned no_bits = 0u;
        auto pre_fc2 = [&] {
            load_ops(o3, lds, lane);
            read_p1(p1r, ad);           // previous tile's v1 x gh
        };
#if CRITIC_OUTER_BF16
        // as in the backward kernel: p3 = du2 x v3 after the top of the chain, p2 = du1 x v2 after its bottom, the
        // factors parked as the bf16 terms of the products that consume them
        OpL du2l[2], du1l[2];
        OpR v3r, v2r[2];
        fc2_block<false, false>(z0, z1, du1, front.w, m1, no_bits, lds, lane, slope, pre_fc2, nullptr,
                                [&](int half, const B3 &b) { park_b3(ad, S_H1 + 3 * SLOT_B * half, b); });
        mfma_p1(ac, p1r);
        f32x4 du2[2] = {z0 * m2[0], z1 * m2[1]};
        load_ops(top, lds, lane);
        FENCE();
        // dv3 = W3 du2 ; dW4 += dv3 * D3
        const f32x4 dv3 = fc3_product(du2[0], du2[1], o3.w, f32x4{0.f, 0.f, 0.f, 0.f},
                                      [&](const B3 &b) { park_b3(ad, S_H2, b); });      // after read_p1: inside the p1 tile
        FENCE();
        dw4 += dv3 * m3;
        B3 sv2;
        chain_top(v3, m2, v2, sv2, top, [&] { load_ops(bot, lds, lane); },
                  [&](const unsigned (&bh)[2], const unsigned (&bm)[2], const unsigned (&bl)[2]) {
                      park_single(ad, S_D3, bh, bm, bl);
                      read_pair_op(du2l[0], ad, S_H2, 0);
                      read_pair_op(du2l[1], ad, S_H2, 1);
                      read_single_op(v3r, ad, S_D3);
                      __builtin_amdgcn_wave_barrier();
                  },
                  [&] {      // p3, beside the split of the second half of v2
                      ac.p3[0] = mfma_bf(du2l[0].hl, v3r.lh, ac.p3[0]);
                      ac.p3[1] = mfma_bf(du2l[1].hl, v3r.lh, ac.p3[1]);
                      ac.p3[0] = mfma_bf(du2l[0].hm, v3r.mh, ac.p3[0]);
                      ac.p3[1] = mfma_bf(du2l[1].hm, v3r.mh, ac.p3[1]);
                      ac.p3[0] = mfma_bf(du2l[0].hm, v3r.hm, ac.p3[0]);
                      ac.p3[1] = mfma_bf(du2l[1].hm, v3r.hm, ac.p3[1]);
                  });
        chain_bottom(lds, lane, v2, sv2, m1, v1, bot, [&] {
            read_pair_op(v2r[0], ad, S_D2, 0);
            read_pair_op(v2r[1], ad, S_D2, 1);
            read_pair_op(du1l[0], ad, S_H1, 0);
            __builtin_amdgcn_wave_barrier();
        }, [&](const B3 &b) { park_b3(ad, S_D2, b); });      // the du2 slots: their reads are issued
        // (the fences keep every read one MFMA group ahead of its use; the next tile's operands behind the last
        // transposing read, see k_critic_bwd_mfma)
        read_pair_op(du1l[1], ad, S_H1, 1);
        FENCE();
        outer3x2(ac.p2[0][0], ac.p2[0][1], du1l[0], v2r[0], v2r[1]);
        FENCE();
        read_pair_op(du1l[0], ad, S_H1 + 3 * SLOT_B, 0);
        FENCE();
        outer3x2(ac.p2[1][0], ac.p2[1][1], du1l[1], v2r[0], v2r[1]);
        FENCE();
        read_pair_op(du1l[1], ad, S_H1 + 3 * SLOT_B, 1);
        load_ops(front, lds, lane);           // next tile
        FENCE();
        outer3x2(ac.p2[2][0], ac.p2[2][1], du1l[0], v2r[0], v2r[1]);
        FENCE();
        outer3x2(ac.p2[3][0], ac.p2[3][1], du1l[1], v2r[0], v2r[1]);
        PRIO_V();
        __builtin_amdgcn_wave_barrier();
        // gh tile: column g (< 2) of row j
        if (g < 2) *reinterpret_cast<float *>(ad.cst) = ghb;
        park_p1(ad, v1);
#else
        P2R p2r;
        P3R p3r;
#if CRITIC_BF16X3
        fc2_block<false, false>(z0, z1, du1, front.w, m1, no_bits, lds, lane, slope, pre_fc2);
#else
        fc2_block<false, false>(z0, z1, du1, front.w0, front.w1, m1, no_bits, lds, lane, slope, pre_fc2);
#endif
        mfma_p1(ac, p1r);
        f32x4 du2[2] = {z0 * m2[0], z1 * m2[1]};
        // every outer-product phase is parked as soon as its factors exist and run after the next block of
        // chain MFMAs (which covers the LDS round trip): p3 = du2 x v3, p2 = du1 x v2, p1 = v1 x gh
        park_p3(ad, du2, v3);
        // dv3 = W3 du2 ; dW4 += dv3 * D3
        load_ops(top, lds, lane);
        read_p3(p3r, ad);
        FENCE();
#if CRITIC_BF16X3
        const f32x4 dv3 = fc3_product(du2[0], du2[1], o3.w, f32x4{0.f, 0.f, 0.f, 0.f});
#else
        f32x4 y0 = {0.f, 0.f, 0.f, 0.f}, y1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y0 = mfma4(o3.w30[r], du2[0][r], y0);
            y1 = mfma4(o3.w31[r], du2[1][r], y1);
        }
        const f32x4 dv3 = y0 + y1;
#endif
        FENCE();
        dw4 += dv3 * m3;
        mfma_p3(ac, p3r);
        B3 sv2;
        chain_top(v3, m2, v2, sv2, top, [&] { load_ops(bot, lds, lane); });
        park_p2(ad, du1, v2);
        chain_bottom(lds, lane, v2, sv2, m1, v1, bot, [&] { read_p2(p2r, ad); });
        load_ops(front, lds, lane);           // next tile
        FENCE();
        mfma_p2(ac, p2r);
        // gh tile: column g (< 2) of row j
        if (g < 2) *reinterpret_cast<float *>(ad.cst) = ghb;
        park_p1(ad, v1);
#endif
    }
    {
        P1R p1r;
        read_p1(p1r, ad);
        mfma_p1(ac, p1r);
    }
    __syncthreads();
    float *row = lds + wave * C_SLAB;
    for (int e = lane; e <= C_NP; e += 64) row[e] = 0.f;    // bias gradients and W1[:, 2:4] get no contribution
    __builtin_amdgcn_wave_barrier();
    store_row(row, ac, j, g, 2, false);
    dw4 = sum_over_j(dw4);
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) row[C_W4 + 4 * g + r] = dw4[r];
    }
    combine_rows(lds, slab + (int64_t)blockIdx.x * C_SLAB);
}

#if CRITIC_OUTER_BF16
// ---- gradient penalty in one pass over the samples ------------------------------------------------------------
// The weight gradient of the penalty is linear in the row coefficient c_b = 2(|g_b| - 1) / (|g_b| B):
//   d gp / dW = sum over rows b of c_b * G_b,   G_b = sum over the samples of row b of (terms in g, not in c_b),
// so the reverse chain of pass C can run with the unscaled input gradient g in the same kernel that computes g, and
// the coefficient is applied afterwards to partial sums that do not cross a row: every hand-out unit (64, 32 or 16
// tiles; rows are a multiple of 16 tiles here and the longer units are used only when the rows are a multiple of
// them, so a unit lies inside one row) writes its own partial weight
// gradient, in accumulator layout, and k_gp_combine forms sum_u c_row(u) * partial[u].  Against the three-pass form
// this saves the second evaluation of the v-chain (36 MFMAs and its splits), packing and unpacking the mask
// bits, the mask words and g in HBM, and a launch: 150 instead of 186 MFMAs, about 600 instead of 770 VALU
// instructions per tile.
constexpr int GP_PART = 48;          // floats per lane of a partial: W1 gradient 4, W2 gradient 32, W3 gradient 8, W4 gradient 4
// longest unit (of 64, 32, 16 tiles) that stays inside a row of tpr tiles (tpr is a multiple of SMALL = 16): fewer,
// larger partials for long rows
__host__ DEV unsigned gp_big_unit(unsigned tpr) { return tpr % 64 == 0 ? 64u : (tpr % 32 == 0 ? 32u : TileQueue::SMALL); }
__global__ __launch_bounds__(WG_WAVES * 64, WG_PER_CU) void k_gp_fused_mfma(const float *__restrict__ real, const float *__restrict__ fake,
                                                              const float *__restrict__ cond, const float *__restrict__ alpha,
                                                              int64_t N, int64_t L, const float *__restrict__ wb, float slope,
                                                              float *__restrict__ partial, float *__restrict__ tile_sq,
                                                              unsigned *__restrict__ queue) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // LDS_WORDS
    build_image(lds, wb);
    PRIO_V();
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the tile walk then runs on the scalar unit
    char *wbuf = reinterpret_cast<char *>(lds + IMG_WORDS + wave * WAVE_WORDS);
    Addr ad;
    Accum ac;
    make_addr(ad, wbuf, j, g);
    zero_accum(ac);
    {
        float *cst = lds + IMG_WORDS + wave * WAVE_WORDS + CONST_W;
        for (int k = lane; k < 256; k += 64) cst[k] = 0.f;     // columns 2, 3 of the g tile stay 0
        float *left = lds + IMG_WORDS + wave * WAVE_WORDS + LEFT_W;
        for (int k = lane; k < 16 * P64; k += 64) left[k] = 0.f;
        __builtin_amdgcn_wave_barrier();
    }
    f32x4 dw4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 w4 = quad(lds, Q_W4, lane);
    const int64_t ntiles = (N + 15) >> 4;
    const unsigned tpr = (unsigned)(L >> 4);      // tiles per row
    // inputs one tile ahead: branch-free, carried untouched (see k_critic_bwd_mfma)
    const float *src_a = g < 2 ? real + g : cond + (g - 2), *src_b = g < 2 ? fake + g : cond + (g - 2);
    struct Raw { float al, a, b; };
    auto fetch = [&](int64_t tile) {
        const int64_t n = tile * 16 + j;
        const int64_t m = n < N ? n : 0;
        Raw r;
        r.al = alpha[(int64_t)((unsigned)(tile < ntiles ? tile : 0) / tpr)];
        r.a = src_a[2 * m];
        r.b = src_b[2 * m];
        return r;
    };
    FrontOps front;
    Fc3Ops o3;
    load_resident(front, lds, lane);
    load_resident(o3, lds, lane);
    load_ops(front, lds, lane);
    TileQueue tq;
    tq.init(queue, blockIdx.x * WG_WAVES + wave, gridDim.x * WG_WAVES, ntiles, gp_big_unit(tpr));
    int64_t tile_next = tq.tile();
    Raw raw_next = fetch(tile_next);
    while (tile_next < ntiles) {
        const int64_t tile = tile_next;
        const int64_t n = tile * 16 + j;
        const bool valid = n < N;
        const float in = valid ? (g < 2 ? raw_next.al * raw_next.a + (1.f - raw_next.al) * raw_next.b : raw_next.a) : 0.f;
        const unsigned unit_now = tq.unit;
        const bool unit_ends = tq.in_unit + 1 == tq.len;      // wave-uniform
        tq.advance(lane);
        tile_next = tq.tile();
        raw_next = fetch(tile_next);
        Acts a;
        TopOps top;
        BotOps bot;
        InOps ino;
        P1R p1r;
        // forward at the interpolate; the previous tile's v1 x g (parked a whole tile ago) beside it
        forward_front(a, lds, lane, in, slope, front, [&] {
            load_ops(o3, lds, lane);
            read_p1(p1r, ad);
        });
        mfma_p1(ac, p1r);
        forward_back<false, false>(a, lds, lane, slope, o3, [&] { load_ops(top, lds, lane); });
        const f32x4 (&m1)[4] = a.m1;
        const f32x4 (&m2)[2] = a.m2;
        const f32x4 m3 = a.m3;
        // v-chain: input gradient of the score; the terms of v3 and v2 stay parked for the outer products
        const f32x4 v3 = w4 * m3;
        f32x4 v2[2], v1[4];
        B3 sv2;
        chain_top(v3, m2, v2, sv2, top, [&] { load_ops(bot, lds, lane); },
                  [&](const unsigned (&bh)[2], const unsigned (&bm)[2], const unsigned (&bl)[2]) { park_single(ad, S_D3, bh, bm, bl); });
        bf16x8 a1u[4];           // (FC1 operands once more, for the u-chain: kept from the forward they would cost 16 registers)
        chain_bottom(lds, lane, v2, sv2, m1, v1, bot, [&] {
            load_ops(ino, lds, lane);
            if (CRITIC_FC1_BF16) {
#pragma unroll
                for (int t = 0; t < 4; ++t) a1u[t] = bquad(lds, QB_A1B + t, lane);
            }
        }, [&](const B3 &b) { park_b3(ad, S_D2, b); });      // after read_p1: inside the p1 tile
        PRIO_V();
        float g0, g1;
        to_input(v1, g0, g1, ino);
        if (!valid) { g0 = 0.f; g1 = 0.f; }
        {
            const float sq = sum_over_j1(g0 * g0 + g1 * g1);
            if (lane == 0) tile_sq[tile] = sq;
        }
        // u-chain with the unscaled g: du1 = (W1[:, 0:2] g) * D1, du2 = (W2 du1) * D2, dv3 = W3 du2
        const float ghb = g == 0 ? g0 : (g == 1 ? g1 : 0.f);
        f32x4 du1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            du1[t] = (CRITIC_FC1_BF16 ? mfma_bf(a1u[t], fc1_operand(ghb, 0u), f32x4{0.f, 0.f, 0.f, 0.f})
                                      : mfma4(front.a1[t], ghb, f32x4{0.f, 0.f, 0.f, 0.f})) * m1[t];
        f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
        unsigned no_bits = 0u;
        f32x4 unused_m[4];
        OpL du2l[2], du1l[2];
        OpR v3r, v2r[2];
        fc2_block<false, false>(z0, z1, du1, front.w, unused_m, no_bits, lds, lane, slope, [&] {
            load_ops(o3, lds, lane);
            read_pair_op(v2r[0], ad, S_D2, 0);
            read_pair_op(v2r[1], ad, S_D2, 1);
            __builtin_amdgcn_wave_barrier();
        }, nullptr, [&](int half, const B3 &b) { park_b3(ad, S_H1 + 3 * SLOT_B * half, b); });
        // p2 = du1 x v2 first: the slots of v2 are then free for the terms of du2
        read_pair_op(du1l[0], ad, S_H1, 0);
        read_pair_op(du1l[1], ad, S_H1, 1);
        FENCE();
        PRIO_M();
        outer3x2(ac.p2[0][0], ac.p2[0][1], du1l[0], v2r[0], v2r[1]);
        FENCE();
        read_pair_op(du1l[0], ad, S_H1 + 3 * SLOT_B, 0);
        FENCE();
        outer3x2(ac.p2[1][0], ac.p2[1][1], du1l[1], v2r[0], v2r[1]);
        FENCE();
        read_pair_op(du1l[1], ad, S_H1 + 3 * SLOT_B, 1);
        FENCE();
        outer3x2(ac.p2[2][0], ac.p2[2][1], du1l[0], v2r[0], v2r[1]);
        FENCE();
        outer3x2(ac.p2[3][0], ac.p2[3][1], du1l[1], v2r[0], v2r[1]);
        PRIO_V();
        __builtin_amdgcn_wave_barrier();
        f32x4 du2[2] = {z0 * m2[0], z1 * m2[1]};
        FENCE();
        const f32x4 dv3 = fc3_product(du2[0], du2[1], o3.w, f32x4{0.f, 0.f, 0.f, 0.f}, [&](const B3 &b) {
            park_b3(ad, S_H2, b);                             // the slots of v2: their reads are issued
            read_pair_op(du2l[0], ad, S_H2, 0);
            read_pair_op(du2l[1], ad, S_H2, 1);
            read_single_op(v3r, ad, S_D3);
            __builtin_amdgcn_wave_barrier();
        });
        FENCE();
        dw4 += dv3 * m3;
        load_ops(front, lds, lane);           // next tile
        // p3 = du2 x v3
        PRIO_M();
        ac.p3[0] = mfma_bf(du2l[0].hl, v3r.lh, ac.p3[0]);
        ac.p3[1] = mfma_bf(du2l[1].hl, v3r.lh, ac.p3[1]);
        ac.p3[0] = mfma_bf(du2l[0].hm, v3r.mh, ac.p3[0]);
        ac.p3[1] = mfma_bf(du2l[1].hm, v3r.mh, ac.p3[1]);
        ac.p3[0] = mfma_bf(du2l[0].hm, v3r.hm, ac.p3[0]);
        ac.p3[1] = mfma_bf(du2l[1].hm, v3r.hm, ac.p3[1]);
        PRIO_V();
        __builtin_amdgcn_wave_barrier();
        // p1 = v1 x g: the g tile (column g < 2 of row j) and v1; runs beside the next tile's forward
        if (g < 2) *reinterpret_cast<float *>(ad.cst) = ghb;
        park_p1(ad, v1);
        if (unit_ends) {
            // the unit's partial sums leave the wave: the pending p1 first, then an empty g tile so that the next
            // tile's deferred p1 adds nothing
            P1R last;
            read_p1(last, ad);
            mfma_p1(ac, last);
            if (g < 2) *reinterpret_cast<float *>(ad.cst) = 0.f;
            __builtin_amdgcn_wave_barrier();
            float *out = partial + (size_t)unit_now * (GP_PART * 64) + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) out[64 * c] = ac.p1[c] + ac.p1b[c];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[64 * (4 + (2 * t + u) * 4 + r)] = ac.p2[t][u][r];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[64 * (36 + 4 * u + r)] = ac.p3[u][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) out[64 * (44 + r)] = dw4[r];
            zero_accum(ac);
            dw4 = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// grads (one slab row per workgroup, flat layout) = sum over this workgroup's units of c_row(unit) * partial[unit]
__global__ __launch_bounds__(256) void k_gp_combine(const float *__restrict__ partial, const float *__restrict__ rowc,
                                                    unsigned n_units, unsigned n_big, unsigned big, unsigned tpr,
                                                    float *__restrict__ slab) {
    __shared__ float row[C_SLAB];
    __shared__ float w4part[4 * 64];              // W4 gradient per lane, summed over the 16 samples of a lane group below
    constexpr int PER = GP_PART * 64 / 256;       // 12 values per thread
    float acc[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) acc[k] = 0.f;
    for (unsigned u = blockIdx.x; u < n_units; u += gridDim.x) {
        const unsigned first = u < n_big ? u * big : n_big * big + (u - n_big) * TileQueue::SMALL;
        const float c = rowc[first / tpr];
        const float *src = partial + (size_t)u * (GP_PART * 64) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < PER; ++k) acc[k] = __builtin_fmaf(c, src[256 * k], acc[k]);
    }
    for (int e = threadIdx.x; e < C_SLAB; e += 256) row[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int idx = 256 * k + threadIdx.x, part = idx >> 6, lane = idx & 63, j = lane & 15, g = lane >> 4;
        const float v = acc[k];
        if (part < 4) {
            if (part < 2) row[C_W1 + lane * 4 + part] = v;         // columns 2, 3 of W1 get no contribution
        } else if (part < 36) {
            const int q = part - 4, t = q >> 3, u = (q >> 2) & 1, r = q & 3;
            row[C_W2 + (16 * u + j) * 64 + 16 * t + 4 * g + r] = v;
        } else if (part < 44) {
            const int q = part - 36, u = q >> 2, r = q & 3;
            row[C_W3 + j * 32 + 16 * u + 4 * g + r] = v;
        } else {
            w4part[64 * (part - 44) + lane] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) {       // (g, r): the 16 samples of the lane group in a fixed order
        const int g = threadIdx.x >> 2, r = threadIdx.x & 3;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += w4part[64 * r + 16 * g + j];
        row[C_W4 + 4 * g + r] = sum;
    }
    __syncthreads();
    float *out = slab + (size_t)blockIdx.x * C_SLAB;
    for (int e = threadIdx.x; e < C_SLAB; e += 256) out[e] = row[e];
}
#endif

int grid_tiles(int64_t N, int per_cu, int waves = WAVES) {
    int64_t blocks = (((N + 15) >> 4) + waves - 1) / waves;
#ifdef ABL_ONE_WAVE
    per_cu = 1;   // experiment: one workgroup per CU, i.e. every wave has its SIMD to itself
#endif
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

// ---- launch interface used by critic.hip -------------------------------------------------------------
template <typename K>
static int allow_big_lds(K kernel, size_t bytes) {
    DPD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return DPD_OK;
}
constexpr size_t BIG_LDS = (size_t)LDS_WORDS * sizeof(float), IMG_LDS = (size_t)IMG_WORDS * sizeof(float);

static int init_attrs() {
    static PerDeviceOnce once;        // the attribute is per device: a process that drives several GPUs sets it on each
    int dev;
    if (!once.first_use(&dev)) return DPD_OK;
    int rc = allow_big_lds(k_critic_bwd_mfma<true, true>, BIG_LDS);
    rc = rc ? rc : allow_big_lds(k_critic_bwd_mfma<true, false>, BIG_LDS);
    rc = rc ? rc : allow_big_lds(k_gp_c_mfma<true>, BIG_LDS);
    rc = rc ? rc : allow_big_lds(k_gp_c_mfma<false>, BIG_LDS);
#if CRITIC_OUTER_BF16
    rc = rc ? rc : allow_big_lds(k_gp_fused_mfma, BIG_LDS);
#endif
    if (rc == DPD_OK) once.mark(dev);
    return rc;
}

int critic_mfma_forward(const float *x, const float *cond, int64_t N, const float *wb, float slope, float *score,
                        float *score_sum, hipStream_t st) {
    const int grid = grid_tiles(N, NOGRAD_PER_CU);
    hipLaunchKernelGGL(k_critic_fwd_mfma, dim3(grid), dim3(WAVES * 64), 0, st, x, cond, N, wb, slope, score, score_sum);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

int g_critic_deterministic = 0;   // dpd_set_option("critic_deterministic", 0 / 1)

// the tile counter of a launch: one zeroed word of the caller's workspace (NULL: fixed tile order)
static int tile_counter(unsigned *word, hipStream_t st, unsigned **out) {
    *out = nullptr;
    if (g_critic_deterministic || !word) return DPD_OK;
    DPD_HIP(hipMemsetAsync(word, 0, 2 * sizeof(unsigned), st));
    *out = word;
    return DPD_OK;
}

int critic_mfma_backward(const float *x, const float *cond, int64_t N, const float *wb, float slope, const float *gscore,
                         float gconst, float *grads, float *gx, float *score_sum, float *ws, hipStream_t st) {
    float2 *gx2 = reinterpret_cast<float2 *>(gx);
    int rc = init_attrs();
    if (rc) return rc;
    unsigned *queue;
    if (!grads) {
        // no slab in this variant: the workspace, when there is one, only holds the counter
        rc = tile_counter(reinterpret_cast<unsigned *>(ws), st, &queue);
        if (rc) return rc;
        hipLaunchKernelGGL((k_critic_bwd_mfma<false, true>), dim3(grid_tiles(N, NOGRAD_PER_CU)), dim3(WAVES * 64), IMG_LDS, st, x, cond, N, wb,
                           slope, gscore, gconst, gx2, nullptr, score_sum, queue);
        DPD_LAUNCH_CHECK();
        return DPD_OK;
    }
    const int grid = grid_tiles(N, WG_PER_CU, WG_WAVES);
    rc = tile_counter(reinterpret_cast<unsigned *>(ws + (size_t)512 * C_SLAB), st, &queue);   // behind the slab rows
    if (rc) return rc;
    if (gx)
        hipLaunchKernelGGL((k_critic_bwd_mfma<true, true>), dim3(grid), dim3(WG_WAVES * 64), BIG_LDS, st, x, cond, N, wb, slope,
                           gscore, gconst, gx2, ws, score_sum, queue);
    else
        hipLaunchKernelGGL((k_critic_bwd_mfma<true, false>), dim3(grid), dim3(WG_WAVES * 64), BIG_LDS, st, x, cond, N, wb, slope,
                           gscore, gconst, gx2, ws, score_sum, queue);
    DPD_LAUNCH_CHECK();
    return gen_reduce_slab_generic(ws, grid, C_SLAB, C_NP, grads, score_sum, 1.f, st);
}

// ws layout: slab (512 rows) | mask words [ntiles * 64] | g [N] float2 | tile_sq [ntiles] | rowc [B] | pen [B] |
//            tile counters of pass A and pass C (inside the 1024 spare bytes)
int g_gp_fused = 1;               // dpd_set_option("gp_fused", 0 / 1): the one-pass form where the shape allows it

static size_t gp_fused_ws_bytes(int64_t B, int64_t L);
// rows of a multiple of 16 tiles: no hand-out unit crosses a row (units of 32 tiles only if rows are a multiple of 32).
// The per-unit partials of short rows (units of 16 tiles: 48 bytes per sample) can need more than the documented
// workspace, dpd_workspace_bytes(B * L): such a call takes the three-pass form, which always fits.
static bool gp_fused_shape(int64_t B, int64_t L, size_t ws_bytes) {
#if CRITIC_OUTER_BF16
    return g_gp_fused && L >= 256 && (L & 255) == 0 && gp_fused_ws_bytes(B, L) <= ws_bytes;
#else
    return false;
#endif
}
#if CRITIC_OUTER_BF16
static void gp_units(int64_t nt, int64_t L, unsigned *n_big, unsigned *n_units) {
    const unsigned big = gp_big_unit((unsigned)(L >> 4));
    *n_big = (unsigned)((nt - (nt >> 3)) / big);
    *n_units = *n_big + (unsigned)((nt - (int64_t)*n_big * big + TileQueue::SMALL - 1) / TileQueue::SMALL);
}
#endif
// fused form: slab (512 rows) | partials [units][48][64] | tile_sq [ntiles] | rowc [B] | pen [B] | combine slab (256 rows) | counter
static size_t gp_fused_ws_bytes(int64_t B, int64_t L) {
    const int64_t N = B * L, nt = (N + 15) >> 4;
    unsigned n_big = 0, n_units = 0;
#if CRITIC_OUTER_BF16
    gp_units(nt, L, &n_big, &n_units);
#endif
    return (size_t)512 * C_SLAB * 4 + (size_t)n_units * 48 * 64 * 4 + (size_t)nt * 4 + (size_t)B * 8 + (size_t)256 * C_SLAB * 4 + 1024;
}
// what the call will use of a workspace of ws_bytes (the one-pass form only if it fits)
size_t critic_mfma_gp_ws_bytes(int64_t B, int64_t L, size_t ws_bytes) {
    const int64_t N = B * L, nt = (N + 15) >> 4;
    const size_t three_pass = (size_t)512 * C_SLAB * 4 + (size_t)nt * 64 * 4 + (size_t)N * 8 + (size_t)nt * 4 + (size_t)B * 8 + 1024;
    return gp_fused_shape(B, L, ws_bytes) ? gp_fused_ws_bytes(B, L) : three_pass;
}

int critic_mfma_gp(const float *real, const float *fake, const float *cond, const float *alpha, int64_t B, int64_t L,
                   const float *wb, float slope, float *gp, float *rownorm, float *grads, void *ws, size_t ws_bytes,
                   hipStream_t st) {
    const int64_t N = B * L, nt = (N + 15) >> 4;
    char *base = reinterpret_cast<char *>(ws);
    float *slab = reinterpret_cast<float *>(base);
    size_t off = (size_t)512 * C_SLAB * 4;
    unsigned *maskw = reinterpret_cast<unsigned *>(base + off); off += (size_t)nt * 64 * 4;
    float2 *g = reinterpret_cast<float2 *>(base + off); off += (size_t)N * 8;
    float *tile_sq = reinterpret_cast<float *>(base + off); off += (size_t)nt * 4;
    float *rowc = reinterpret_cast<float *>(base + off); off += (size_t)B * 4;
    float *pen = reinterpret_cast<float *>(base + off); off += (size_t)B * 4;
    unsigned *counters = reinterpret_cast<unsigned *>(base + ((off + 63) & ~(size_t)63));
    const bool row1 = (L == 1);
    int rc = init_attrs();
    if (rc) return rc;
    unsigned *queue;
#if CRITIC_OUTER_BF16
    if (gp_fused_shape(B, L, ws_bytes)) {
        unsigned n_big, n_units;
        gp_units(nt, L, &n_big, &n_units);
        size_t o = (size_t)512 * C_SLAB * 4;
        float *partial = reinterpret_cast<float *>(base + o); o += (size_t)n_units * 48 * 64 * 4;
        float *sq = reinterpret_cast<float *>(base + o); o += (size_t)nt * 4;
        float *rc_row = reinterpret_cast<float *>(base + o); o += (size_t)B * 4;
        float *pen_row = reinterpret_cast<float *>(base + o); o += (size_t)B * 4;
        float *cslab = reinterpret_cast<float *>(base + o); o += (size_t)256 * C_SLAB * 4;
        rc = tile_counter(reinterpret_cast<unsigned *>(base + ((o + 63) & ~(size_t)63)), st, &queue);
        if (rc) return rc;
        const int gf = grid_tiles(N, WG_PER_CU, WG_WAVES);
        hipLaunchKernelGGL(k_gp_fused_mfma, dim3(gf), dim3(WG_WAVES * 64), BIG_LDS, st, real, fake, cond, alpha, N, L, wb, slope,
                           partial, sq, queue);
        DPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_gp_rows_tiles, dim3(grid_for(B, 1, 4096)), dim3(64), 0, st, sq, B, L / 16, rownorm, rc_row, pen_row);
        DPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, pen_row, (int)B, gp, 1.0f / (float)B);
        DPD_LAUNCH_CHECK();
        const int gcomb = n_units < 256 ? (int)n_units : 256;
        hipLaunchKernelGGL(k_gp_combine, dim3(gcomb), dim3(256), 0, st, partial, rc_row, n_units, n_big,
                           gp_big_unit((unsigned)(L / 16)), (unsigned)(L / 16), cslab);
        DPD_LAUNCH_CHECK();
        return gen_reduce_slab_generic(cslab, gcomb, C_SLAB, C_NP, grads, nullptr, 1.f, st);
    }
#endif
    rc = tile_counter(counters, st, &queue);
    if (rc) return rc;
    unsigned *queue_c = queue ? queue + 1 : nullptr;
    const int ga = grid_tiles(N, NOGRAD_PER_CU);
    if (row1)
        hipLaunchKernelGGL((k_gp_a_mfma<true>), dim3(ga), dim3(WAVES * 64), 0, st, real, fake, cond, alpha, N, L, wb, slope,
                           g, maskw, tile_sq, queue);
    else
        hipLaunchKernelGGL((k_gp_a_mfma<false>), dim3(ga), dim3(WAVES * 64), 0, st, real, fake, cond, alpha, N, L, wb, slope,
                           g, maskw, tile_sq, queue);
    DPD_LAUNCH_CHECK();
    if (row1)
        hipLaunchKernelGGL(k_gp_rows_single, dim3(grid_for(B, 256, 1024)), dim3(256), 0, st, g, B, rownorm, rowc, pen);
    else
        hipLaunchKernelGGL(k_gp_rows_tiles, dim3(grid_for(B, 1, 4096)), dim3(64), 0, st, tile_sq, B, L / 16, rownorm, rowc, pen);
    DPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, pen, (int)B, gp, 1.0f / (float)B);
    DPD_LAUNCH_CHECK();
    const int gc = grid_tiles(N, WG_PER_CU, WG_WAVES);
    if (row1)
        hipLaunchKernelGGL((k_gp_c_mfma<true>), dim3(gc), dim3(WG_WAVES * 64), BIG_LDS, st, g, maskw, rowc, N, L, wb, slope, slab, queue_c);
    else
        hipLaunchKernelGGL((k_gp_c_mfma<false>), dim3(gc), dim3(WG_WAVES * 64), BIG_LDS, st, g, maskw, rowc, N, L, wb, slope, slab, queue_c);
    DPD_LAUNCH_CHECK();
    return gen_reduce_slab_generic(slab, gc, C_SLAB, C_NP, grads, nullptr, 1.f, st);
}
