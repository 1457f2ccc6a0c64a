// Generator kernels, MFMA family (gfx950): tap assembly + 30->32->16->2 forward (+ backward) fused
// per 16-sample tile.  Default build: every product on v_mfma_f32_16x16x32_bf16 at fp32 accuracy -- the forward layers
// of the QAT generator as exact digit products (see mfma_digits), those of the float generator and every backward
// product as three-way split products (see mfma_split3, outer3).  The v_mfma_f32_16x16x4_f32 formulation described
// below is the code's other path (GEN_QAT_DIGITS = 0, GEN_F32_SPLIT3 = 0); geometry and schedule are common to both.
// Reference semantics as in gen_valu.hip.
//
// Geometry.  A wave owns a chunk of 64 consecutive output samples of one IQ row (4 tiles of 16).
// Lane l = (g = l >> 4, j = l & 15).  Every product is computed transposed, D = W * X^T, so that
// the sample index sits on the MFMA column (lane j) and the neuron index on the rows: a 16x16
// result tile leaves lane (g, j) holding neurons 4g..4g+3 of sample j, which is exactly the B
// operand layout (k = lane group) of the next layer's MFMA; no lane movement between layers.
//
// FC1 contracts over 32 "slots" (30 features + 2 zeros) in this kernel's own order
//   slot = 6 * type + tap,  type 0..4 = I, Q, |x|, |x|^2, |x|^4,  tap 0..5
// with the W1 columns gathered to match.  The five per-time-point series of a chunk live in LDS
// (69 points each); a feature operand is one ds_read_b32 at series[type][j + 5 - tap].
//
// The weight gradient contracts over samples, so samples must move from lanes to the MFMA K axis:
// gz1, h1 and gz2 go through XOR-swizzled wave-private LDS tiles ([sample][neuron]); the x operand
// of dW1 is read straight from the series.  FC3 (2 outputs), its gradient and all bias gradients
// stay on the VALU with per-lane accumulators that are reduced once at the end of the kernel.
// All LDS traffic is wave-private, so there are no workgroup barriers: LDS operations of one wave
// execute in program order.
#include "gen_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 4;
#ifndef BWD_WAVES_PER_SIMD
#define BWD_WAVES_PER_SIMD 2        // measured: 2 resident waves per SIMD (<= 256 registers, no spills) beat 3 with spills
#endif
#ifndef FWD_WAVES_PER_SIMD
#define FWD_WAVES_PER_SIMD 4        // forward only: <= 128 registers (measured: 3..5 within 4 %, 4 best, no spills)
#endif
constexpr int CHUNK = 64;           // samples per wave iteration
constexpr int SER_STRIDE = 71;      // words between series (71 = 7 mod 32: bank-disjoint operand reads)
constexpr int SER_BASE0 = 5;
constexpr int ZERO_W = 360;         // 64 words that always hold 0: operand of the padding slot 31
                                    // (their reads add the same tile / sample offsets as real slots)
constexpr int ONE_W = 424;          // 64 words that always hold 1: slot 30 is the bias input (its W1 column is b1,
                                    // its dW1 row is the bias gradient)
constexpr int TG_W = 488;           // 64 x float2: the chunk's target / upstream gradient, sample-major (backward only)
constexpr int PAC_W = 616;          // 48 words: memory-polynomial coefficients [tap 0..5][re k0..3 | im k0..3] (tap 5: zeros), fused chain backward
constexpr int SER_WORDS = PAC_W + 48;   // 664
constexpr int P32 = 48;             // row pitch of the transposition tiles (16 mod 32)
// backward: two tile buffers per wave (two tiles are in flight), each = [16 samples][gz1 (32) | gz2 (16)] and
// [16 samples][h1 (32) | unused (16)], both with the 48-word pitch and the XOR swizzle of swz()
constexpr int TILE_W = 2 * 16 * P32;             // words per tile buffer (1536)
constexpr int GZ_W = SER_WORDS;                  // + TILE_W * buffer
constexpr int H1_W = GZ_W + 16 * P32;
constexpr int BWD_WORDS = SER_WORDS + 2 * TILE_W;   // 3736 words per wave
static_assert(WAVES * BWD_WORDS >= WAVES * G_SLAB, "the gradient rows are combined in the staging LDS");
static_assert((size_t)WAVES * BWD_WORDS * sizeof(float) <= 65536, "static LDS of the backward kernel: 58.4 KiB, two workgroups per CU");
// backward in the digit form: the five series a second time, as digit words (FC1 operands), between the series and the
// tile buffers; the series themselves then hold the quantised integers as floats (the x operand of dW1)
constexpr int DSER_WORDS = 356;
// split form: words B lie a multiple of 64 words behind words A, so that one ds_read2st64_b32 fetches both
#ifndef GEN_SPB_FWD
#define GEN_SPB_FWD 704
#endif
constexpr int SPB_FWD = GEN_SPB_FWD, SPB_BWD = 384;
constexpr int FWD_ROW_SP3 = (SER_BASE0 + SPB_FWD + DSER_WORDS + 3) / 4 * 4;   // words per wave of the forward kernel in the split form (1068)
static_assert(SER_BASE0 + SPB_FWD >= SER_WORDS && SER_BASE0 + SPB_FWD + DSER_WORDS <= FWD_ROW_SP3, "words B behind the series region");
static_assert((size_t)WAVES * (BWD_WORDS + DSER_WORDS) * sizeof(float) <= 65536, "63.9 KiB, two workgroups per CU");
// (the split form of the float backward keeps two more sets, words A and B: 69.9 KiB of static LDS, two workgroups per CU)
static_assert((size_t)2 * WAVES * (BWD_WORDS + SPB_BWD + DSER_WORDS) * sizeof(float) <= 160 * 1024, "two workgroups per CU");

#ifdef ABL_NO_MFMA      // experiment (timing only, results are wrong): every matrix instruction dropped, its dependencies kept
DEV f32x4 mfma4(float a, float b, f32x4 c) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return c; }
#else
DEV f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
#endif

// reference feature index (models/tdnn_generator.py:176-193) of a slot; -1 for the two zero slots
DEV int slot_ref(int slot) {
    if (slot >= 30) return -1;
    int type = slot / 6, tap = slot - 6 * type;
    if (type == 0) return tap == 0 ? 0 : 20 + 2 * (tap - 1);
    if (type == 1) return tap == 0 ? 1 : 21 + 2 * (tap - 1);
    return 2 + 6 * (type - 2) + tap;
}
// word offset inside the series region of (slot, sample 0 of tile 0); add sample index and 16 * tile
DEV int slot_word(int slot) {
    if (slot >= 30) return slot == 30 ? ONE_W : ZERO_W;
    int type = slot / 6, tap = slot - 6 * type;
    return SER_BASE0 + SER_STRIDE * type + 5 - tap;
}
DEV int swz(int s) { return 4 * ((s >> 1) & 3); }
struct Lane {
    int lane, j, g;
};

// The chunk's 69 IQ points as loaded from HBM: point lane in p[0], point 64 + lane (lanes 0..4) in p[1].
struct ChunkIQ {
    float2 p[2];
};
DEV void load_chunk_iq(ChunkIQ &c, const float2 *__restrict__ iqrow, int64_t t0, int64_t S, int lane) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        int64_t idx = t0 + lane + 64 * rep;
        c.p[rep] = make_float2(0.f, 0.f);
        if ((rep == 0 || lane < 5) && idx < S) c.p[rep] = iqrow[idx];
    }
}
template <bool ASM>
DEV f32x4 fq16x4(f32x4 v, QS k);
DEV f32x4 q16x4(f32x4 v, QS k);
DEV float q16(float v, QS k);
DEV float digit_word(float q);
// Derive the five series from the loaded points and park them in LDS.
template <bool QAT>
DEV void commit_series(float *ser, const ChunkIQ &c, int lane, QS s0) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        int tau = lane + 64 * rep;
        if (rep == 1 && lane >= 5) break;
        float2 v = c.p[rep];
        float e = sqrtf(v.x * v.x + v.y * v.y);
        float e2 = e * e;
        float e4 = e2 * e2;
        if (QAT) {
            const f32x4 qv = fq16x4<true>(f32x4{v.x, v.y, e, e2}, s0);
            v.x = qv[0]; v.y = qv[1]; e = qv[2]; e2 = qv[3];
            e4 = fq16(e4, s0);
        }
        float *p = ser + SER_BASE0 + tau;
        p[0] = v.x;
        p[SER_STRIDE] = v.y;
        p[2 * SER_STRIDE] = e;
        p[3 * SER_STRIDE] = e2;
        p[4 * SER_STRIDE] = e4;
    }
}
// Digit form of the backward: the quantised integers as floats in the series (dW1 contracts them with gz1 and is scaled
// by the input scale once, at the end of the kernel), their digit words in the second set of series at dser.
DEV void commit_series_digits(float *ser, float *dser, const ChunkIQ &c, int lane, QS s0) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        int tau = lane + 64 * rep;
        if (rep == 1 && lane >= 5) break;
        const float2 v = c.p[rep];
        const float e = sqrtf(v.x * v.x + v.y * v.y);
        const float e2 = e * e;
        const float e4 = e2 * e2;
        const f32x4 qv = q16x4(f32x4{v.x, v.y, e, e2}, s0);
        const float q4 = q16(e4, s0);
        float *p = ser + SER_BASE0 + tau, *d = dser + tau;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float q = k < 4 ? qv[k] : q4;
            p[k * SER_STRIDE] = q;
            d[k * SER_STRIDE] = digit_word(q);
        }
    }
}
DEV void split_words(float x, float &a, float &b);
// Split form of the float backward: the values themselves in the series (the x operand of dW1), their words A and B (see
// split_words) in two more sets of series.
DEV void commit_series_split(float *ser, float *aser, float *bser, const ChunkIQ &c, int lane) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        int tau = lane + 64 * rep;
        if (rep == 1 && lane >= 5) break;
        const float2 v = c.p[rep];
        const float e = sqrtf(v.x * v.x + v.y * v.y);
        const float e2 = e * e;
        const float x[5] = {v.x, v.y, e, e2, e2 * e2};
        float *p = ser + SER_BASE0 + tau;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float a, b;
            split_words(x[k], a, b);
            p[k * SER_STRIDE] = x[k];
            aser[tau + k * SER_STRIDE] = a;
            bser[tau + k * SER_STRIDE] = b;
        }
    }
}
// One time point: derive its five series values and park them at position tau of the chunk's series.
// DIG (QAT forward on the bf16 matrix cores): a series word holds the two bf16 digits of the quantised integer.
DEV void split_words(float x, float &a, float &b);
template <bool QAT, bool DIG = false, bool SP3 = false>
DEV void commit_point(float *ser, float2 v, int tau, QS s0) {
    float e = sqrtf(v.x * v.x + v.y * v.y);
    float e2 = e * e;
    float e4 = e2 * e2;
    if (SP3) {           // float generator, split form: word A in the series, word B in the second set behind them
        const float x[5] = {v.x, v.y, e, e2, e4};
        float *p = ser + SER_BASE0 + tau, *p2 = p + SPB_FWD;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float a, b;
            split_words(x[k], a, b);
            p[k * SER_STRIDE] = a;
            p2[k * SER_STRIDE] = b;
        }
        return;
    }
    if (DIG) {
        const f32x4 qv = q16x4(f32x4{v.x, v.y, e, e2}, s0);
        v.x = digit_word(qv[0]); v.y = digit_word(qv[1]); e = digit_word(qv[2]); e2 = digit_word(qv[3]);
        e4 = digit_word(q16(e4, s0));
    } else if (QAT) {
        const f32x4 qv = fq16x4<true>(f32x4{v.x, v.y, e, e2}, s0);
        v.x = qv[0]; v.y = qv[1]; e = qv[2]; e2 = qv[3];
        e4 = fq16(e4, s0);
    }
    float *p = ser + SER_BASE0 + tau;
    p[0] = v.x;
    p[SER_STRIDE] = v.y;
    p[2 * SER_STRIDE] = e;
    p[3 * SER_STRIDE] = e2;
    p[4 * SER_STRIDE] = e4;
}
struct FwdWeights {
    float A1[2][8];   // W1 tile t, step s: lane (i, g) holds W1[16t + i][slot 4s + g]
    float A2[2][4];   // W2[i][16t + 4g + r]
    f32x4 bias2;      // b2[4g + r]
};

DEV void load_fwd_weights(FwdWeights &w, const float *__restrict__ wq, int j, int g) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int ref = slot_ref(4 * s + g);
            w.A1[t][s] = ref >= 0 ? wq[G_W1 + (16 * t + j) * 30 + ref] : (4 * s + g == 30 ? wq[G_B1 + 16 * t + j] : 0.f);
        }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) w.A2[t][r] = wq[G_W2 + j * 32 + 16 * t + 4 * g + r];
#pragma unroll
    for (int r = 0; r < 4; ++r) w.bias2[r] = wq[G_B2 + 4 * g + r];
}

// Digit form of the forward weights (see the note on the QAT forward on the bf16 matrix cores below).  The prepared
// weights are q * row scale with q an integer of at most 16 bits; the row scale is recovered as the row's largest
// magnitude over 32767 (equal to the one that made it, up to one rounding, which cannot move rint(w / scale)).
// A1h / A1l[t]: lane (g, j) holds the hi / lo digits of row 16 t + j at slots 8 g .. 8 g + 7 (slot = 6 * series + tap,
// slots 30 and 31 zero); A2h / A2l: row j at k = (e < 4: neuron 4 g + e, else neuron 16 + 4 g + e - 4), which is what a
// lane holds of h1.  sc1 / b1 [t][r], sc2 / b2 [r]: row scale times input scale, and bias, of the output neurons
// 16 t + 4 g + r the lane holds of a result.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct FwdDigits {
    u32x4 A1h[2], A1l[2], A2h, A2l;
    f32x4 sc1[2], b1[2], sc2, b2;
};
DEV void digit_pair(float q0, float q1, unsigned &hi, unsigned &lo);
template <bool FC2>
DEV void load_fwd_digits(FwdDigits &w, const float *__restrict__ wq, float *scratch, int lane, int j, int g, float s_in,
                         float s_h1) {
    // row scales of W1 (32 rows) and W2 (16 rows): one row per lane, through the wave's LDS
    {
        float am = 0.f;
        if (lane < 32) {
            for (int k = 0; k < 30; ++k) am = fmaxf(am, fabsf(wq[G_W1 + lane * 30 + k]));
        } else if (lane < 48) {
            for (int k = 0; k < 32; ++k) am = fmaxf(am, fabsf(wq[G_W2 + (lane - 32) * 32 + k]));
        }
        scratch[lane] = scale_of(am);
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 16 * t + j;
        const float sw = scratch[n];
        float q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ref = slot_ref(8 * g + e);
            q[e] = ref >= 0 ? rintf(wq[G_W1 + n * 30 + ref] / sw) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned hi, lo;
            digit_pair(q[2 * i], q[2 * i + 1], hi, lo);
            w.A1h[t][i] = hi;
            w.A1l[t][i] = lo;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            w.sc1[t][r] = scratch[16 * t + 4 * g + r] * s_in;
            w.b1[t][r] = wq[G_B1 + 16 * t + 4 * g + r];
        }
    }
    if (FC2) {
        const float sw = scratch[32 + j];
        float q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int m = e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4);
            q[e] = rintf(wq[G_W2 + j * 32 + m] / sw);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned hi, lo;
            digit_pair(q[2 * i], q[2 * i + 1], hi, lo);
            w.A2h[i] = hi;
            w.A2l[i] = lo;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            w.sc2[r] = scratch[32 + 4 * g + r] * s_h1;
            w.b2[r] = wq[G_B2 + 4 * g + r];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// tanh to fp32 rounding level: odd series below 0.1 (next term 62/2835 x^9 < 2e-11), above it
// (1 - t) / (1 + t) with t = exp(-2|x|) (no cancellation left: 1 - t >= 0.18)
DEV float tanh_f32(float x) {
    const float ax = fabsf(x);
    const float x2 = x * x;
    const float small = x * __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
    const float t = __builtin_amdgcn_exp2f(ax * -2.885390081777927f);   // exp(-2|x|)
    const float big = copysignf((1.f - t) * __builtin_amdgcn_rcpf(1.f + t), x);
    return ax < 0.1f ? small : big;
}

// FC3 of a tile pair on the VALU: 2 outputs x 16 neurons fill an eighth of a 16x16x4 MFMA tile, while
// the per-lane partial sums (this lane group's four h2 neurons) cost 16 FMAs for both tiles and the
// sum over the four lane groups takes three row swaps.  Lane (g, j) ends up with z3[g & 1] of sample j
// of tile g >> 1.  w3[k][r] = W3[k][4g + r]; b3k = b3[g & 1].
typedef float f32x2 __attribute__((ext_vector_type(2)));
// quad arithmetic as two packed instructions (the compiler leaves about half of such quads as four scalar
// instructions; every VALU instruction costs the fp32 MFMAs issue time)
// ASM = false wherever bf16 MFMAs may be in flight: an instruction inside an asm statement is invisible to the
// compiler's hazard handling, and a VALU write to a register that a matrix instruction still executing reads or
// writes needs wait states that nobody then inserts (seen as sporadic wrong values in the second tile of a pair; the
// fp32 MFMA never showed it because nothing issues while one executes).
template <bool ASM = true>
DEV f32x4 mul4(f32x4 a, f32x4 b) {
#ifdef NO_PK_ASM
    return a * b;
#endif
    if (!ASM) return a * b;
    f32x2 lo, hi;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(f32x2{a[0], a[1]}), "v"(f32x2{b[0], b[1]}));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(f32x2{a[2], a[3]}), "v"(f32x2{b[2], b[3]}));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
// quantise-dequantise of a quad, the arithmetic of fq16(float, QS) element by element with its three
// multiplies and two FMAs packed: 3 VALU instructions per element instead of 6
template <bool ASM = true>
DEV f32x4 fq16x4(f32x4 v, QS k) {
    const f32x2 inv = {k.inv, k.inv}, sc = {k.s, k.s};
    f32x4 out;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f32x2 h = {v[2 * i], v[2 * i + 1]};
        f32x2 q0, t, q, r;
        if (!ASM) {
            q0 = h * inv;
            q = __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, sc, h), inv, q0);
            q[0] = rintf(q[0]);
            q[1] = rintf(q[1]);
            r = q * sc;
            out[2 * i] = r[0];
            out[2 * i + 1] = r[1];
            continue;
        }
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(h), "v"(inv));
        asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(t) : "v"(q0), "v"(sc), "v"(h));
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(q) : "v"(t), "v"(inv), "v"(q0));
        // no clamp: the scale comes from the abs-max of these very values (or of a superset, after a max-reduce
        // over ranks), so |v / s| <= 32767 up to one rounding and the nearest integer is within the 16-bit range
        q[0] = rintf(q[0]);
        q[1] = rintf(q[1]);
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(q), "v"(sc));
        out[2 * i] = r[0];
        out[2 * i + 1] = r[1];
    }
    return out;
}
// ---- QAT forward on the bf16 matrix cores: exact 16-bit digits ---------------------------------------------------
// A fake-quantised value is an integer of at most 16 bits times its tensor's (or row's) scale, and such an integer
// is the sum of two bf16 numbers exactly: hi = its upper 8 significant bits (the upper 16 bits of the fp32 pattern),
// lo = the rest (at most 8 bits).  So a layer of the QAT generator is an integer matrix product that the bf16 matrix
// cores evaluate with four MFMAs of K = 32 per 16 x 16 tile (lo.lo, hi.lo, lo.hi, hi.hi into one fp32 accumulator,
// small terms first: every product is exact, the accumulation rounds like any fp32 sum), and the two scales and
// the bias enter once per output value: z = acc * (row scale * input scale) + b.  Against the fp32 MFMA form (GEN_QAT_DIGITS
// = 0: 24 v_mfma_f32_16x16x4_f32 per tile at 32 cycles each, during which nothing else issues) this is 12
// v_mfma_f32_16x16x32_bf16 at 16 cycles with VALU work issuing beside them; the quantiser's last multiply is
// replaced by the digit split.  Same mathematical value as the reference's sum of products of dequantised numbers,
// rounded in fewer places.  The fused training kernels use the same form for FC1 and FC2 of their forward half (identical
// operand order, so the pre-activations equal those of the abs-max passes bit for bit); their weight gradients contract
// the integers (series and h1: the digits are the two terms of the left factor) and take the tensor scale once at the
// end of the kernel.
#ifndef GEN_QAT_DIGITS
#define GEN_QAT_DIGITS 1
#endif
#ifndef GEN_SPLIT_CVT
#define GEN_SPLIT_CVT 1      // the last term of a split through v_cvt_pk_bf16_f32 (exact here) instead of a byte permute
#endif
#ifdef ABL_NO_MFMA
DEV f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return c; }
#else
DEV f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#endif
// c + a x b for operands given as digit pairs
DEV f32x4 mfma_digits(u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl, f32x4 c) {
    c = mfma_bf(al, bl, c);
    c = mfma_bf(ah, bl, c);
    c = mfma_bf(al, bh, c);
    return mfma_bf(ah, bh, c);
}
// rint(v / s) of a quad: fq16x4 without its last multiply (the integers themselves); no asm, see mul4
DEV f32x4 q16x4(f32x4 v, QS k) {
    const f32x2 inv = {k.inv, k.inv}, sc = {k.s, k.s};
    f32x4 out;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f32x2 h = {v[2 * i], v[2 * i + 1]};
        const f32x2 q0 = h * inv;
        const f32x2 q = __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, sc, h), inv, q0);
        out[2 * i] = rintf(q[0]);
        out[2 * i + 1] = rintf(q[1]);
    }
    return out;
}
DEV float q16(float v, QS k) {
    const float q0 = v * k.inv;
    return rintf(__builtin_fmaf(__builtin_fmaf(-q0, k.s, v), k.inv, q0));
}
// two float integers -> their hi digits as a bf16 pair, their lo digits as a bf16 pair
DEV void digit_pair(float q0, float q1, unsigned &hi, unsigned &lo) {
    const unsigned u0 = __builtin_bit_cast(unsigned, q0), u1 = __builtin_bit_cast(unsigned, q1);
    hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float l0 = q0 - __builtin_bit_cast(float, u0 & 0xffff0000u), l1 = q1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
#if GEN_SPLIT_CVT       // (at most 8 significant bits: exact; see split_pair3)
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    lo = __builtin_bit_cast(unsigned, bf16x2{(__bf16)l0, (__bf16)l1});
#else
    lo = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, l1), __builtin_bit_cast(unsigned, l0), 0x07060302u);
#endif
}
// one float integer -> the word (hi digit << 16 | lo digit) a series holds per point
DEV float digit_word(float q) {
    const unsigned u = __builtin_bit_cast(unsigned, q);
    const float l = q - __builtin_bit_cast(float, u & 0xffff0000u);
    return __builtin_bit_cast(float, __builtin_amdgcn_perm(u, __builtin_bit_cast(unsigned, l), 0x07060302u));
}

// ---- float generator on the bf16 matrix cores: three-way split operands ------------------------------------------
// An fp32 value is the sum of three bf16 numbers exactly: hi = its upper 16 bits, mid = the upper 16 bits of what hi
// leaves, lo = the rest (at most 8 significant bits: truncation is exact at every step).  A product of two such sums to
// fp32 accuracy is six bf16 products (mm, lh, hl, mh, hm, hh into one fp32 accumulator, small terms first; the three
// dropped ones are below 2^-24 of the product), as in the critic kernels.  For FC1 (K = 32 slots) and FC2 (K = 32
// neurons) of the float generator that is 18 v_mfma_f32_16x16x32_bf16 per tile, with VALU work issuing beside them,
// instead of 24 v_mfma_f32_16x16x4_f32 during which nothing else issues; the biases start the accumulators.  A series
// point is kept in LDS as two words: A = hi << 16 | mid, B = the rest as a float (its upper half is lo).
#ifndef GEN_F32_SPLIT3
#define GEN_F32_SPLIT3 1
#endif
// GEN_SPLIT_PK = 1 writes the two subtractions of a split step as one two-element vector subtraction (v_pk_add_f32 with
// negated operand: 33 fewer instructions per chunk).  This compiler (ROCm 7.2) then emits byte permutes that take both
// halves from the first element (v_perm_b32 d, a, a) and the float generator fails its golden test: off.
#ifndef GEN_SPLIT_PK
#define GEN_SPLIT_PK 0
#endif

DEV void split_words(float x, float &a, float &b) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const float r1 = x - __builtin_bit_cast(float, u & 0xffff0000u);
    const unsigned v = __builtin_bit_cast(unsigned, r1);
    b = r1 - __builtin_bit_cast(float, v & 0xffff0000u);
    a = __builtin_bit_cast(float, __builtin_amdgcn_perm(u, v, 0x07060302u));
}
// two values -> their hi, mid and lo terms as bf16 pairs (first value in the lower half)
DEV void split_pair3(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
#if GEN_SPLIT_PK
    // the two subtractions of a step as one packed instruction
    const f32x2 r = f32x2{x0, x1} - f32x2{__builtin_bit_cast(float, u0 & 0xffff0000u), __builtin_bit_cast(float, u1 & 0xffff0000u)};
    const unsigned v0 = __builtin_bit_cast(unsigned, r[0]), v1 = __builtin_bit_cast(unsigned, r[1]);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const f32x2 q = r - f32x2{__builtin_bit_cast(float, v0 & 0xffff0000u), __builtin_bit_cast(float, v1 & 0xffff0000u)};
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q[1]), __builtin_bit_cast(unsigned, q[0]), 0x07060302u);
#else
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float q0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u), q1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
#if GEN_SPLIT_CVT
    // (the rest has at most 8 significant bits: the conversion is exact; written this way the compiler pairs the
    // subtractions above into packed instructions by itself)
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    l = __builtin_bit_cast(unsigned, bf16x2{(__bf16)q0, (__bf16)q1});
#else
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302u);
#endif
#endif
}
struct Split3 {          // eight values per lane as three bf16 operands
    u32x4 h, m, l;
};
DEV void split8(Split3 &o, const f32x4 &x0, const f32x4 &x1) {
    unsigned h[4], m[4], l[4];
    split_pair3(x0[0], x0[1], h[0], m[0], l[0]);
    split_pair3(x0[2], x0[3], h[1], m[1], l[1]);
    split_pair3(x1[0], x1[1], h[2], m[2], l[2]);
    split_pair3(x1[2], x1[3], h[3], m[3], l[3]);
    o.h = u32x4{h[0], h[1], h[2], h[3]};
    o.m = u32x4{m[0], m[1], m[2], m[3]};
    o.l = u32x4{l[0], l[1], l[2], l[3]};
}
// c + a x b to fp32 accuracy
DEV f32x4 mfma_split3(const Split3 &a, const Split3 &b, f32x4 c) {
    c = mfma_bf(a.m, b.m, c);
    c = mfma_bf(a.l, b.h, c);
    c = mfma_bf(a.h, b.l, c);
    c = mfma_bf(a.m, b.h, c);
    c = mfma_bf(a.h, b.m, c);
    return mfma_bf(a.h, b.h, c);
}
// Split form of the forward weights.  A1[t]: lane (g, j) holds row 16 t + j of W1 at slots 8 g .. 8 g + 7 (slot = 6 *
// series + tap; slots 30 and 31 zero); A2: row j of W2 at k = (e < 4: neuron 4 g + e, else neuron 16 + 4 g + e - 4), which
// is what a lane holds of h1; b1[t][r], b2[r]: biases of the output neurons a lane holds of a result (they start the
// accumulators).
struct FwdSplit {
    Split3 A1[2], A2;
    f32x4 b1[2], b2;
};
DEV void load_fwd_split(FwdSplit &w, const float *__restrict__ wq, int j, int g) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 16 * t + j;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ref = slot_ref(8 * g + e);
            v[e] = ref >= 0 ? wq[G_W1 + n * 30 + ref] : 0.f;
        }
        split8(w.A1[t], f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
#pragma unroll
        for (int r = 0; r < 4; ++r) w.b1[t][r] = wq[G_B1 + 16 * t + 4 * g + r];
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wq[G_W2 + j * 32 + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4))];
    split8(w.A2, f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
#pragma unroll
    for (int r = 0; r < 4; ++r) w.b2[r] = wq[G_B2 + 4 * g + r];
}
// the eight (A, B) word pairs a lane reads for FC1 -> the three operands
DEV void split_operand(Split3 &x, const float (&a)[8], const float (&b)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned a0 = __builtin_bit_cast(unsigned, a[2 * i]), a1 = __builtin_bit_cast(unsigned, a[2 * i + 1]);
        x.h[i] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
        x.m[i] = __builtin_amdgcn_perm(a1, a0, 0x05040100u);
        x.l[i] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b[2 * i + 1]), __builtin_bit_cast(unsigned, b[2 * i]), 0x07060302u);
    }
}

// ---- weight gradients on the bf16 matrix cores (backward of the bf16 forms) ---------------------------------------
// dW = sum over the samples of a tile of (left factor) x (right factor) contracts over the lane index of both factors.
// Their bf16 terms are parked in LDS as [sample][neuron] slots (16 rows of 64 bytes: chunk c of a row holds neurons
// 4c..4c+3 of an even 16-neuron tile and of the next one, 8 bytes each; the chunk index is XORed with row bits 1..2 so
// that a store group covers all banks) and read back with the transposing ds_read_b64_tr_b16: lane (kg, n) receives
// samples 4kg..4kg+3 of neuron n.  A K = 32 instruction takes two terms per operand (k-slots 0..3 of a lane group: four
// samples of one term, 4..7: the same samples of another), so the six products of a 16 x 16 tile are three MFMAs:
//   [h|l] x [l|h] = hl + lh,  [h|m] x [m|h] = hm + mh,  [h|m] x [h|m] = hh + mm      (left term first)
constexpr int SLOT_B = 1024;                // bytes of a slot
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
struct OpL { u32x4 hm, hl; };               // left factor (rows of the result): [h|m], [h|l]
struct OpR { u32x4 hm, mh, lh; };           // right factor (columns): [h|m], [m|h], [l|h]
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
DEV u32x2 tr_read(lds_char *p) { return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p)); }
DEV u32x4 cat(u32x2 a, u32x2 b) { return u32x4{a[0], a[1], b[0], b[1]}; }
// h, m, l: byte offsets of the 8-byte pieces of the three terms (l < 0: the factor has two terms only).  tr[0..2] hold
// the same address, the copies opaque to the compiler: a term that goes into several operands is read once per
// operand -- merged into one read, the value would have to be copied into place with VALU moves, and an LDS read is
// the cheaper instruction here.
struct TrAddr {
    lds_char *p[3];
};
DEV void read_op(OpL &o, const TrAddr &tr, int off, int h, int m, int l) {
    o.hm = cat(tr_read(tr.p[0] + off + h), tr_read(tr.p[0] + off + m));
    o.hl = l >= 0 ? cat(tr_read(tr.p[1] + off + h), tr_read(tr.p[0] + off + l)) : cat(tr_read(tr.p[1] + off + h), u32x2{0u, 0u});
}
DEV void read_op(OpR &o, const TrAddr &tr, int off, int h, int m, int l) {
    o.hm = cat(tr_read(tr.p[0] + off + h), tr_read(tr.p[0] + off + m));
    o.mh = cat(tr_read(tr.p[1] + off + m), tr_read(tr.p[1] + off + h));
    o.lh = cat(tr_read(tr.p[0] + off + l), tr_read(tr.p[2] + off + h));
}
DEV f32x4 outer3(f32x4 acc, const OpL &a, const OpR &b) {
    acc = mfma_bf(a.hl, b.lh, acc);
    acc = mfma_bf(a.hm, b.mh, acc);
    return mfma_bf(a.hm, b.hm, acc);
}

// LeakyReLU of a quad without a comparison: max(z, slope z) for slope <= 1, min(z, slope z) above,
// as the median of (z, slope z, +-inf); the product is the one the comparing form selects
template <bool ASM = true>
DEV f32x4 lrelu4(f32x4 z, float slope) {
    const f32x4 p = mul4<ASM>(z, f32x4{slope, slope, slope, slope});
    const float sel = slope <= 1.f ? __builtin_inff() : -__builtin_inff();
    f32x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = __builtin_amdgcn_fmed3f(z[r], p[r], sel);
    return h;
}
template <bool ASM = true>
DEV f32x4 add4(f32x4 a, f32x4 b) {
#ifdef NO_PK_ASM
    return a + b;
#endif
    if (!ASM) return a + b;
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(f32x2{a[0], a[1]}), "v"(f32x2{b[0], b[1]}));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(f32x2{a[2], a[3]}), "v"(f32x2{b[2], b[3]}));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
// (the partial sums of the two outputs travel as a pair: packed multiplies and FMAs, half the VALU
// instructions, which on this hardware take issue time away from the fp32 MFMAs one for one)
// PACKED = false (kernels with bf16 MFMAs in flight): the partial sums on scalar FMAs.  With packed FMAs feeding the
// row swaps, sixteen lanes of one output now and then came out wrong (a few tiles per 10^6, never the same ones)
// once matrix instructions of other waves were executing beside them; the fp32-MFMA kernels, in which nothing issues
// while a matrix instruction executes, never showed it.
template <bool PACKED = true>
DEV float fc3_pair(const f32x2 (&w3p)[4], float b3k, const f32x4 &hA, const f32x4 &hB) {
    float pa0, pa1, pb0, pb1;
    if (PACKED) {
        f32x2 pa = w3p[0] * f32x2{hA[0], hA[0]}, pb = w3p[0] * f32x2{hB[0], hB[0]};
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            pa = __builtin_elementwise_fma(w3p[r], f32x2{hA[r], hA[r]}, pa);
            pb = __builtin_elementwise_fma(w3p[r], f32x2{hB[r], hB[r]}, pb);
        }
        pa0 = pa[0]; pa1 = pa[1]; pb0 = pb[0]; pb1 = pb[1];
    } else {
        pa0 = w3p[0][0] * hA[0]; pa1 = w3p[0][1] * hA[0]; pb0 = w3p[0][0] * hB[0]; pb1 = w3p[0][1] * hB[0];
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            pa0 = __builtin_fmaf(w3p[r][0], hA[r], pa0);
            pa1 = __builtin_fmaf(w3p[r][1], hA[r], pa1);
            pb0 = __builtin_fmaf(w3p[r][0], hB[r], pb0);
            pb1 = __builtin_fmaf(w3p[r][1], hB[r], pb1);
        }
        asm volatile("" : "+v"(pa0), "+v"(pa1), "+v"(pb0), "+v"(pb1));      // (keeps the compiler from packing them again)
    }
    // halves: lanes 0..31 collect tile A (rows g and g + 2), lanes 32..63 tile B
    const auto h0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, pa0), __builtin_bit_cast(unsigned, pb0), false, false);
    const auto h1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, pa1), __builtin_bit_cast(unsigned, pb1), false, false);
    unsigned x0 = h0[0], y0 = h0[1], x1 = h1[0], y1 = h1[1];
    asm volatile("" : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1));   // keeps both results of a swap apart
    const float s0 = __builtin_bit_cast(float, x0) + __builtin_bit_cast(float, y0);
    const float s1 = __builtin_bit_cast(float, x1) + __builtin_bit_cast(float, y1);
    // rows: even rows collect output 0, odd rows output 1
    const auto q = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s0), __builtin_bit_cast(unsigned, s1), false, false);
    unsigned u = q[0], v = q[1];
    asm volatile("" : "+v"(u), "+v"(v));
    return (__builtin_bit_cast(float, u) + __builtin_bit_cast(float, v)) + b3k;
}

// Forward only.  DEPTH limits how far a tile runs (1: through act1, 2: through act2, >= 3: to the output);
// the QAT instances with DEPTH 1..3 are the abs-max passes of the three activation quantisers.  Two tiles
// (A, B) of the chunk run half a stage apart in one instruction stream (the VALU stage of one in the
// shadow of the other's MFMAs); the output stage handles the pair at once, one (sample, output) per lane.
//
// PA: the chain generator -> PA twin in the same kernel (DEPTH 4; with QAT the output quantiser runs
// in the kernel, absmax[3] must hold its final value).  A chunk then owns 64 - (M - 1) PA outputs: its
// 64 generator outputs are parked in LDS and every lane evaluates one output of the memory polynomial
// (the arithmetic of k_pa_fwd); chunks of a row overlap by M - 1 generator outputs.
struct PaArgs {
    const float *cre, *cim;   // [K, M]
    int K, M;
    float2 *y;                // [B, L - M + 1]
    int64_t OL;
};
// PAV: 0 no PA twin; 1 PA twin with run-time K and M (coefficients in LDS); 2 and 3 the reference's twin, K = 4 odd
// orders and M = 5 taps: loops unrolled, the 40 coefficients in scalar registers, polynomial and complex product on
// packed FMAs.  3 ("carry"): a wave walks whole rows (unit == chunks per row), chunks do not overlap: the last M - 1
// generator outputs of a chunk stay in LDS for the next chunk of the row, whose lanes then evaluate the PA outputs
// t0 - 4 .. t0 + 59.  No generator output is computed twice; the host picks it when the rows fill the waves evenly.
// (the bf16 forms with the PA twin need a few registers more than four waves per SIMD leave: three there)
#ifndef FWD_WAVES_DIGITS_PA
#define FWD_WAVES_DIGITS_PA 3
#endif
template <bool QAT, int DEPTH, int PAV>
__global__ __launch_bounds__(WAVES * 64, (((QAT && GEN_QAT_DIGITS) || (!QAT && GEN_F32_SPLIT3)) && PAV != 0) ? FWD_WAVES_DIGITS_PA : FWD_WAVES_PER_SIMD) void k_gen_fwd_mfma(const float2 *__restrict__ iq, int64_t S, int64_t L,
                                                             int64_t cpr, int64_t n_chunks,
                                                             const float *__restrict__ wq, float slope,
                                                             float *__restrict__ absmax, float2 *__restrict__ out,
                                                             PaArgs pa, unsigned unit) {
    constexpr bool PA = PAV != 0, PAF = PAV >= 2, CARRY = PAV == 3;
    static_assert(!PA || DEPTH == 4, "the fused chain is a final forward");
    constexpr bool DIG = QAT && GEN_QAT_DIGITS;       // QAT: exact digit products on the bf16 matrix cores
    constexpr bool SP3 = !QAT && GEN_F32_SPLIT3;      // float: three-way split products on the bf16 matrix cores
    constexpr bool BF = DIG || SP3;                   // bf16 MFMAs in flight: no asm-wrapped VALU, no packed FMAs before row swaps
    __shared__ float lds[WAVES][SP3 ? FWD_ROW_SP3 : SER_WORDS];
    __shared__ float pa_re[PAV == 1 ? 128 : 1], pa_im[PAV == 1 ? 128 : 1];
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the chunk walk below runs on the scalar unit
    float *ser = lds[wave];
    ser[ZERO_W + lane] = 0.f;
    ser[ONE_W + lane] = 1.f;
    if (PAV == 1) {
        for (int i = threadIdx.x; i < pa.K * pa.M; i += WAVES * 64) {
            pa_re[i] = pa.cre[i];
            pa_im[i] = pa.cim[i];
        }
        __syncthreads();
    }
    // (re, im) of coefficient [k][m]: orders 0..2 in scalar registers (uniform), the leading order in vector registers (a
    // packed FMA takes one scalar operand: the first Horner step would otherwise need a copy)
    f32x2 pac[PAF ? 3 : 1][PAF ? 5 : 1], pac3[PAF ? 5 : 1];
    if constexpr (PAF) {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pac[k][m][0] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pa.cre[k * 5 + m])));
                pac[k][m][1] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pa.cim[k * 5 + m])));
            }
            pac3[m] = f32x2{pa.cre[15 + m], pa.cim[15 + m]};
            asm volatile("" : "+v"(pac3[m]));      // stays in vector registers
        }
    }
    const int chunk_step = CARRY ? CHUNK : PAF ? CHUNK - 4 : PA ? CHUNK - (pa.M - 1) : CHUNK;   // outputs a chunk advances by
    // PA: the chunk's generator outputs (carry: behind the last four of the row's previous chunk)
    float2 *const ybuf = reinterpret_cast<float2 *>(ser + TG_W);
    constexpr int YB0 = CARRY ? 4 : 0;
    static_assert(TG_W + 2 * (CHUNK + 4) <= SER_WORDS, "room for the carried outputs");
    FwdWeights w;
    FwdDigits wd;
    FwdSplit ws;
    QS s0 = qs_one(), s1 = qs_one(), s2 = qs_one(), s3 = qs_one();
    if (QAT) {
        s0 = qs_of(absmax[0]);
        if (DEPTH >= 2) s1 = qs_of(absmax[1]);
        if (DEPTH >= 3) s2 = qs_of(absmax[2]);
        if (PA) s3 = qs_of(absmax[3]);
    }
    if constexpr (DIG) load_fwd_digits<(DEPTH >= 2)>(wd, wq, ser, lane, j, g, s0.s, s1.s);
    else if constexpr (SP3) load_fwd_split(ws, wq, j, g);
    else load_fwd_weights(w, wq, j, g);
    int boff[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        // fp32 form: k-step s takes slot 4 s + g; bf16 forms: the lane group's eight consecutive slots 8 g + s
        int slot = BF ? 8 * g + s : 4 * s + g;
        // split form: the two padding slots read the words of a real slot (their weights are zero, the words finite)
        if (SP3) boff[s] = 4 * (slot_word(slot < 30 ? slot : 29) + j);
        else boff[s] = 4 * ((DIG && slot >= 30 ? ZERO_W : slot_word(slot)) + (slot < 30 ? j : 0));
    }
    f32x2 w3p[4];     // (W3[0][4g + r], W3[1][4g + r]): FC3 runs on the VALU
#pragma unroll
    for (int r = 0; r < 4; ++r) w3p[r] = f32x2{wq[G_W3 + 4 * g + r], wq[G_W3 + 16 + 4 * g + r]};
    const float b3k = wq[G_B3 + (g & 1)];
    float amax = 0.f;
    const char *const ser_b = reinterpret_cast<const char *>(ser);
    const bool odd = (g & 1) != 0;

    struct Tile {
        f32x4 a0, a1, ha, hb, c0, c1, h2;
        u32x4 xh, xl;          // digit form: the tile's h1 as FC2 operands
        Split3 xs;             // split form: the same
    };
    auto S1 = [&](Tile &t, int c) {
        t.a0 = f32x4{0.f, 0.f, 0.f, 0.f};   // b1 enters through slot 30 (fp32 form) or with the scale (digit form)
        t.a1 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (SP3) {
            float a[8], b[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a[e] = *reinterpret_cast<const float *>(ser_b + boff[e] + 64 * c);
                b[e] = *reinterpret_cast<const float *>(ser_b + boff[e] + 64 * c + 4 * SPB_FWD);
            }
            Split3 x;
            split_operand(x, a, b);
            t.a0 = mfma_split3(ws.A1[0], x, ws.b1[0]);
            t.a1 = mfma_split3(ws.A1[1], x, ws.b1[1]);
            return;
        }
        if constexpr (DIG) {
            unsigned d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = *reinterpret_cast<const unsigned *>(ser_b + boff[e] + 64 * c);
            u32x4 xh, xl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xh[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x07060302u);
                xl[i] = __builtin_amdgcn_perm(d[2 * i + 1], d[2 * i], 0x05040100u);
            }
            t.a0 = mfma_digits(wd.A1h[0], wd.A1l[0], xh, xl, t.a0);
            t.a1 = mfma_digits(wd.A1h[1], wd.A1l[1], xh, xl, t.a1);
            return;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float b = *reinterpret_cast<const float *>(ser_b + boff[s] + 64 * c);
            t.a0 = mfma4(w.A1[0][s], b, t.a0);
            t.a1 = mfma4(w.A1[1][s], b, t.a1);
        }
    };
    // Abs-max passes end in the activation whose scale they produce.  |LeakyReLU(z)| = max(z, -slope z) for
    // slope >= 0 and z -> slope z is monotonic, so the abs-max of the activations is max(max z, |slope min z|):
    // the pass tracks the extremes of the raw sums (two three-operand instructions per pair of values) and
    // applies the activation once per wave at the end.  zmax / zmin start at 0: a batch never has fewer values.
    float zmax = 0.f, zmin = 0.f;
    auto extremes = [&](const f32x4 &a, const f32x4 &b, bool valid, bool full) {
        if (full) {            // wave-uniform: every sample of the tile lies inside the row
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                zmax = fmaxf(fmaxf(a[r], b[r]), zmax);
                zmin = fminf(fminf(a[r], b[r]), zmin);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = valid ? a[r] : 0.f, y = valid ? b[r] : 0.f;
                zmax = fmaxf(fmaxf(x, y), zmax);
                zmin = fminf(fminf(x, y), zmin);
            }
        }
    };
    auto V1 = [&](Tile &t, bool valid, bool full) {
        if constexpr (DIG) {            // the integer sums become pre-activations: scales and bias, once per value
            t.a0 = __builtin_elementwise_fma(t.a0, wd.sc1[0], wd.b1[0]);
            t.a1 = __builtin_elementwise_fma(t.a1, wd.sc1[1], wd.b1[1]);
        }
        if (QAT && DEPTH == 1 && slope >= 0.f) {
            extremes(t.a0, t.a1, valid, full);
            return;
        }
        const f32x4 x0 = lrelu4<!BF>(t.a0, slope), x1 = lrelu4<!BF>(t.a1, slope);
        if constexpr (SP3) {
            if (DEPTH >= 2) split8(t.xs, x0, x1);
            return;
        }
        if constexpr (DIG && DEPTH >= 2) {
            const f32x4 q0 = q16x4(x0, s1), q1 = q16x4(x1, s1);
            unsigned dh[4], dl[4];
            digit_pair(q0[0], q0[1], dh[0], dl[0]);
            digit_pair(q0[2], q0[3], dh[1], dl[1]);
            digit_pair(q1[0], q1[1], dh[2], dl[2]);
            digit_pair(q1[2], q1[3], dh[3], dl[3]);
            t.xh = u32x4{dh[0], dh[1], dh[2], dh[3]};
            t.xl = u32x4{dl[0], dl[1], dl[2], dl[3]};
            return;
        }
        t.ha = (QAT && DEPTH >= 2) ? fq16x4(x0, s1) : x0;
        t.hb = (QAT && DEPTH >= 2) ? fq16x4(x1, s1) : x1;
        if (QAT && DEPTH == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) amax = fmaxf(amax, valid ? fmaxf(fabsf(x0[r]), fabsf(x1[r])) : 0.f);
        }
    };
    auto S2 = [&](Tile &t) {
        if constexpr (DIG) {
            t.c0 = mfma_digits(wd.A2h, wd.A2l, t.xh, t.xl, f32x4{0.f, 0.f, 0.f, 0.f});
            return;
        }
        if constexpr (SP3) {
            t.c0 = mfma_split3(ws.A2, t.xs, ws.b2);
            return;
        }
        t.c0 = w.bias2;
        t.c1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t.c0 = mfma4(w.A2[0][r], t.ha[r], t.c0);
            t.c1 = mfma4(w.A2[1][r], t.hb[r], t.c1);
        }
    };
    auto V2 = [&](Tile &t, bool valid, bool full) {
        f32x4 z2;
        if constexpr (DIG) z2 = __builtin_elementwise_fma(t.c0, wd.sc2, wd.b2);
        else if constexpr (SP3) z2 = t.c0;
        else z2 = add4(t.c0, t.c1);
        if (QAT && DEPTH == 2 && slope >= 0.f) {
            extremes(z2, z2, valid, full);
            return;
        }
        const f32x4 x = lrelu4<!BF>(z2, slope);
        t.h2 = (QAT && DEPTH >= 3) ? fq16x4<!BF>(x, s2) : x;
        if (QAT && DEPTH == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) amax = fmaxf(amax, valid ? fabsf(x[r]) : 0.f);
        }
    };
#define PHASE() __builtin_amdgcn_sched_barrier(0)
    // bf16 forms: a matrix instruction occupies its pipe for 16 cycles but the issue port only for a few, so the VALU
    // stage that shares a phase with an MFMA stage is spread between the MFMAs (NV vector instructions after each of NM
    // matrix instructions) instead of following them as a block; the fp32 MFMA blocks the port, nothing to gain there
#ifndef GEN_MIX
#define GEN_MIX 1
#endif
#define MIX(NM, NV)                                                  \
    if constexpr (BF && GEN_MIX) {                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < (NM); ++i_) {        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       \
            __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);    \
        }                                                            \
    }

    // A wave walks through units of `unit` consecutive chunks (units dealt round robin over the waves; all of
    // this is scalar).  A chunk needs 69 time points; consecutive chunks of a row share the last
    // H = 69 - chunk_step of them (the 5 taps of history, plus the M - 1 generator outputs the PA chain
    // recomputes), and their five series values are moved inside LDS instead of being loaded and derived
    // again: each lane then loads and derives exactly one new point per chunk, one chunk ahead.  Only the
    // first chunk of a run (start of a unit or of a row) loads its H leading points as well.
    const unsigned n_waves = gridDim.x * WAVES, cpr32 = (unsigned)cpr, n_chunks32 = (unsigned)n_chunks;
    const int H = 69 - chunk_step;
    unsigned chunk = (blockIdx.x * WAVES + wave) * unit, k_in_unit = 0;
    unsigned row = chunk / cpr32, pos = chunk - row * cpr32;
    bool fresh = true;                     // the current chunk starts a run
    float2 pre = make_float2(0.f, 0.f), pre_h = make_float2(0.f, 0.f);
    auto fetch = [&](unsigned r, unsigned ps, bool fr) {
        const float2 *iqrow = iq + (int64_t)r * S;
        const int64_t tb = (int64_t)ps * chunk_step;
        pre = make_float2(0.f, 0.f);
        if (lane < chunk_step && tb + H + lane < S) pre = iqrow[tb + H + lane];
        if (fr) {
            pre_h = make_float2(0.f, 0.f);
            if (lane < H && tb + lane < S) pre_h = iqrow[tb + lane];
        }
    };
    if (chunk < n_chunks32) fetch(row, pos, true);
    while (chunk < n_chunks32) {
        if (fresh) {
            if (lane < H) commit_point<QAT, DIG, SP3>(ser, pre_h, lane, s0);
        } else {
            for (int c = lane; c < 5 * H; c += 64) {
                const int type = c / H, w0 = SER_BASE0 + SER_STRIDE * type + (c - type * H);
                ser[w0] = ser[w0 + chunk_step];
                if (SP3) ser[w0 + SPB_FWD] = ser[w0 + SPB_FWD + chunk_step];
            }
        }
        if (lane < chunk_step) commit_point<QAT, DIG, SP3>(ser, pre, H + lane, s0);
        const int64_t t0_cur = (int64_t)pos * chunk_step, row_cur = row;
        // next chunk of this wave
        if (++k_in_unit < unit) {
            chunk += 1;
            fresh = ++pos == cpr32;
            if (fresh) {
                pos = 0;
                ++row;
            }
        } else {
            k_in_unit = 0;
            chunk += 1 + (n_waves - 1) * unit;
            row = chunk / cpr32;
            pos = chunk - row * cpr32;
            fresh = true;
        }
        if (chunk < n_chunks32) fetch(row, pos, fresh);
        __builtin_amdgcn_wave_barrier();
        // scalar per chunk: outputs left in the row, base of the chunk's outputs (lanes add a 32-bit index)
        const int rem = (int)(L - t0_cur < CHUNK ? L - t0_cur : CHUNK);
        float2 *const orow = out ? out + (row_cur * L + t0_cur) : nullptr;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (t0_cur + 32 * p >= L) break;      // wave-uniform: the wave stays convergent
            const int cA = 2 * p, cB = 2 * p + 1;
            const bool vA = (t0_cur + 16 * cA + j) < L, vB = (t0_cur + 16 * cB + j) < L;
            const bool fA = t0_cur + 16 * cA + 16 <= L, fB = t0_cur + 16 * cB + 16 <= L;
            Tile A, B;
            S1(A, cA);
            PHASE();
            S1(B, cB);
            V1(A, vA, fA);
            if constexpr (BF && GEN_MIX) {        // the operand reads and their byte permutes first
                __builtin_amdgcn_sched_group_barrier(0x100, SP3 ? 16 : 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, SP3 ? 12 : 8, 0);
            }
            MIX(SP3 ? 12 : 8, SP3 ? 4 : 6)
            PHASE();
            if (DEPTH >= 2) {
                S2(A);
                V1(B, vB, fB);
                MIX(SP3 ? 6 : 4, SP3 ? 8 : 12)
                PHASE();
                S2(B);
                V2(A, vA, fA);
                MIX(SP3 ? 6 : 4, 3)
                PHASE();
                if (DEPTH >= 3) {
                    V2(B, vB, fB);
                    // lane (g, j): output g & 1 of sample j of tile g >> 1
                    float y = tanh_f32(fc3_pair<!BF>(w3p, b3k, A.h2, B.h2));
                    const int tl = 32 * p + 16 * (g >> 1) + j;          // output index inside the chunk
                    const bool in_row = tl < rem;                        // rem: outputs left in the row from t0_cur (scalar)
                    if (QAT && !PA) amax = fmaxf(amax, in_row ? fabsf(y) : 0.f);
                    if (QAT && PA) y = fq16(y, s3);
                    // one row swap pairs the two outputs of a sample: rows 0 and 2 then store float2
                    const unsigned bits = __builtin_bit_cast(unsigned, y);
                    const auto r = __builtin_amdgcn_permlane16_swap(bits, bits, false, false);
                    unsigned u0 = r[0], u1 = r[1];
                    asm volatile("" : "+v"(u0), "+v"(u1));   // see the note on folded swap results in the backward kernel
                    const float2 yy = make_float2(__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1));
                    if (PA) {
                        if (!odd) ybuf[YB0 + tl] = yy;
                        // generator outputs in the overlap belong to the next chunk of the row
                        const bool mine = CARRY || tl < chunk_step || t0_cur + chunk_step >= pa.OL;
                        if (out && !odd && in_row && mine) orow[tl] = yy;
                    } else {
                        if (out && !odd && in_row) orow[tl] = yy;
                    }
                } else {
                    V2(B, vB, fB);
                }
            } else {
                V1(B, vB, fB);
            }
        }
        if constexpr (PAF) {
            __builtin_amdgcn_wave_barrier();
            // PA output of this lane: n = t0_cur - YB0 + lane; its bounds and the base of the stores are scalars
            const int lane_lo = (CARRY && t0_cur == 0) ? YB0 : 0;
            const int64_t left = pa.OL - (t0_cur - YB0);
            const int lane_hi = (int)(left < chunk_step ? left : chunk_step);
            if (lane >= lane_lo && lane < lane_hi) {
                const f32x2 *const yb = reinterpret_cast<const f32x2 *>(ybuf);
                f32x2 *const yrow = reinterpret_cast<f32x2 *>(pa.y + (row_cur * pa.OL + t0_cur - YB0));
                f32x2 acc = {0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    const f32x2 v = yb[lane + 4 - m];
                    const float pw = __builtin_fmaf(v[1], v[1], v[0] * v[0]);
                    const f32x2 pw2 = {pw, pw};
                    f32x2 P = __builtin_elementwise_fma(pac3[m], pw2, pac[2][m]);   // (Re, Im) of the tap polynomial, Horner in |u|^2
                    P = __builtin_elementwise_fma(P, pw2, pac[1][m]);
                    P = __builtin_elementwise_fma(P, pw2, pac[0][m]);
                    // acc += v * P (complex): (v.x P.re, v.x P.im), then (-v.y P.im, v.y P.re) with the halves of v and P picked
                    // and the sign applied by the instruction's own modifiers
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(v), "v"(P));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(v), "v"(P));
                }
                yrow[lane] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (CARRY) {
                if (lane < 4) ybuf[lane] = ybuf[CHUNK + lane];
            }
        } else if (PA) {
            __builtin_amdgcn_wave_barrier();
            const int64_t n = t0_cur + lane;                     // PA output of this lane
            if (lane < chunk_step && n < pa.OL) {
                float yr = 0.f, yi = 0.f;
                for (int m = 0; m < pa.M; ++m) {
                    const float2 v = ybuf[lane + pa.M - 1 - m];
                    const float pw = v.x * v.x + v.y * v.y;
                    float pr = pa_re[(pa.K - 1) * pa.M + m], pi = pa_im[(pa.K - 1) * pa.M + m];
                    for (int k = pa.K - 2; k >= 0; --k) {
                        pr = __builtin_fmaf(pr, pw, pa_re[k * pa.M + m]);
                        pi = __builtin_fmaf(pi, pw, pa_im[k * pa.M + m]);
                    }
                    yr += v.x * pr - v.y * pi;
                    yi += v.x * pi + v.y * pr;
                }
                pa.y[row_cur * pa.OL + n] = make_float2(yr, yi);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#undef PHASE
    if (QAT && !PA) {
        if (DEPTH <= 2 && slope >= 0.f) {
            // the activation of the two extremes, with the arithmetic of lrelu4 (one rounding of slope * z)
            // LeakyReLU(zmax) = zmax >= 0, LeakyReLU(zmin) = slope * zmin <= 0 (one rounding, as in lrelu4)
            amax = fmaxf(zmax, fabsf(slope * zmin));
        }
        // one atomic per workgroup: atomics on one address take their turn in the L2 (about 12 ns each), and the
        // waves of a launch all arrive here at about the same time
        amax = wave_max(amax);
        __shared__ float wg_max[WAVES];
        if (lane == 0) wg_max[wave] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = wg_max[0];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) m = fmaxf(m, wg_max[w]);
            atomic_max_nonneg(absmax + DEPTH, m);
        }
    }
}

// max over all time points of |I|, |Q|, |x|, |x|^2, |x|^4 == abs-max of the assembled feature tensor.
// The correctly rounded square root, e -> e * e and e2 -> e2 * e2 are monotonic, so the maxima of |x|,
// |x|^2 and |x|^4 are those functions of max(I*I + Q*Q): the stream pays two multiplies, one addition and
// three maxima per point, the square root is taken once per wave.  float4 loads (two points), four in
// flight per lane; one atomic per workgroup.  `iq4` is the 16-byte aligned body of the stream, `edge`
// the at most two points (one before, one behind) that do not fit it.
__global__ __launch_bounds__(256) void k_iq_absmax(const float4 *__restrict__ iq4, int64_t n4,
                                                   const float2 *__restrict__ edge0, const float2 *__restrict__ edge1,
                                                   float *__restrict__ absmax) {
    __shared__ float part[2][4];
    float mc = 0.f, mp = 0.f;   // max |component|, max I*I + Q*Q
    auto point = [&](float a, float b) {
        mc = fmaxf(mc, fmaxf(fabsf(a), fabsf(b)));
        mp = fmaxf(mp, a * a + b * b);
    };
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 v0 = iq4[i], v1 = iq4[i + stride], v2 = iq4[i + 2 * stride], v3 = iq4[i + 3 * stride];
        point(v0.x, v0.y); point(v0.z, v0.w);
        point(v1.x, v1.y); point(v1.z, v1.w);
        point(v2.x, v2.y); point(v2.z, v2.w);
        point(v3.x, v3.y); point(v3.z, v3.w);
    }
    for (; i < n4; i += stride) {
        const float4 v = iq4[i];
        point(v.x, v.y); point(v.z, v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (edge0) point(edge0->x, edge0->y);
        if (edge1) point(edge1->x, edge1->y);
    }
    mc = wave_max(mc);
    mp = wave_max(mp);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[0][wave] = mc;
        part[1][wave] = mp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mc = fmaxf(fmaxf(part[0][0], part[0][1]), fmaxf(part[0][2], part[0][3]));
        mp = fmaxf(fmaxf(part[1][0], part[1][1]), fmaxf(part[1][2], part[1][3]));
        const float e = sqrtf(mp), e2 = e * e;
        atomic_max_nonneg(absmax, fmaxf(mc, fmaxf(e, fmaxf(e2, e2 * e2))));
    }
}

// ------------------------------------------------------------------------------------ backward
// PA: the chain generator -> (output quantiser) -> memory-polynomial PA twin with its backward in the same
// kernel (K <= 4 odd orders, M = 5 taps: the reference's PA).  The loss or upstream gradient then lives on
// the PA output [B, OL = L - 4, 2].  The unit of the chain is the tile PAIR, whose two tiles are in registers
// together in the output stage anyway: its 32 generator outputs give 28 PA outputs, whose gradient goes back
// to those 32 generator outputs; pairs (and chunks) overlap by 4 generator outputs.  A generator output
// in an overlap is run through the backward in both pairs, each with the part of its gradient that comes
// from that pair's own PA outputs: everything downstream of that gradient is linear in it, so the two
// parts add up to the whole.  A chunk therefore advances by 56 PA outputs (pairs at 0 and 28).
struct PaBwdArgs {
    const float *cre, *cim;   // [K, 5]
    int K;
    int64_t OL;               // PA outputs per row
};
constexpr int PA_M = 5, PA_PAIR = 32 - (PA_M - 1), PA_CHUNK = 2 * PA_PAIR;
template <bool QAT, bool MSE, bool PA>
__global__ __launch_bounds__(WAVES * 64, BWD_WAVES_PER_SIMD) void k_gen_bwd_mfma(const float2 *__restrict__ iq, int64_t S, int64_t L,
                                                             int64_t cpr, int64_t n_chunks,
                                                             const float *__restrict__ wq, float slope,
                                                             const float *__restrict__ absmax,
                                                             const float2 *__restrict__ grad_out,
                                                             const float2 *__restrict__ target, float mse_scale,
                                                             float *__restrict__ slab, unsigned *__restrict__ queue,
                                                             PaBwdArgs pa) {
    // DIG: FC1 and FC2 of the forward half as digit products on the bf16 matrix cores (see the note at mfma_digits); the
    // backward half keeps the fp32 MFMA.  No asm-wrapped VALU instructions and no packed FMAs in front of row swaps then.
    constexpr bool DIG = QAT && GEN_QAT_DIGITS;
    // SP3: FC1 and FC2 of the float generator's forward half as three-way split products on the bf16 matrix cores
    constexpr bool SP3 = !QAT && GEN_F32_SPLIT3;
    constexpr bool BF = DIG || SP3;
    // OB: the weight gradients dW1 and dW2 on the bf16 matrix cores as well (see the note at outer3).  The series then hold
    // the forward's words only (digit words, or words A with words B SPB_FWD behind them, as in the forward kernel: the x
    // operand of dW1 is built from them), and a tile buffer is five term slots: h1 (three terms; two digits in the digit
    // form), overwritten by gz1 once dW2 has read them, and gz2 ([h|m] in one slot, l in half of the next).
#ifndef GEN_BWD_BF16
#define GEN_BWD_BF16 1
#endif
    constexpr bool OB = BF && GEN_BWD_BF16;
    constexpr int EXTRA_WORDS = OB ? (SP3 ? FWD_ROW_SP3 - SER_WORDS : 0) : DIG ? DSER_WORDS : SP3 ? SPB_BWD + DSER_WORDS : 0;
    constexpr int TB_WORDS = OB ? 5 * SLOT_B / 4 : TILE_W;         // words per tile buffer
    constexpr int WAVE_WORDS = SER_WORDS + EXTRA_WORDS + 2 * TB_WORDS;
    constexpr int DSER_W = SER_WORDS;                              // !OB: digit series (DIG) / words A then words B (SP3)
    constexpr int GZ_W = SER_WORDS + EXTRA_WORDS;                  // + TB_WORDS * buffer (these two shadow the namespace-scope ones)
    constexpr int H1_W = GZ_W + 16 * P32;
    constexpr int S_H1 = 0, S_G1 = 0, S_G2 = 3 * SLOT_B;           // OB: byte offsets of the slot groups in a tile buffer
    static_assert(WAVES * WAVE_WORDS >= WAVES * G_SLAB, "the gradient rows are combined in the staging LDS");
    __shared__ __attribute__((aligned(16))) float lds[WAVES * WAVE_WORDS];   // static: addresses fold into the instructions
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: chunk, row and position arithmetic run on the scalar unit
    float *ser = lds + wave * WAVE_WORDS;
    ser[ZERO_W + lane] = 0.f;
    ser[ONE_W + lane] = 1.f;
    if (PA && lane < 48) {   // coefficient block of this wave: [tap][re k0..3 | im k0..3], orders >= K and tap 5 are zero
        const int m = lane >> 3, k = lane & 3;
        const float *src = (lane & 4) ? pa.cim : pa.cre;
        ser[PAC_W + lane] = (m < PA_M && k < pa.K) ? src[k * PA_M + m] : 0.f;
    }
    QS s0 = qs_one(), s1 = qs_one(), s2 = qs_one(), s3 = qs_one();
    if (QAT) {
        s0 = qs_of(absmax[0]);
        s1 = qs_of(absmax[1]);
        s2 = qs_of(absmax[2]);
        s3 = qs_of(absmax[3]);
    }
    FwdWeights w;
    FwdDigits wd;
    FwdSplit ws;
    if constexpr (DIG) load_fwd_digits<true>(wd, wq, ser + GZ_W, lane, j, g, s0.s, s1.s);
    else if constexpr (SP3) load_fwd_split(ws, wq, j, g);
    else load_fwd_weights(w, wq, j, g);
    float A2T[2][4];  // W2[4g + r][16t + i]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) A2T[t][r] = wq[G_W2 + (4 * g + r) * 32 + 16 * t + j];
    // OB: the same weights as split operands.  The contraction of gh1 = W2^T gz2 runs over the 16 layer-2 neurons only, so
    // a K = 32 instruction takes two terms of the lane group's four neurons: T2[2t] = [h|l], T2[2t + 1] = [h|m] of row tile t
    u32x4 T2[4];
    if constexpr (OB) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned h[2], m[2], l[2];
            split_pair3(A2T[t][0], A2T[t][1], h[0], m[0], l[0]);
            split_pair3(A2T[t][2], A2T[t][3], h[1], m[1], l[1]);
            T2[2 * t] = u32x4{h[0], h[1], l[0], l[1]};
            T2[2 * t + 1] = u32x4{h[0], h[1], m[0], m[1]};
        }
    }
    f32x2 w3p[4];     // (W3[0][4g + r], W3[1][4g + r]): FC3 and its input gradient run on the VALU
    f32x4 w3r[2];     // the same weights by row, for the input gradient
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w3p[r] = f32x2{wq[G_W3 + 4 * g + r], wq[G_W3 + 16 + 4 * g + r]};
        w3r[0][r] = w3p[r][0];
        w3r[1][r] = w3p[r][1];
    }
    const float b3k = wq[G_B3 + (g & 1)];
    int boff[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        // fp32 form: k-step s takes slot 4 s + g; digit form: the lane group's eight consecutive slots 8 g + s, digit series
        int slot = BF ? 8 * g + s : 4 * s + g;
        if (OB && DIG) boff[s] = 4 * ((slot >= 30 ? ZERO_W : slot_word(slot)) + (slot < 30 ? j : 0));
        else if (OB && SP3) boff[s] = 4 * (slot_word(slot < 30 ? slot : 29) + j);
        else if (DIG) boff[s] = 4 * (slot >= 30 ? ZERO_W : DSER_W + slot_word(slot) - SER_BASE0 + j);
        else if (SP3) boff[s] = 4 * (DSER_W + slot_word(slot < 30 ? slot : 29) - SER_BASE0 + j);   // padding slots: finite words, zero weights
        else boff[s] = 4 * (slot_word(slot) + (slot < 30 ? j : 0));
    }
    // OB: x operand of dW1 (left factor, rows = slots): lane (kg = g, n = j) reads the words of samples 4g..4g+3 of slot
    // 16u + j (slot 30: the constant 1 whose row is the bias gradient, slot 31: zeros; their words B are zeros)
    int xa[2], xb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int slot = 16 * u + j;
        xa[u] = 4 * (slot_word(slot) + 4 * g);
        xb[u] = slot < 30 ? xa[u] + 4 * SPB_FWD : 4 * (ZERO_W + 4 * g);
    }
    // OB: this lane's 16 bytes of row j of a slot (store) and the 8 bytes it points the transposing read at
    const int tw_off = 64 * j + 16 * (g ^ ((j >> 1) & 3));
    const int tr_row = 4 * g + (j >> 2);
    const int tr_off = 64 * tr_row + 16 * ((j & 3) ^ ((tr_row >> 1) & 3));
    TrAddr tra;       // tile buffer 0; buffer 1 is an immediate offset
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        tra.p[k] = (lds_char *)(reinterpret_cast<char *>(ser) + 4 * GZ_W + tr_off);
        asm volatile("" : "+v"(tra.p[k]));      // (volatile: identical plain asm statements on one input are one value to the optimiser)
    }
    // dW1 A operand: lane (i = j, g) reads slot 16u + i of sample 4q + g
    int aoff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int slot = 16 * u + j;
        aoff[u] = 4 * (slot_word(slot) + (slot < 30 ? g : 0));
    }
    // transposition tiles: write address (row j) and read addresses (row 4q + g)
    const int wr32 = 4 * (j * P32 + ((4 * g) ^ swz(j)));
    int rd32[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int s = 4 * q + g;
        rd32[q] = 4 * (s * P32 + (j ^ swz(s)));
    }
    char *const ser_b = reinterpret_cast<char *>(ser);
    f32x4 dW1[2][2], dW2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        dW2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) dW1[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 dW3[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 db2 = {0.f, 0.f, 0.f, 0.f};
    float db3 = 0.f;
    float loss = 0.f;
    const bool odd = (g & 1) != 0;
    float2 *const tg_lds = reinterpret_cast<float2 *>(ser + TG_W);
    const float *const tg_f = ser + TG_W;

    // One tile's life is an alternation of MFMA stages (S1 FC1, S2 FC2, S5 gh1, S6 weight-gradient outer
    // products) and VALU/LDS stages (V1, V2, the output stage of the pair with FC3 and its gradients, V4,
    // V5) that each need the previous result.  Two tiles (A, B) of the chunk run half a stage apart in
    // one instruction stream, so the VALU stage of one tile sits in the shadow of the other's MFMAs:
    //   S1A | S1B+V1A | S2A+V1B | S2B+V2A | V2B+V3(A,B)+V4A | S5A+V4B | S5B+V5A | S6A+V5B | S6B
    struct Tile {
        f32x4 a0, a1, m1a, m1b, ha, hb, c0, c1, m2, h2, gz2, e0, e1;
        u32x4 xh, xl;          // digit form: the tile's h1 as FC2 operands (ha, hb, c1 unused then)
        Split3 xs;             // split form: the same
        unsigned g2h[2], g2m[2], g2l[2];   // OB: the terms of gz2
    };
    // FC1 operands are fetched from LDS one stage before the MFMAs that use them.  A tile is named by the
    // byte offset of its first sample inside the chunk's series (64 * tile, or 4 * (28 * pair + 16 * half) in
    // the chain, whose pairs overlap)
    struct Fc1Operand {
        float v[8], b[8];      // b: words B of the split form
    };
    auto F1 = [&](Fc1Operand &b, int toff) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            b.v[s] = *reinterpret_cast<const float *>(ser_b + boff[s] + toff);
            if (SP3) b.b[s] = *reinterpret_cast<const float *>(ser_b + boff[s] + toff + 4 * (OB ? SPB_FWD : SPB_BWD));
        }
    };
    auto S1 = [&](f32x4 &a0, f32x4 &a1, const Fc1Operand &b) {
        a0 = f32x4{0.f, 0.f, 0.f, 0.f};   // b1 enters through slot 30 (fp32 form) or with the scales (digit form)
        a1 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (SP3) {
            Split3 x;
            split_operand(x, b.v, b.b);
            a0 = mfma_split3(ws.A1[0], x, ws.b1[0]);
            a1 = mfma_split3(ws.A1[1], x, ws.b1[1]);
            return;
        }
        if constexpr (DIG) {
            u32x4 xh, xl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned d0 = __builtin_bit_cast(unsigned, b.v[2 * i]), d1 = __builtin_bit_cast(unsigned, b.v[2 * i + 1]);
                xh[i] = __builtin_amdgcn_perm(d1, d0, 0x07060302u);
                xl[i] = __builtin_amdgcn_perm(d1, d0, 0x05040100u);
            }
            a0 = mfma_digits(wd.A1h[0], wd.A1l[0], xh, xl, a0);
            a1 = mfma_digits(wd.A1h[1], wd.A1l[1], xh, xl, a1);
            return;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            a0 = mfma4(w.A1[0][s], b.v[s], a0);
            a1 = mfma4(w.A1[1][s], b.v[s], a1);
        }
    };
    auto V1 = [&](Tile &t, int buf) {
        if constexpr (DIG) {            // the integer sums become pre-activations: scales and bias, once per value
            t.a0 = __builtin_elementwise_fma(t.a0, wd.sc1[0], wd.b1[0]);
            t.a1 = __builtin_elementwise_fma(t.a1, wd.sc1[1], wd.b1[1]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t.m1a[r] = t.a0[r] > 0.f ? 1.f : slope;
            t.m1b[r] = t.a1[r] > 0.f ? 1.f : slope;
        }
        t.a0 = mul4<!BF>(t.a0, t.m1a);
        t.a1 = mul4<!BF>(t.a1, t.m1b);
        if constexpr (SP3) {
            t.ha = t.a0;
            t.hb = t.a1;
            split8(t.xs, t.ha, t.hb);
        } else if constexpr (DIG) {
            // h1 as integers: digits for FC2, the integers themselves (as floats) for dW2, which is scaled once at the end
            t.ha = q16x4(t.a0, s1);
            t.hb = q16x4(t.a1, s1);
            unsigned dh[4], dl[4];
            digit_pair(t.ha[0], t.ha[1], dh[0], dl[0]);
            digit_pair(t.ha[2], t.ha[3], dh[1], dl[1]);
            digit_pair(t.hb[0], t.hb[1], dh[2], dl[2]);
            digit_pair(t.hb[2], t.hb[3], dh[3], dl[3]);
            t.xh = u32x4{dh[0], dh[1], dh[2], dh[3]};
            t.xl = u32x4{dl[0], dl[1], dl[2], dl[3]};
        } else {
            t.ha = QAT ? fq16x4(t.a0, s1) : t.a0;
            t.hb = QAT ? fq16x4(t.a1, s1) : t.a1;
        }
        if constexpr (OB) {             // the terms of h1, the left factor of dW2
            char *tw = ser_b + 4 * (GZ_W + TB_WORDS * buf) + tw_off;
            if constexpr (SP3) {
                *reinterpret_cast<u32x4 *>(tw + S_H1) = t.xs.h;
                *reinterpret_cast<u32x4 *>(tw + S_H1 + SLOT_B) = t.xs.m;
                *reinterpret_cast<u32x4 *>(tw + S_H1 + 2 * SLOT_B) = t.xs.l;
            } else {
                *reinterpret_cast<u32x4 *>(tw + S_H1) = t.xh;
                *reinterpret_cast<u32x4 *>(tw + S_H1 + SLOT_B) = t.xl;
            }
            __builtin_amdgcn_wave_barrier();
            return;
        }
        char *h1_b = ser_b + 4 * (H1_W + TILE_W * buf);
        *reinterpret_cast<f32x4 *>(h1_b + wr32) = t.ha;
        *reinterpret_cast<f32x4 *>(h1_b + wr32 + 64) = t.hb;
    };
    auto S2 = [&](Tile &t) {
        if constexpr (DIG) {
            t.c0 = mfma_digits(wd.A2h, wd.A2l, t.xh, t.xl, f32x4{0.f, 0.f, 0.f, 0.f});
            return;
        }
        if constexpr (SP3) {
            t.c0 = mfma_split3(ws.A2, t.xs, ws.b2);
            return;
        }
        t.c0 = w.bias2;
        t.c1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t.c0 = mfma4(w.A2[0][r], t.ha[r], t.c0);
            t.c1 = mfma4(w.A2[1][r], t.hb[r], t.c1);
        }
    };
    auto V2 = [&](Tile &t) {
        if constexpr (DIG) t.c0 = __builtin_elementwise_fma(t.c0, wd.sc2, wd.b2);
        else if constexpr (!SP3) t.c0 = add4(t.c0, t.c1);
#pragma unroll
        for (int r = 0; r < 4; ++r) t.m2[r] = t.c0[r] > 0.f ? 1.f : slope;
        t.c0 = mul4<!BF>(t.c0, t.m2);
        t.h2 = QAT ? fq16x4<!BF>(t.c0, s2) : t.c0;
    };
    // Output stage of a tile pair: 32 samples x 2 outputs = one (sample, output) per lane.  Lane (g, j)
    // takes output g & 1 of sample j of tile g >> 1, so tanh, loss and gz3 are evaluated once each; gz3
    // then goes to all lane groups with three row swaps, where dW3 and gh2 = W3^T gz3 (two FMAs per
    // neuron) follow on the VALU.
    const int tg_off = 2 * (16 * (g >> 1) + j) + (g & 1);
    // Chain: the pair's 32 generator outputs -> 28 PA outputs -> loss / upstream gradient -> gradient w.r.t. the
    // 32 generator outputs, all inside the output stage.  With u = the (quantised) generator output of a sample,
    // p = |u|^2 and P_m the tap polynomials, y[n] = sum_m u[n+4-m] P_m(p[n+4-m]) and, for G = dL/dy (real and
    // imaginary part as a complex number), dL/du[t] = sum_m G[t-4+m] conj(P_m(p_t)) + 2 u_t sum_m Re(conj(G[t-4+m]) u_t P_m'(p_t)).
    // Lane (g, j) owns component c = g & 1 of sample tl = 16 (g >> 1) + j of the pair; the two lanes of a sample
    // split the taps (c = 0: taps 0..2, c = 1: taps 3, 4 and an all-zero tap 5) and exchange three partial sums.
    // Scratch: the gz tile buffer 0, which is dead between the pairs (LDS operations of a wave run in order).
    auto pa_stage = [&](float y, int p, int64_t t0_cur) -> float {
        const int c = g & 1, tl = 16 * (g >> 1) + j;
        // (OB: the h1 terms of tile A are parked at the start of buffer 0 by now; its gz2 slots are still free)
        constexpr int PA_SCR = GZ_W + (OB ? S_G2 / 4 : 0);
        static_assert(PA_SCR + 384 + 74 <= GZ_W + TB_WORDS, "chain scratch inside tile buffer 0");
        float *const fbuf = ser + PA_SCR;        // [6 taps][32 samples] float2: u P_m(p)
        float *const gbuf = ser + PA_SCR + 384;  // [4 zeros | 28 G | 5 zeros] float2
        const float o = QAT ? fq16(y, s3) : y;   // straight-through: the gradient passes the quantiser unchanged
        const unsigned ob = __builtin_bit_cast(unsigned, o);
        const auto sw = __builtin_amdgcn_permlane16_swap(ob, ob, false, false);   // (component 0, component 1) in every lane
        unsigned u0 = sw[0], u1 = sw[1];
        asm volatile("" : "+v"(u0), "+v"(u1));
        const float ur = __builtin_bit_cast(float, u0), ui = __builtin_bit_cast(float, u1);
        const float pw = ur * ur + ui * ui;
        float Pr[3], Pi[3], Dr[3], Di[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int m = 3 * c + i;
            const f32x4 cr = *reinterpret_cast<const f32x4 *>(ser + PAC_W + 8 * m);
            const f32x4 ci = *reinterpret_cast<const f32x4 *>(ser + PAC_W + 8 * m + 4);
            float pr = cr[3], pi = ci[3], dr = 0.f, di = 0.f;   // Horner with derivative in p
#pragma unroll
            for (int k = 2; k >= 0; --k) {
                dr = __builtin_fmaf(dr, pw, pr);
                di = __builtin_fmaf(di, pw, pi);
                pr = __builtin_fmaf(pr, pw, cr[k]);
                pi = __builtin_fmaf(pi, pw, ci[k]);
            }
            Pr[i] = pr; Pi[i] = pi; Dr[i] = dr; Di[i] = di;
            const float2 f = make_float2(__builtin_fmaf(ur, pr, -(ui * pi)), __builtin_fmaf(ur, pi, ui * pr));
            *reinterpret_cast<float2 *>(fbuf + 2 * (32 * m + tl)) = f;
        }
        __builtin_amdgcn_wave_barrier();
        // PA output n = tl of the pair, component c (n >= 28 has no full window here: computed on clamped taps, masked)
        float yc = 0.f;
#pragma unroll
        for (int m = 0; m < PA_M; ++m) {
            const int idx = min(tl + PA_M - 1 - m, 31);
            yc += fbuf[2 * (32 * m + idx) + c];
        }
        const bool nvalid = tl < PA_PAIR && (t0_cur + PA_PAIR * p + tl) < pa.OL;
        const float v = tg_f[2 * (PA_PAIR * p + tl) + c];
        float G;
        if (MSE) {
            const float d = nvalid ? yc - v : 0.f;
            loss = __builtin_fmaf(d, d, loss);
            G = d * mse_scale;
        } else {
            G = nvalid ? v : 0.f;
        }
        gbuf[2 * (4 + tl) + c] = G;
        if (tl < 5) gbuf[2 * (tl < 4 ? tl : 36) + c] = 0.f;
        __builtin_amdgcn_wave_barrier();
        float ar = 0.f, ai = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float2 Gm = *reinterpret_cast<const float2 *>(gbuf + 2 * (tl + 3 * c + i));   // G[tl - 4 + m]
            ar = __builtin_fmaf(Gm.x, Pr[i], __builtin_fmaf(Gm.y, Pi[i], ar));
            ai = __builtin_fmaf(Gm.y, Pr[i], __builtin_fmaf(-Gm.x, Pi[i], ai));
            const float qr = __builtin_fmaf(ur, Dr[i], -(ui * Di[i])), qi = __builtin_fmaf(ur, Di[i], ui * Dr[i]);
            b = __builtin_fmaf(Gm.x, qr, __builtin_fmaf(Gm.y, qi, b));
        }
        auto both = [&](float x) {   // sum over the two lanes of the sample
            const unsigned xb = __builtin_bit_cast(unsigned, x);
            const auto r = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);
            unsigned x0 = r[0], x1 = r[1];
            asm volatile("" : "+v"(x0), "+v"(x1));
            return __builtin_bit_cast(float, x0) + __builtin_bit_cast(float, x1);
        };
        ar = both(ar);
        ai = both(ai);
        b = both(b);
        __builtin_amdgcn_wave_barrier();
        return __builtin_fmaf(2.f * (c ? ui : ur), b, c ? ai : ar);
    };
    auto V3P = [&](Tile &A, Tile &B, int p, int64_t t0_cur) {
        const float y = tanh_f32(fc3_pair<!BF>(w3p, b3k, A.h2, B.h2));
        float gup;
        if constexpr (PA) {
            gup = pa_stage(y, p, t0_cur);
        } else {
            const bool valid = (t0_cur + 32 * p + 16 * (g >> 1) + j) < L;
            const float v = tg_f[tg_off + 64 * p];
            if (MSE) {
                const float o = QAT ? fq16(y, s3) : y;
                const float d = valid ? o - v : 0.f;
                loss = __builtin_fmaf(d, d, loss);
                gup = d * mse_scale;
            } else {
                gup = v;                            // zero beyond L (never loaded)
            }
        }
        const float gz3 = gup * (1.f - y * y);
        db3 += gz3;                             // lanes with g & 1 == k collect db3[k]
        // each swap exchanges rows between its two operands.  The four results pass through an empty asm:
        // without it this compiler folds the second result of a swap into the first once they are
        // reinterpreted as floats (observed on ROCm 7.2: b0 == a0, b1 == a1)
        const unsigned bits = __builtin_bit_cast(unsigned, gz3);
        const auto r = __builtin_amdgcn_permlane16_swap(bits, bits, false, false);      // rows (0,0,2,2), (1,1,3,3)
        const auto q0 = __builtin_amdgcn_permlane32_swap(r[0], r[0], false, false);     // row 0 | row 2 everywhere
        const auto q1 = __builtin_amdgcn_permlane32_swap(r[1], r[1], false, false);     // row 1 | row 3 everywhere
        unsigned ua0 = q0[0], ub0 = q0[1], ua1 = q1[0], ub1 = q1[1];
        asm volatile("" : "+v"(ua0), "+v"(ub0), "+v"(ua1), "+v"(ub1));
        const float a0 = __builtin_bit_cast(float, ua0), b0 = __builtin_bit_cast(float, ub0);
        const float a1 = __builtin_bit_cast(float, ua1), b1 = __builtin_bit_cast(float, ub1);
        // whole quads at a time: the compiler turns these into packed multiplies and FMAs
        const f32x4 a0v = {a0, a0, a0, a0}, a1v = {a1, a1, a1, a1}, b0v = {b0, b0, b0, b0}, b1v = {b1, b1, b1, b1};
        dW3[0] = __builtin_elementwise_fma(a0v, A.h2, dW3[0]);
        dW3[1] = __builtin_elementwise_fma(a1v, A.h2, dW3[1]);
        dW3[0] = __builtin_elementwise_fma(b0v, B.h2, dW3[0]);
        dW3[1] = __builtin_elementwise_fma(b1v, B.h2, dW3[1]);
        A.gz2 = mul4<!BF>(__builtin_elementwise_fma(w3r[1], a1v, mul4<!BF>(w3r[0], a0v)), A.m2);
        B.gz2 = mul4<!BF>(__builtin_elementwise_fma(w3r[1], b1v, mul4<!BF>(w3r[0], b0v)), B.m2);
        db2 = add4<!BF>(db2, A.gz2);
        db2 = add4<!BF>(db2, B.gz2);
    };
    auto V4 = [&](Tile &t, int buf) {
        if constexpr (OB) {             // the terms of gz2, the right factor of dW2: [h|m] in one slot, l in half of the next
            char *tw = ser_b + 4 * (GZ_W + TB_WORDS * buf) + tw_off;
            unsigned (&h)[2] = t.g2h, (&m)[2] = t.g2m, (&l)[2] = t.g2l;
            split_pair3(t.gz2[0], t.gz2[1], h[0], m[0], l[0]);
            split_pair3(t.gz2[2], t.gz2[3], h[1], m[1], l[1]);
            *reinterpret_cast<u32x4 *>(tw + S_G2) = u32x4{h[0], h[1], m[0], m[1]};
            *reinterpret_cast<u32x2 *>(tw + S_G2 + SLOT_B) = u32x2{l[0], l[1]};
            __builtin_amdgcn_wave_barrier();
            return;
        }
        *reinterpret_cast<f32x4 *>(ser_b + 4 * (GZ_W + TILE_W * buf) + wr32 + 128) = t.gz2;
    };
    auto S5 = [&](Tile &t) {   // gh1^T = W2^T gz2^T  (two row tiles of 16 h1-neurons)
        t.e0 = f32x4{0.f, 0.f, 0.f, 0.f};
        t.e1 = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (OB) {        // three K = 32 instructions per row tile: [h|l] x [l|h], [h|m] x [m|h], [h|m] x [h|m]
            const u32x4 b_lh = {t.g2l[0], t.g2l[1], t.g2h[0], t.g2h[1]}, b_mh = {t.g2m[0], t.g2m[1], t.g2h[0], t.g2h[1]};
            const u32x4 b_hm = {t.g2h[0], t.g2h[1], t.g2m[0], t.g2m[1]};
            t.e0 = mfma_bf(T2[0], b_lh, t.e0);
            t.e1 = mfma_bf(T2[2], b_lh, t.e1);
            t.e0 = mfma_bf(T2[1], b_mh, t.e0);
            t.e1 = mfma_bf(T2[3], b_mh, t.e1);
            t.e0 = mfma_bf(T2[1], b_hm, t.e0);
            t.e1 = mfma_bf(T2[3], b_hm, t.e1);
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t.e0 = mfma4(A2T[0][r], t.gz2[r], t.e0);
            t.e1 = mfma4(A2T[1][r], t.gz2[r], t.e1);
        }
    };
    auto V5 = [&](Tile &t, int buf) {
        t.e0 = mul4<!BF>(t.e0, t.m1a);
        t.e1 = mul4<!BF>(t.e1, t.m1b);
        if constexpr (OB) {             // the terms of gz1, the right factor of dW1, over the slots of h1 (dW2 has read them)
            char *tw = ser_b + 4 * (GZ_W + TB_WORDS * buf) + tw_off;
            Split3 sp;
            split8(sp, t.e0, t.e1);
            *reinterpret_cast<u32x4 *>(tw + S_G1) = sp.h;
            *reinterpret_cast<u32x4 *>(tw + S_G1 + SLOT_B) = sp.m;
            *reinterpret_cast<u32x4 *>(tw + S_G1 + 2 * SLOT_B) = sp.l;
            __builtin_amdgcn_wave_barrier();
            return;
        }
        char *gz_b = ser_b + 4 * (GZ_W + TILE_W * buf);
        *reinterpret_cast<f32x4 *>(gz_b + wr32) = t.e0;
        *reinterpret_cast<f32x4 *>(gz_b + wr32 + 64) = t.e1;
    };
    // OB: dW2 += h1 (x) gz2 of one tile
    auto S6b = [&](int buf) {
        const int tb = 4 * TB_WORDS * buf;
        OpR g2;
        read_op(g2, tra, tb, S_G2, S_G2 + 8, S_G2 + SLOT_B);
        OpL h1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) read_op(h1[u], tra, tb, S_H1 + 8 * u, S_H1 + SLOT_B + 8 * u, SP3 ? S_H1 + 2 * SLOT_B + 8 * u : -1);
        __builtin_amdgcn_wave_barrier();
        dW2[0] = outer3(dW2[0], h1[0], g2);
        dW2[1] = outer3(dW2[1], h1[1], g2);
    };
    // OB: dW1 += x (x) gz1 of one tile; the x operand comes straight from the words of the series
    auto S6a = [&](int toff, int buf) {
        const int tb = 4 * TB_WORDS * buf;
        OpR g1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) read_op(g1[t], tra, tb, S_G1 + 8 * t, S_G1 + SLOT_B + 8 * t, S_G1 + 2 * SLOT_B + 8 * t);
        OpL x[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            unsigned a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const unsigned *>(ser_b + xa[u] + toff + 4 * i);
                if (SP3) b[i] = *reinterpret_cast<const unsigned *>(ser_b + xb[u] + toff + 4 * i);
            }
            const unsigned h01 = __builtin_amdgcn_perm(a[1], a[0], 0x07060302u), h23 = __builtin_amdgcn_perm(a[3], a[2], 0x07060302u);
            x[u].hm = u32x4{h01, h23, __builtin_amdgcn_perm(a[1], a[0], 0x05040100u), __builtin_amdgcn_perm(a[3], a[2], 0x05040100u)};
            if (SP3) x[u].hl = u32x4{h01, h23, __builtin_amdgcn_perm(b[1], b[0], 0x07060302u), __builtin_amdgcn_perm(b[3], b[2], 0x07060302u)};
            else x[u].hl = u32x4{h01, h23, 0u, 0u};
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) dW1[u][t] = outer3(dW1[u][t], x[u], g1[t]);
    };
    auto S6 = [&](int toff, int buf) {   // samples move from lanes to the MFMA K axis through the LDS tiles
        const char *gz_b = ser_b + 4 * (GZ_W + TILE_W * buf);
        const char *h1_b = ser_b + 4 * (H1_W + TILE_W * buf);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // samples 4q + g of this tile
            float x0 = *reinterpret_cast<const float *>(ser_b + aoff[0] + toff + 16 * q);
            float x1 = *reinterpret_cast<const float *>(ser_b + aoff[1] + toff + 16 * q);
            float gza = *reinterpret_cast<const float *>(gz_b + rd32[q]);
            float gzb = *reinterpret_cast<const float *>(gz_b + rd32[q] + 64);
            dW1[0][0] = mfma4(x0, gza, dW1[0][0]);
            dW1[0][1] = mfma4(x0, gzb, dW1[0][1]);
            dW1[1][0] = mfma4(x1, gza, dW1[1][0]);
            dW1[1][1] = mfma4(x1, gzb, dW1[1][1]);
            float ha = *reinterpret_cast<const float *>(h1_b + rd32[q]);
            float hb = *reinterpret_cast<const float *>(h1_b + rd32[q] + 64);
            float g2 = *reinterpret_cast<const float *>(gz_b + rd32[q] + 128);
            dW2[0] = mfma4(ha, g2, dW2[0]);
            dW2[1] = mfma4(hb, g2, dW2[1]);
        }
    };
#if defined(ABL_TIMING) || defined(ABL_CLOCK)
    // experiment: shader-clock cycles and 100 MHz ticks between here and the end of the wave's loop
    const unsigned long long clk0 = __builtin_readcyclecounter(), real0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef ABL_TIMING
    // experiment: cycles between phase boundaries, summed per wave and added into queue[16 + k]
    unsigned long long tacc[20] = {0}, tlast = clk0;
    int tk = 0;
#define PHASE()                                            \
    {                                                      \
        __builtin_amdgcn_sched_barrier(0);                 \
        unsigned long long now = __builtin_readcyclecounter(); \
        tacc[tk] += now - tlast;                           \
        tlast = now;                                       \
        ++tk;                                              \
        __builtin_amdgcn_sched_barrier(0);                 \
    }
#else
#define PHASE() __builtin_amdgcn_sched_barrier(0)
#endif

    // experiment switch: the VALU stage of a phase spread between its bf16 MFMAs (as in the forward kernel)
#ifndef GEN_BWD_MIX
#define GEN_BWD_MIX 0
#endif
#define BMIX(NM, NV)                                                 \
    if constexpr (OB && GEN_BWD_MIX) {                               \
        _Pragma("unroll") for (int i_ = 0; i_ < (NM); ++i_) {        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       \
            __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);    \
        }                                                            \
    }
    // The HBM reads of a chunk (69 IQ points, 64 targets or upstream gradients) are issued one chunk
    // ahead and parked in LDS at the top of the next iteration: no load latency inside the tile code.
    const float2 *gsrc = MSE ? target : grad_out;
    constexpr int CSTEP = PA ? PA_CHUNK : CHUNK, PSTEP = PA ? PA_PAIR : 32;   // outputs a chunk / a pair advances by
    const int64_t LT = PA ? pa.OL : L;                                        // row length of the target / upstream gradient
    // Units of UNIT consecutive chunks are handed out through a device counter (first come, first
    // served): waves on faster or earlier-started CUs take more units, none idles at the end.  The
    // counter is read one unit ahead, so its latency stays hidden.  n_chunks < 2^31 (checked on the host).
#ifndef GEN_BWD_UNIT
#define GEN_BWD_UNIT 4
#endif
    constexpr unsigned UNIT = GEN_BWD_UNIT;
    const unsigned n_waves = gridDim.x * WAVES, cpr32 = (unsigned)cpr, n_chunks32 = (unsigned)n_chunks;
    // The counter is bumped with a hand-issued atomic whose result stays in a vector register until the
    // unit is started (one unit later): the compiler's own atomic sequence waits for the round trip at
    // once.  The asm narrows exec to lane 0 itself (the loop is wave-uniform, all lanes active), so no
    // branch and no merge of register values surrounds it, and it takes next_raw as an in/out operand.
    // The compiler does not count this load, hence the explicit wait, in one asm statement with the read
    // of the register: nothing can touch the register between the issue and the wait.  By then the only
    // other loads in flight are the next chunk's, which are needed at that point anyway.
    unsigned static_next = blockIdx.x * WAVES + wave;     // queue == NULL: units in fixed round-robin order
    auto take = [&](unsigned v) -> unsigned {
        if (queue == nullptr) {
            v = static_next;
            static_next += n_waves;
        } else {
            const unsigned one = 1u;
            unsigned long long saved;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, off sc0\n\ts_mov_b64 exec, %1"
                         : "+v"(v), "=&s"(saved)
                         : "v"(queue), "v"(one)
                         : "memory");
        }
        return v;
    };
    auto uniform = [&](unsigned v) {
        unsigned raw;
        if (queue == nullptr) {
            raw = v;
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %0, %1" : "=s"(raw) : "v"(v) : "memory");
        }
        return n_waves + raw;
    };
    unsigned unit = blockIdx.x * WAVES + wave, next_raw = 0, in_unit = 0;
    unsigned chunk = unit * UNIT;
    ChunkIQ pre;
    float2 pre_tg = make_float2(0.f, 0.f);
    int64_t t0 = 0;
    if (chunk < n_chunks32) {
        const unsigned row = chunk / cpr32;
        t0 = (int64_t)(chunk - row * cpr32) * CSTEP;
        load_chunk_iq(pre, iq + row * S, t0, S, lane);
        if (lane < CSTEP && t0 + lane < LT) pre_tg = gsrc[row * LT + t0 + lane];
    }
    while (chunk < n_chunks32) {
#ifdef ABL_TIMING
        { unsigned long long now = __builtin_readcyclecounter(); tacc[18] += now - tlast; tlast = now; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        { unsigned long long now = __builtin_readcyclecounter(); tacc[17] += now - tlast; tlast = now; }
#endif
        if constexpr (OB) {
            commit_point<QAT, DIG, SP3>(ser, pre.p[0], lane, s0);
            if (lane < 5) commit_point<QAT, DIG, SP3>(ser, pre.p[1], 64 + lane, s0);
        } else if constexpr (DIG) commit_series_digits(ser, ser + DSER_W, pre, lane, s0);
        else if constexpr (SP3) commit_series_split(ser, ser + DSER_W, ser + DSER_W + SPB_BWD, pre, lane);
        else commit_series<QAT>(ser, pre, lane, s0);
        tg_lds[lane] = pre_tg;
#ifdef ABL_TIMING
        { __builtin_amdgcn_sched_barrier(0); unsigned long long now = __builtin_readcyclecounter(); tacc[16] += now - tlast; tlast = now; __builtin_amdgcn_sched_barrier(0); }
#endif
        const int64_t t0_cur = t0;
        // the counter is bumped right after the top-of-loop wait of a unit's second chunk, so the atomic
        // has two chunks to return before anything waits on the memory counter again
        if (in_unit == 1) next_raw = take(next_raw);
        if (++in_unit == UNIT) {
            in_unit = 0;
            unit = uniform(next_raw);
        }
        chunk = unit * UNIT + in_unit;
        pre_tg = make_float2(0.f, 0.f);
        if (chunk < n_chunks32) {
            const unsigned row = chunk / cpr32;
            t0 = (int64_t)(chunk - row * cpr32) * CSTEP;
            load_chunk_iq(pre, iq + row * S, t0, S, lane);
            if (lane < CSTEP && t0 + lane < LT) pre_tg = gsrc[row * LT + t0 + lane];
        }
        __builtin_amdgcn_wave_barrier();
        // (measured: spreading the VALU stage of a phase between its bf16 MFMAs with sched_group_barrier, as the forward
        // kernel does, is 1-2 % slower here)
        // (measured: starting FC1 of the second pair inside the output stage of the first, the one stretch
        // of a pair without MFMA work of its own, is 3% slower: the MFMAs hold up the VALU work behind them)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // tiles beyond L inside a started pair run with zero upstream gradient: they add nothing
            if (t0_cur + PSTEP * p >= LT) break;
#ifdef ABL_TIMING
            tk = 0;
            PHASE();   // slot 0: everything outside the pair code (staging, prefetch, loop control)
#endif
            const int cA = 4 * PSTEP * p, cB = cA + 64;   // byte offsets of the two tiles in the series
            Tile A, B;
            Fc1Operand bA, bB;
            F1(bA, cA);
            F1(bB, cB);
            PHASE();
            S1(A.a0, A.a1, bA);
            PHASE();
            S1(B.a0, B.a1, bB);
            V1(A, 0);
            BMIX(SP3 ? 12 : 8, SP3 ? 5 : 8)
            PHASE();
            S2(A);
            V1(B, 1);
            BMIX(SP3 ? 6 : 4, SP3 ? 10 : 16)
            PHASE();
            S2(B);
            V2(A);
            PHASE();
            V2(B);
            V3P(A, B, p, t0_cur);
            V4(A, 0);
            PHASE();
            S5(A);
            V4(B, 1);
            PHASE();
            if constexpr (OB) {
                // h1 and gz1 of a tile share their slots: dW2 reads h1 before the gz1 terms are parked over it
                S5(B);
                S6b(0);
                PHASE();
                S6b(1);
                V5(A, 0);
                BMIX(6, 8)
                PHASE();
                S6a(cA, 0);
                V5(B, 1);
                BMIX(12, 4)
                PHASE();
                S6a(cB, 1);
                PHASE();
            } else {
                S5(B);
                V5(A, 0);
                PHASE();
                S6(cA, 0);
                V5(B, 1);
                PHASE();
                S6(cB, 1);
                PHASE();
            }
        }
    }
#undef PHASE
#ifdef ABL_TIMING
    if (lane == 0)
        for (int k = 0; k < 20; ++k) atomicAdd(reinterpret_cast<unsigned long long *>(queue + 16) + k, tacc[k]);
#endif
#if defined(ABL_TIMING) || defined(ABL_CLOCK)
    if (lane == 0) {   // slots 20, 21: their ratio is the shader clock in units of 100 MHz
        atomicAdd(reinterpret_cast<unsigned long long *>(queue + 16) + 20, __builtin_readcyclecounter() - clk0);
        atomicAdd(reinterpret_cast<unsigned long long *>(queue + 16) + 21, __builtin_amdgcn_s_memrealtime() - real0);
    }
#endif

    // ---- partial gradient row of this wave -> LDS, the workgroup's four rows are summed, one row per
    // workgroup goes to the slab (the staging regions are dead: every wave has left its loop)
    __syncthreads();
    float *comb = lds;
    float *rowp = comb + wave * G_SLAB;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int slot = 16 * u + 4 * g + r, ref = slot_ref(slot);
                if (ref >= 0) rowp[G_W1 + (16 * t + j) * 30 + ref] = DIG ? dW1[u][t][r] * s0.s : dW1[u][t][r];
                else if (slot == 30) rowp[G_B1 + 16 * t + j] = dW1[u][t][r];   // sum over samples of 1 * gz1
            }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) rowp[G_W2 + j * 32 + 16 * u + 4 * g + r] = DIG ? dW2[u][r] * s1.s : dW2[u][r];
    // per-lane accumulators: sum over the 16 sample lanes of each lane group
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dW3[0][r] += __shfl_xor(dW3[0][r], off);
            dW3[1][r] += __shfl_xor(dW3[1][r], off);
            db2[r] += __shfl_xor(db2[r], off);
        }
        db3 += __shfl_xor(db3, off);
    }
    db3 += __shfl_xor(db3, 32);                 // lane groups g and g ^ 2 hold the same output index
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rowp[G_W3 + 4 * g + r] = dW3[0][r];
            rowp[G_W3 + 16 + 4 * g + r] = dW3[1][r];
            rowp[G_B2 + 4 * g + r] = db2[r];
        }
        if (g < 2) rowp[G_B3 + g] = db3;
    }
    loss = wave_sum(loss);
    if (lane == 0) rowp[G_NP] = loss;
    __syncthreads();
    float *out_row = slab + (int64_t)blockIdx.x * G_SLAB;
    for (int e = threadIdx.x; e <= G_NP; e += WAVES * 64)
        out_row[e] = (comb[e] + comb[G_SLAB + e]) + (comb[2 * G_SLAB + e] + comb[3 * G_SLAB + e]);
}

}  // namespace

int g_gen_deterministic = 0;   // dpd_set_option("gen_deterministic", 0 / 1)

static int mfma_grid(int64_t n_chunks, int per_cu = 2) {
    int64_t blocks = (n_chunks + WAVES - 1) / WAVES;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;  // resident workgroups per CU
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// consecutive chunks per unit of the forward walk: as many as keep every wave busy, at most a row's worth of 16
static unsigned fwd_unit(int64_t n_chunks, unsigned grid) {
    const int64_t per_wave = n_chunks / ((int64_t)grid * WAVES);
    return (unsigned)(per_wave < 1 ? 1 : per_wave > 16 ? 16 : per_wave);
}

int gen_mfma_forward(const GenProblem &p, int depth, float *out, hipStream_t st) {
    const float2 *iq = reinterpret_cast<const float2 *>(p.x);
    if (p.qat && depth == 0) {
        // split the stream into an unaligned leading point, float4 pairs and a trailing point
        const int64_t n_points = p.B * p.S;
        const int64_t head = ((reinterpret_cast<uintptr_t>(iq) & 15) != 0 && n_points > 0) ? 1 : 0;
        const int64_t n4 = (n_points - head) / 2, tail = n_points - head - 2 * n4;
        hipLaunchKernelGGL(k_iq_absmax, dim3(grid_for(n4, 1024)), dim3(256), 0, st,
                           reinterpret_cast<const float4 *>(iq + head), n4, head ? iq : nullptr,
                           tail ? iq + n_points - 1 : nullptr, p.absmax);
        DPD_LAUNCH_CHECK();
        return DPD_OK;
    }
    const int64_t cpr = (p.L + CHUNK - 1) / CHUNK, n_chunks = cpr * p.B;
    DPD_REQUIRE(n_chunks < ((int64_t)1 << 31) - 65536, DPD_ERR_SHAPE, "gen_forward: %lld chunks exceed the 32-bit chunk counter", (long long)n_chunks);
    const dim3 grid(mfma_grid(n_chunks, FWD_WAVES_PER_SIMD)), block(WAVES * 64);
    const unsigned unit = fwd_unit(n_chunks, grid.x);
    float2 *o = reinterpret_cast<float2 *>(out);
    const PaArgs none{nullptr, nullptr, 0, 0, nullptr, 0};
#define GO(Q, D) hipLaunchKernelGGL((k_gen_fwd_mfma<Q, D, 0>), grid, block, 0, st, iq, p.S, p.L, cpr, n_chunks, p.wq, \
                                    p.slope, p.absmax, o, none, unit)
    if (!p.qat) GO(false, 4);
    else if (depth == 1) GO(true, 1);
    else if (depth == 2) GO(true, 2);
    else GO(true, 3);
#undef GO
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

// Fused chain generator -> PA twin; out (generator output) may be NULL.  K * M <= 128, M <= 16.
int g_chain_carry = 2;         // dpd_set_option("chain_carry", 0 off / 1 on / 2 automatic)

int gen_mfma_pa_forward(const GenProblem &p, const float *cre, const float *cim, int K, int M, float *out, float *y,
                        hipStream_t st) {
    const float2 *iq = reinterpret_cast<const float2 *>(p.x);
    const int64_t OL = p.L - M + 1;
    // whole rows per wave without overlap between chunks ("carry") when the rows fill the resident waves evenly
    // (and there are enough of them to fill the chip); otherwise chunks of 64 - (M - 1) outputs that overlap
    bool carry = false;
    if (K == 4 && M == 5 && g_chain_carry != 0) {
        const int64_t waves = (int64_t)mfma_grid(p.B, FWD_WAVES_PER_SIMD) * WAVES;
        const int64_t rows_per_wave = (p.B + waves - 1) / waves;
        carry = g_chain_carry == 1 || (p.B >= 2048 && 10 * p.B >= 9 * rows_per_wave * waves);
    }
    const int64_t step = carry ? CHUNK : CHUNK - (M - 1), rowlen = carry ? p.L : OL;
    const int64_t cpr = (rowlen + step - 1) / step, n_chunks = cpr * p.B;
    DPD_REQUIRE(n_chunks < ((int64_t)1 << 31) - 65536 && cpr < ((int64_t)1 << 31), DPD_ERR_SHAPE,
                "gen_pa_forward: %lld chunks exceed the 32-bit chunk counter", (long long)n_chunks);
    const dim3 grid(mfma_grid(carry ? p.B : n_chunks, FWD_WAVES_PER_SIMD)), block(WAVES * 64);
    const unsigned unit = carry ? (unsigned)cpr : fwd_unit(n_chunks, grid.x);
    const PaArgs pa{cre, cim, K, M, reinterpret_cast<float2 *>(y), OL};
    float2 *o = reinterpret_cast<float2 *>(out);
#define GO(Q, V) hipLaunchKernelGGL((k_gen_fwd_mfma<Q, 4, V>), grid, block, 0, st, iq, p.S, p.L, cpr, n_chunks, p.wq, p.slope, \
                                    p.absmax, o, pa, unit)
    if (carry)            { if (p.qat) GO(true, 3); else GO(false, 3); }
    else if (K == 4 && M == 5) { if (p.qat) GO(true, 2); else GO(false, 2); }
    else                  { if (p.qat) GO(true, 1); else GO(false, 1); }
#undef GO
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

int gen_mfma_backward(const GenProblem &p, const float *grad_out, const float *target, float *slab, int *n_partials,
                      hipStream_t st, const GenPaChain *chain) {
    const float2 *iq = reinterpret_cast<const float2 *>(p.x);
    // chain: chunks of 56 PA outputs over rows of OL = L - 4; otherwise chunks of 64 generator outputs
    const int64_t rowlen = chain ? p.L - (PA_M - 1) : p.L, cstep = chain ? PA_CHUNK : CHUNK;
    const int64_t cpr = (rowlen + cstep - 1) / cstep, n_chunks = cpr * p.B;
    DPD_REQUIRE(n_chunks < ((int64_t)1 << 31) - 4096 && p.S < ((int64_t)1 << 31), DPD_ERR_SHAPE,
                "gen_backward: %lld chunks exceed the 32-bit chunk counter", (long long)n_chunks);
    DPD_REQUIRE(!chain || (chain->M == PA_M && chain->K >= 1 && chain->K <= 4 && rowlen >= 1), DPD_ERR_INVALID_ARG,
                "gen_backward: the fused chain is built for M = 5 taps and K <= 4 orders");
    const int gridn = mfma_grid(n_chunks, BWD_WAVES_PER_SIMD);
    *n_partials = gridn;
    // chunk counter: one word behind the largest slab this call can write (the workspace is far larger);
    // none in deterministic mode (fixed assignment of chunks to waves: bitwise reproducible gradients)
    unsigned *queue = g_gen_deterministic ? nullptr : reinterpret_cast<unsigned *>(slab + (size_t)1024 * G_SLAB);
    if (queue) DPD_HIP(hipMemsetAsync(queue, 0, sizeof(unsigned), st));
    const dim3 grid(gridn), block(WAVES * 64);
    const bool mse = grad_out == nullptr;
    const double n_elem = 2.0 * (double)p.B * (double)rowlen;          // elements of the tensor the loss averages over
    const float mse_scale = mse ? (float)(2.0 / n_elem) : 0.f;
    const float2 *go = reinterpret_cast<const float2 *>(grad_out), *tg = reinterpret_cast<const float2 *>(target);
    const PaBwdArgs pa = chain ? PaBwdArgs{chain->cre, chain->cim, chain->K, rowlen} : PaBwdArgs{nullptr, nullptr, 0, 0};
#define GO(Q, M, P) hipLaunchKernelGGL((k_gen_bwd_mfma<Q, M, P>), grid, block, 0, st, iq, p.S, p.L, cpr, n_chunks, p.wq, \
                                       p.slope, p.absmax, go, tg, mse_scale, slab, queue, pa)
    if (chain) {
        if (p.qat) { if (mse) GO(true, true, true); else GO(true, false, true); }
        else       { if (mse) GO(false, true, true); else GO(false, false, true); }
    } else {
        if (p.qat) { if (mse) GO(true, true, false); else GO(true, false, false); }
        else       { if (mse) GO(false, true, false); else GO(false, false, false); }
    }
#undef GO
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}
