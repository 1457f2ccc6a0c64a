// Bit-exact Q8.8 fixed-point DPD inference on the integer VALU.
//
// Contract (every step round-half-even + saturate to int16, as quantize_to_fixed_point in the
// reference's utils/quantization.py:88-120):
//   IQ int16 Q1.15 -> tap features (blocked order of models/tdnn_generator.py:176-193) in Q8.8:
//       I, Q       rhe(i / 128)
//       |x|        rhe(sqrt(s) / 128),  s = i^2 + q^2     (exact integer square root)
//       |x|^2      rhe(s / 2^22)
//       |x|^4      rhe(s^2 / 2^52)
//   layer: acc(Q9.23) = sum a(Q8.8) * w(Q1.15) + (b(Q8.8) << 15);  z = sat16(rhe(acc / 2^15))
//   LeakyReLU(0.2): z > 0 ? z : rhe(z / 5);  output: table[clamp(rhe((z + 1024) * 255 / 2048), 0, 255)]
// Products of two int16 fit int32; sums are kept in 32 bits (v_dot2_i32_i16, two MACs per lane
// and instruction) when the host has proven from the weights that no layer can overflow, else in
// 64 bits.
//
// Layout of the work.  A workgroup takes a segment of 512 consecutive outputs of one row (two per
// lane).  The per-time-point features are computed once per point (not once per tap) and parked in
// LDS already paired the way the dot products want them: P_x[u] = (x[u], x[u - 1]) as two int16, so
// the 15 activation pairs of a sample are 15 conflict-free ds_read_b32.  An int16 weight row is a
// row of K/2 dwords in memory as it stands (pairs along K), so the weights are never staged: they
// are read with scalar loads (wave-uniform addresses, scalar cache) straight into the SGPR operand
// of v_dot2_i32_i16, each dword feeding both samples of the lane.
#include "common.h"

namespace {

typedef short short2_t __attribute__((ext_vector_type(2)));
constexpr int TPB = 256;
constexpr int SPT = 2;               // samples per lane
constexpr int SEG = TPB * SPT;       // outputs per segment
constexpr int PTS = SEG + 5;         // input points per segment

DEV int rhe_shift(int v, int sh) { return (v + ((1 << (sh - 1)) - 1) + ((v >> sh) & 1)) >> sh; }
DEV long long rhe_shift64(long long v, int sh) { return (v + ((1ll << (sh - 1)) - 1) + ((v >> sh) & 1)) >> sh; }
DEV int sat16(int v) { return min(max(v, -32768), 32767); }
DEV int sat16_64(long long v) { return (int)min(max(v, -32768ll), 32767ll); }

DEV unsigned isqrt32(unsigned s) {
    unsigned t = (unsigned)sqrtf((float)s);
    while ((unsigned long long)t * t > s) --t;
    while ((unsigned long long)(t + 1) * (t + 1) <= s) ++t;
    return t;
}

DEV int envelope_q88(unsigned s) {
    const unsigned t = isqrt32(s);
    const int q = (int)(t >> 7), rem = (int)(t & 127);
    const bool exact = (unsigned long long)t * t == s;
    const bool up = rem > 64 || (rem == 64 && (!exact || (q & 1)));
    return q + (up ? 1 : 0);
}

// z > 0 ? z : -((-z + 2) / 5) for int16-range z, without a division: for z <= 0 the second form is
// ceil((z - 2) / 5) = floor((z - 2) * 52429 / 2^18) + 1 (52429 / 2^18 = 1/5 + 1/1310720, exact for
// |z| < 2^16), it is >= z there, and for z > 0 the same expression stays <= z: one max selects.
DEV int lrelu_q(int z) { return max(z, (__mul24(z, 52429) + (262144 - 2 * 52429)) >> 18); }
// The same for a z that has not been saturated yet (|z| < 2^16: what a 32-bit accumulator can leave after the
// shift by 15).  Saturation and LeakyReLU commute except below -32768, where the saturated value gives
// lrelu(-32768) = -6554: one three-operand max covers it.  The upper side is left to the saturating pack.
DEV int lrelu_q_unsat(int z) {
    const int t = (__mul24(z, 52429) + (262144 - 2 * 52429)) >> 18;
    return max(max(z, t), -6554);
}
// two int32 -> two int16 with saturation, packed: one instruction (v_cvt_pk_i16_i32)
DEV int pack2_sat(int lo, int hi) { return __builtin_bit_cast(int, __builtin_amdgcn_cvt_pk_i16(lo, hi)); }

DEV int pack2(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }

// One dense layer for the SPT samples of a lane.  wp: the layer's int16 weights viewed as dwords
// (row j = KP dwords), bp: its int16 biases viewed as dwords.  a[s][k]: activation pairs.
// ACT: LeakyReLU and pair the outputs again (z[s][j / 2]); else raw z[s][j].
template <bool ACC64, bool ACT, int O, int KP, int ZN>
DEV void dense_q(const int *__restrict__ wp, const int *__restrict__ bp, const int (&a)[SPT][KP], int (&z)[SPT][ZN]) {
    int done[O + 2][SPT];   // done[j + 2]: the results of row j; done[0], done[1]: the layer's input
#pragma unroll
    for (int s = 0; s < SPT; ++s) done[0][s] = done[1][s] = a[s][0];
#pragma unroll
    for (int j = 0; j < O; ++j) {
        // The row offset passes through an empty asm that also takes a result of the row before the
        // previous one (for both samples of the lane): the weight loads of row j can then neither be hoisted out of the segment loop
        // (the weights are loop-invariant) nor gathered at the top of the layer (either would need all
        // 752 weight dwords in SGPRs at once); they are issued while row j - 1 computes.
        int row = j * KP;
        static_assert(SPT == 2, "the asm below names both samples");
        asm volatile("" : "+s"(row) : "v"(done[j][0]), "v"(done[j][1]));
        int w[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) w[k] = wp[row + k];              // scalar loads
        const int bias = (int)(short)(bp[j >> 1] >> (16 * (j & 1)));
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            int r;
            if (ACC64) {
                long long acc = (long long)bias << 15;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    acc += (long long)((short)(a[s][k] & 0xffff)) * (short)(w[k] & 0xffff);
                    acc += (long long)(a[s][k] >> 16) * (w[k] >> 16);
                }
                r = sat16_64(rhe_shift64(acc, 15));
            } else {
                int acc = bias << 15;
#pragma unroll
                for (int k = 0; k < KP; ++k)
                    acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, a[s][k]), __builtin_bit_cast(short2_t, w[k]),
                                                 acc, false);
                // with the activation behind it the saturation moves into the pack of the pair (see lrelu_q_unsat)
                r = ACT ? rhe_shift(acc, 15) : sat16(rhe_shift(acc, 15));
            }
            done[j + 2][s] = r;
            if (ACT) {
                r = lrelu_q_unsat(r);
                if (j & 1) z[s][j >> 1] = pack2_sat(z[s][j >> 1], r);
                else z[s][j >> 1] = r;
            } else {
                z[s][j] = r;
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // one weight row in flight at a time: keeps the scalar loads from piling up
    }
}

template <bool ACC64>
__global__ __launch_bounds__(TPB) void k_q88(const int *__restrict__ iq, int64_t S, int64_t L, int64_t segs_per_row,
                                             int64_t n_segs, const int *__restrict__ w1, const int *__restrict__ b1,
                                             const int *__restrict__ w2, const int *__restrict__ b2,
                                             const int *__restrict__ w3, const int *__restrict__ b3,
                                             const short *__restrict__ lut, int *__restrict__ out) {
    __shared__ int s_iq[PTS], s_env[PTS + 1], s_e2[PTS + 1], s_e4[PTS + 1], s_lut[256];
    const int tid = threadIdx.x;
    s_lut[tid] = lut[tid];
    for (int64_t seg = blockIdx.x; seg < n_segs; seg += gridDim.x) {
        const int64_t row = seg / segs_per_row, t0 = (seg - row * segs_per_row) * SEG;
        const int *p = iq + row * S + t0;    // one int = (I, Q) int16 pair; p[t + 5] is the current sample of output t
        __syncthreads();                     // the previous segment's reads (and the table) are done
        for (int u = tid; u < PTS; u += TPB) {
            const int v = (t0 + u < S) ? p[u] : 0;
            const int i = (short)(v & 0xffff), q = v >> 16;
            const unsigned ss = (unsigned)(i * i) + (unsigned)(q * q);
            s_iq[u] = pack2(sat16(rhe_shift(i, 7)), sat16(rhe_shift(q, 7)));
            const int env = envelope_q88(ss);
            const int e2 = (int)(((unsigned long long)ss + ((1u << 21) - 1) + ((ss >> 22) & 1u)) >> 22);
            const int e4 = (int)(((unsigned long long)ss * ss + (1ull << 51)) >> 52);
            // low half of pair u, high half of pair u + 1
            reinterpret_cast<short *>(s_env)[2 * u] = (short)env;
            reinterpret_cast<short *>(s_env)[2 * u + 3] = (short)env;
            reinterpret_cast<short *>(s_e2)[2 * u] = (short)e2;
            reinterpret_cast<short *>(s_e2)[2 * u + 3] = (short)e2;
            reinterpret_cast<short *>(s_e4)[2 * u] = (short)e4;
            reinterpret_cast<short *>(s_e4)[2 * u + 3] = (short)e4;
        }
        __syncthreads();
        int a1[SPT][15];
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const int u5 = tid + TPB * s + 5;    // point of tap 0
            a1[s][0] = s_iq[u5];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                a1[s][1 + m] = s_env[u5 - 2 * m];
                a1[s][4 + m] = s_e2[u5 - 2 * m];
                a1[s][7 + m] = s_e4[u5 - 2 * m];
            }
#pragma unroll
            for (int g = 1; g < 6; ++g) a1[s][9 + g] = s_iq[u5 - g];
        }
        int a2[SPT][16], a3[SPT][8], z3[SPT][2];
        dense_q<ACC64, true, 32, 15, 16>(w1, b1, a1, a2);
        dense_q<ACC64, true, 16, 16, 8>(w2, b2, a2, a3);
        dense_q<ACC64, false, 2, 8, 2>(w3, b3, a3, z3);
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            int o[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                int idx = rhe_shift((z3[s][c] + 1024) * 255, 11);
                idx = min(max(idx, 0), 255);
                o[c] = s_lut[idx];
            }
            const int64_t t = t0 + tid + TPB * s;
            if (t < L) out[row * L + t] = pack2(o[0], o[1]);
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// Matrix-core form: the same integer contract on v_mfma_i32_16x16x64_i8, every int16 as two 8-bit digits.
//   activation a = 256 ah + (al + 128),  ah = a >> 8 in [-128, 127],  al = (a & 255) - 128 in [-128, 127]
//   weight     w = 256 wh + wl,          wl = (int8)(w & 255),        wh = (w - wl) >> 8 in [-128, 127] for w <= 32639
//   sum a w = 65536 sum ah wh + 256 sum (ah wl + al wh) + sum al wl + 128 sum w
// Three int32 accumulators per result tile (HH, X, LL), each far from overflow; the constant 128 sum_k w[j][k] joins the
// bias (the start value of LL); the combination and everything behind it is the arithmetic of the VALU form.
// The host checks the weight range (dpd_q88_digits_fit) and the 32-bit proof before it selects this form.
//
// Geometry as in the float generator kernels: a wave owns 64 consecutive outputs of a row (4 tiles of 16), products are
// computed transposed (D = W X^T): lane (g, j) of a result holds neurons 4g..4g+3 of sample j, and the 16 bytes a lane
// feeds into the next product are the digits of exactly the values it holds (the K order of the next layer's weight
// image is chosen to match), so no value moves between lanes.  K = 64 = 4 lane groups x 16 bytes; groups 0, 1 carry
// high digits, groups 2, 3 low digits (FC1), or bytes 0..7 high and 8..15 low digits of the lane's eight values (FC2).
// FC1 operands: the five per-point series live in LDS as byte planes [digit][series][point]; the six taps of a series
// are six consecutive bytes, fetched as aligned dwords and shifted into place with v_alignbyte_b32.
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int MW = 4;                 // waves per workgroup
#ifndef Q88_MFMA_WAVES
#define Q88_MFMA_WAVES 4              // resident waves per SIMD (register budget 128; measured 3 / 4 / 5: 0.365 / 0.355 / 0.354 ms, 5 spills)
#endif
constexpr int PLANE = 96;             // bytes per digit plane: 69 points + the over-read of the last dword, multiple of 4
constexpr int WAVE_LDS = 10 * PLANE;  // [digit 0..1][series 0..4]
DEV i32x4 mfma_i8(i32x4 a, i32x4 b, i32x4 c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
DEV int digit_lo(int w) { return (int)(signed char)(w & 0xff); }
DEV int digit_hi(int w) { return (w - digit_lo(w)) >> 8; }
// reference feature index (models/tdnn_generator.py:176-193) of (series, tap); series 0..4 = I, Q, |x|, |x|^2, |x|^4
DEV int feature_index(int series, int tap) {
    if (series == 0) return tap == 0 ? 0 : 20 + 2 * (tap - 1);
    if (series == 1) return tap == 0 ? 1 : 21 + 2 * (tap - 1);
    return 2 + 6 * (series - 2) + tap;
}
// FC1 slot of byte e in a lane of half h (0: lane groups 0 and 2, 1: groups 1 and 3): (series, tap), or series -1 for the
// two padding bytes.  Bytes run over ascending points: tap = 5 - point offset.
DEV void fc1_slot(int half, int e, int &series, int &tap) {
    if (e < 6) { series = half ? 3 : 0; tap = 5 - e; }
    else if (e < 12) { series = half ? 4 : 1; tap = 5 - (e - 6); }
    else if (!half) { series = 2; tap = 5 - (e - 12); }
    else if (e < 14) { series = 2; tap = 5 - (e - 12 + 4); }
    else { series = -1; tap = 0; }
}
// one byte of a weight image: lanes / bytes that meet high activation digits (act_hi) or low ones
DEV int image_byte(int w, bool act_hi, int variant) {   // variant 0: HH, 1: X, 2: LL
    const int wl = digit_lo(w), wh = digit_hi(w);
    if (variant == 0) return act_hi ? wh : 0;
    if (variant == 1) return act_hi ? wl : wh;
    return act_hi ? 0 : wl;
}
DEV int pack4(int b0, int b1, int b2, int b3) { return (b0 & 0xff) | ((b1 & 0xff) << 8) | ((b2 & 0xff) << 16) | ((unsigned)(b3 & 0xff) << 24); }

struct Q88Weights {
    const short *w1, *b1, *w2, *b2, *w3, *b3;
};

// epilogue of a result quad: combine the three accumulators, requantise, LeakyReLU (saturation rides on the pack)
DEV void requant_act4(const i32x4 &hh, const i32x4 &x, const i32x4 &ll, int &p01, int &p23) {
    int h[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // (modular arithmetic: a single term may pass 2^31, the sum is the true accumulator, which the host has bounded)
        const int acc = (int)(((unsigned)hh[r] << 16) + (((unsigned)x[r] << 8) + (unsigned)ll[r]));
        h[r] = lrelu_q_unsat(rhe_shift(acc, 15));
    }
    p01 = pack2_sat(h[0], h[1]);
    p23 = pack2_sat(h[2], h[3]);
}
// digits of two packed int16 pairs (four values) -> four high-digit bytes, four low-digit bytes
DEV void digits4(int p01, int p23, int &hi, int &lo) {
    hi = __builtin_amdgcn_perm(p23, p01, 0x07050301u);                  // the upper byte of an int16 is a >> 8
    lo = __builtin_amdgcn_perm(p23, p01, 0x06040200u) ^ 0x80808080;     // (a & 255) - 128 as a signed byte
}

// exact floor(sqrt(s)) for s <= 2^31 in 32-bit arithmetic: the float estimate is off by at most one
DEV unsigned isqrt31(unsigned s) {
    unsigned t = (unsigned)sqrtf((float)s);
    t -= (t * t > s) ? 1u : 0u;
    t += ((t + 1u) * (t + 1u) <= s) ? 1u : 0u;
    return t;
}
DEV int envelope_q88_fast(unsigned s) {      // envelope_q88 with isqrt31 (s = i^2 + q^2 <= 2^31)
    const unsigned t = isqrt31(s);
    const int q = (int)(t >> 7), rem = (int)(t & 127);
    const bool exact = t * t == s;
    const bool up = rem > 64 || (rem == 64 && (!exact || (q & 1)));
    return q + (up ? 1 : 0);
}

// Weight images and accumulator start values of the matrix-core form, per lane: IMG_N quads in the order
// A1[t][v] (6), A2[v] (3), A3 high / low digits (2), c1[t] (2), c2, c3.  Built once per set of weights
// (dpd_q88_matrix_prepare), stored [quad][lane], read by every wave with 15 coalesced 16-byte loads.
constexpr int IMG_N = 15;
__global__ __launch_bounds__(64) void k_q88_prepare(Q88Weights wt, i32x4 *__restrict__ image) {
    const int lane = threadIdx.x, j = lane & 15, kb = lane >> 4, quad = blockIdx.x;
    const bool act_hi_lane = kb < 2;          // FC1: lane groups 0, 1 meet high activation digits
    const int half = kb & 1;
    int by[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) by[e] = 0;
    i32x4 res;
    if (quad < 6) {                           // FC1 row tile t, variant v
        const int t = quad / 3, v = quad % 3;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int series, tap;
            fc1_slot(half, e, series, tap);
            const int w = series >= 0 ? wt.w1[(16 * t + j) * 30 + feature_index(series, tap)] : 0;
            by[e] = image_byte(w, act_hi_lane, v);
        }
    } else if (quad < 9) {                    // FC2, variant v
        const int v = quad - 6;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ee = e & 7;
            const int m2 = ee < 4 ? 4 * kb + ee : 16 + 4 * kb + (ee - 4);     // FC2 input neuron of this byte
            by[e] = image_byte(wt.w2[j * 32 + m2], e < 8, v);
        }
    } else if (quad < 11) {
        // FC3 for the four tiles of a chunk at once: result row 4 c + o = output o of tile c, K slot (kb, c, r) = neuron
        // 4 kb + r of tile c (what lane (kb, j) holds of tile c after FC2).  Lane (i, kb) of the weight image: row
        // i = 4 c' + o, byte 4 c + r: digit of W3[o][4 kb + r] where c == c', else 0.  High digits (quad 9), low (10).
        const int cq = j >> 2, o = j & 3;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = e >> 2, r = e & 3;
            const int w = (c == cq && o < 2) ? wt.w3[o * 16 + 4 * kb + r] : 0;
            by[e] = quad == 9 ? digit_hi(w) : digit_lo(w);
        }
    }
    if (quad < 11) {
        res = i32x4{pack4(by[0], by[1], by[2], by[3]), pack4(by[4], by[5], by[6], by[7]),
                    pack4(by[8], by[9], by[10], by[11]), pack4(by[12], by[13], by[14], by[15])};
    } else {
        // start of the LL accumulators: (bias << 15) + 128 sum_k w[n][k] for the rows 4 kb + r of a result tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int sw = 0, bias = 0;
            if (quad < 13) {                  // FC1 row tile t
                const int n = 16 * (quad - 11) + 4 * kb + r;
                for (int k = 0; k < 30; ++k) sw += wt.w1[n * 30 + k];
                bias = wt.b1[n];
            } else if (quad == 13) {
                const int n = 4 * kb + r;
                for (int k = 0; k < 32; ++k) sw += wt.w2[n * 32 + k];
                bias = wt.b2[n];
            } else if (r < 2) {               // FC3 result rows 4 c + r: output r of every tile
                for (int k = 0; k < 16; ++k) sw += wt.w3[r * 16 + k];
                bias = wt.b3[r];
            }
            res[r] = (bias << 15) + 128 * sw;
        }
    }
    image[quad * 64 + lane] = res;
}

__global__ __launch_bounds__(MW * 64, Q88_MFMA_WAVES) void k_q88_mfma(const int *__restrict__ iq, int64_t S, int64_t L, unsigned cpr,
                                                         unsigned n_chunks, unsigned unit, const i32x4 *__restrict__ image,
                                                         const short *__restrict__ lut, int *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char planes[MW][WAVE_LDS];
    __shared__ int s_lut[256];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, kb = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    s_lut[tid] = lut[tid];
    unsigned char *pl = planes[wave];
    for (int k = lane; k < WAVE_LDS / 4; k += 64) reinterpret_cast<int *>(pl)[k] = 0;
    __syncthreads();
    const int half = kb & 1;
    i32x4 A1[2][3], A2[3], A3h, A3l, c1[2], c2, c3;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 3; ++v) A1[t][v] = image[(3 * t + v) * 64 + lane];
#pragma unroll
    for (int v = 0; v < 3; ++v) A2[v] = image[(6 + v) * 64 + lane];
    A3h = image[9 * 64 + lane];
    A3l = image[10 * 64 + lane];
    c1[0] = image[11 * 64 + lane];
    c1[1] = image[12 * 64 + lane];
    c2 = image[13 * 64 + lane];
    c3 = image[14 * 64 + lane];
    // FC1 operand gather: three runs of consecutive bytes, the same instruction sequence in every lane
    const int digit = kb >> 1;
    const int pA = (digit * 5 + (half ? 3 : 0)) * PLANE + j;
    const int pB = (digit * 5 + (half ? 4 : 1)) * PLANE + j;
    const int pC = (digit * 5 + 2) * PLANE + j + 4 * half;
    const unsigned sh = j & 3;                 // the same for the three runs: plane bases and the half offset are multiples of 4
    const unsigned char *qA = pl + (pA & ~3), *qB = pl + (pB & ~3), *qC = pl + (pC & ~3);
    const i32x4 zero4 = {0, 0, 0, 0};

    // one time point of the chunk: its five Q8.8 features as digits into the byte planes
    auto commit = [&](int v, int u) {
        const int i = (short)(v & 0xffff), q = v >> 16;
        const unsigned ss = (unsigned)(i * i) + (unsigned)(q * q);
        int f[5];
        f[0] = sat16(rhe_shift(i, 7));
        f[1] = sat16(rhe_shift(q, 7));
        f[2] = envelope_q88_fast(ss);
        f[3] = (int)((ss + ((1u << 21) - 1) + ((ss >> 22) & 1u)) >> 22);       // ss <= 2^31: the sum fits 32 bits
        f[4] = (int)(((unsigned long long)ss * ss + (1ull << 51)) >> 52);
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            pl[s5 * PLANE + u] = (unsigned char)(f[s5] >> 8);
            pl[(5 + s5) * PLANE + u] = (unsigned char)((f[s5] & 0xff) ^ 0x80);
        }
    };

    // A wave walks runs of `unit` consecutive chunks (runs dealt round robin over the waves; scalar arithmetic).  Inside a
    // run of one row the five history points of a chunk are the last five of its predecessor: their digits are moved
    // inside LDS; only the first chunk of a run (or of a row) derives them.  The next chunk's points are loaded one
    // chunk ahead.
    const unsigned n_waves = gridDim.x * MW;
    unsigned chunk = (blockIdx.x * MW + wave) * unit, k_in_unit = 0;
    unsigned row = chunk / cpr, pos = chunk - row * cpr;
    bool fresh = true;
    int pre = 0, pre_h = 0;
    auto fetch = [&](unsigned r, unsigned ps, bool fr) {
        const int *p = iq + (int64_t)r * S + (int64_t)ps * 64;
        const int64_t tb = (int64_t)ps * 64;
        pre = (tb + 5 + lane < S) ? p[5 + lane] : 0;
        if (fr) pre_h = (lane < 5 && tb + lane < S) ? p[lane] : 0;
    };
    if (chunk < n_chunks) fetch(row, pos, true);
    while (chunk < n_chunks) {
        if (fresh) {
            if (lane < 5) commit(pre_h, lane);
        } else if (lane < 50) {
            const int plane = lane / 5, off = lane - 5 * plane;
            pl[plane * PLANE + off] = pl[plane * PLANE + 64 + off];
        }
        commit(pre, 5 + lane);
        const int64_t t0 = (int64_t)pos * 64, row_cur = row;
        if (++k_in_unit < unit) {
            chunk += 1;
            fresh = ++pos == cpr;
            if (fresh) {
                pos = 0;
                ++row;
            }
        } else {
            k_in_unit = 0;
            chunk += 1 + (n_waves - 1) * unit;
            row = chunk / cpr;
            pos = chunk - row * cpr;
            fresh = true;
        }
        if (chunk < n_chunks) fetch(row, pos, fresh);
        __builtin_amdgcn_wave_barrier();
        // FC2 results of the four tiles, as digits.  All four tiles run for every chunk: past the end of the row their
        // points are zeros (or what an earlier chunk left in the planes) and their outputs are not stored, which is
        // cheaper than a guard whose skipped path needs these eight registers zeroed in every chunk.
        int x3h[4], x3l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // ---- FC1 operand
            const int *a = reinterpret_cast<const int *>(qA + 16 * c), *b = reinterpret_cast<const int *>(qB + 16 * c);
            const int *cc = reinterpret_cast<const int *>(qC + 16 * c);
            const int a0 = a[0], a1 = a[1], a2 = a[2], b0 = b[0], b1 = b[1], b2 = b[2], c0 = cc[0], c1w = cc[1];
            const int A03 = __builtin_amdgcn_alignbyte(a1, a0, sh), A47 = __builtin_amdgcn_alignbyte(a2, a1, sh);
            const int B03 = __builtin_amdgcn_alignbyte(b1, b0, sh), B47 = __builtin_amdgcn_alignbyte(b2, b1, sh);
            const int C03 = __builtin_amdgcn_alignbyte(c1w, c0, sh);
            const i32x4 X1 = {A03, (int)__builtin_amdgcn_perm(B03, A47, 0x05040100u), (int)__builtin_amdgcn_perm(B47, B03, 0x05040302u), C03};
            // ---- FC1: two row tiles x (HH, X, LL)
            int p1[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const i32x4 hh = mfma_i8(A1[t][0], X1, zero4), xx = mfma_i8(A1[t][1], X1, zero4), ll = mfma_i8(A1[t][2], X1, c1[t]);
                requant_act4(hh, xx, ll, p1[2 * t], p1[2 * t + 1]);
            }
            int h0, l0, h1, l1;
            digits4(p1[0], p1[1], h0, l0);           // bytes 0..3 / 8..11: neurons 4 kb + r of row tile 0
            digits4(p1[2], p1[3], h1, l1);           // bytes 4..7 / 12..15: row tile 1
            const i32x4 X2 = {h0, h1, l0, l1};
            // ---- FC2
            int p20, p21;
            {
                const i32x4 hh = mfma_i8(A2[0], X2, zero4), xx = mfma_i8(A2[1], X2, zero4), ll = mfma_i8(A2[2], X2, c2);
                requant_act4(hh, xx, ll, p20, p21);
            }
            digits4(p20, p21, x3h[c], x3l[c]);
#ifdef Q88_TILE_FENCE      // (one tile at a time keeps the register count at 104 instead of 120; without the fence 1 % faster at four
                           // waves.  Two tiles half a stage apart in one stream, as in the generator kernels: 2-4 % slower.)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        // ---- FC3 of the whole chunk and the output table: lane (c, j) gets outputs 0, 1 of sample 16 c + j = its own lane index
        {
            const i32x4 Xh = {x3h[0], x3h[1], x3h[2], x3h[3]}, Xl = {x3l[0], x3l[1], x3l[2], x3l[3]};
            const i32x4 hh = mfma_i8(A3h, Xh, zero4);
            i32x4 xx = mfma_i8(A3l, Xh, zero4);
            xx = mfma_i8(A3h, Xl, xx);
            const i32x4 ll = mfma_i8(A3l, Xl, c3);
            int o[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int acc = (int)(((unsigned)hh[r] << 16) + (((unsigned)xx[r] << 8) + (unsigned)ll[r]));
                const int z = sat16(rhe_shift(acc, 15));
                int idx = rhe_shift((z + 1024) * 255, 11);
                idx = min(max(idx, 0), 255);
                o[r] = s_lut[idx];
            }
            const int64_t t = t0 + lane;
            if (t < L) out[row_cur * L + t] = pack2(o[0], o[1]);
        }
    }
}

}  // namespace

extern "C" int dpd_q88_infer(const int16_t *iq, int64_t B, int64_t S, const int16_t *w1, const int16_t *b1,
                             const int16_t *w2, const int16_t *b2, const int16_t *w3, const int16_t *b3,
                             const int16_t *lut, int acc64, int16_t *out, void *stream) {
    DPD_REQUIRE(((iq && out) || B == 0) && w1 && b1 && w2 && b2 && w3 && b3 && lut, DPD_ERR_INVALID_ARG, "q88_infer: null pointer");
    DPD_REQUIRE(S > DPD_GEN_MEMORY, DPD_ERR_SHAPE, "Sequence length %lld must be > memory depth %d", (long long)S, DPD_GEN_MEMORY);
    auto aligned4 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 3) == 0; };
    DPD_REQUIRE(aligned4(iq) && aligned4(out) && aligned4(w1) && aligned4(b1) && aligned4(w2) && aligned4(b2) &&
                    aligned4(w3) && aligned4(b3),
                DPD_ERR_INVALID_ARG, "q88_infer: iq, out, weights and biases must be 4-byte aligned");
    if (B <= 0) return DPD_OK;
    const int64_t L = S - DPD_GEN_MEMORY, segs_per_row = (L + SEG - 1) / SEG, n_segs = B * segs_per_row;
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)(n_segs < 256 * 8 ? n_segs : 256 * 8)), block(TPB);
    const int *iqp = reinterpret_cast<const int *>(iq);
    int *op = reinterpret_cast<int *>(out);
    auto dw = [](const int16_t *q) { return reinterpret_cast<const int *>(q); };
    if (acc64)
        hipLaunchKernelGGL((k_q88<true>), grid, block, 0, st, iqp, S, L, segs_per_row, n_segs, dw(w1), dw(b1), dw(w2),
                           dw(b2), dw(w3), dw(b3), lut, op);
    else
        hipLaunchKernelGGL((k_q88<false>), grid, block, 0, st, iqp, S, L, segs_per_row, n_segs, dw(w1), dw(b1), dw(w2),
                           dw(b2), dw(w3), dw(b3), lut, op);
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

// ---- matrix-core form: prepared weight image + inference ------------------------------------------------
extern "C" size_t dpd_q88_matrix_image_bytes(void) { return (size_t)IMG_N * 64 * sizeof(i32x4); }

extern "C" int dpd_q88_matrix_prepare(const int16_t *w1, const int16_t *b1, const int16_t *w2, const int16_t *b2,
                                      const int16_t *w3, const int16_t *b3, void *image, void *stream) {
    DPD_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && image, DPD_ERR_INVALID_ARG, "q88_matrix_prepare: null pointer");
    DPD_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, DPD_ERR_INVALID_ARG, "q88_matrix_prepare: image must be 16-byte aligned");
    const Q88Weights wt{w1, b1, w2, b2, w3, b3};
    hipLaunchKernelGGL(k_q88_prepare, dim3(IMG_N), dim3(64), 0, as_stream(stream), wt, reinterpret_cast<i32x4 *>(image));
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

extern "C" int dpd_q88_matrix_infer(const int16_t *iq, int64_t B, int64_t S, const void *image, const int16_t *lut,
                                    int16_t *out, void *stream) {
    DPD_REQUIRE(((iq && out) || B == 0) && image && lut, DPD_ERR_INVALID_ARG, "q88_matrix_infer: null pointer");
    DPD_REQUIRE(S > DPD_GEN_MEMORY, DPD_ERR_SHAPE, "Sequence length %lld must be > memory depth %d", (long long)S, DPD_GEN_MEMORY);
    DPD_REQUIRE((reinterpret_cast<uintptr_t>(iq) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(image) & 15) == 0,
                DPD_ERR_INVALID_ARG, "q88_matrix_infer: iq and out must be 4-byte aligned, the image 16-byte aligned");
    if (B <= 0) return DPD_OK;
    const int64_t L = S - DPD_GEN_MEMORY, cpr = (L + 63) / 64, n_chunks = B * cpr;
    DPD_REQUIRE(n_chunks < ((int64_t)1 << 31) - 65536, DPD_ERR_SHAPE, "q88_matrix_infer: %lld chunks exceed the 32-bit chunk counter", (long long)n_chunks);
    int64_t blocks = (n_chunks + MW - 1) / MW;
    if (blocks > 256 * Q88_MFMA_WAVES) blocks = 256 * Q88_MFMA_WAVES;
    // consecutive chunks per run: as many as keep every wave busy, at most a row's worth of 16
    int64_t unit = n_chunks / (blocks * MW);
    unit = unit < 1 ? 1 : unit > 16 ? 16 : unit;
    hipLaunchKernelGGL(k_q88_mfma, dim3((unsigned)blocks), dim3(MW * 64), 0, as_stream(stream), reinterpret_cast<const int *>(iq), S, L,
                       (unsigned)cpr, (unsigned)n_chunks, (unsigned)unit, reinterpret_cast<const i32x4 *>(image), lut,
                       reinterpret_cast<int *>(out));
    DPD_LAUNCH_CHECK();
    return DPD_OK;
}

// 1 if every weight splits into two signed 8-bit digits (w <= 32639: the high digit of a larger weight would be 128).
extern "C" int dpd_q88_digits_fit(const int16_t *h_w1, const int16_t *h_w2, const int16_t *h_w3) {
    if (!h_w1 || !h_w2 || !h_w3) return 0;
    for (int k = 0; k < 32 * 30; ++k) if (h_w1[k] > 32639) return 0;
    for (int k = 0; k < 16 * 32; ++k) if (h_w2[k] > 32639) return 0;
    for (int k = 0; k < 2 * 16; ++k) if (h_w3[k] > 32639) return 0;
    return 1;
}

// Interval propagation over the three layers: worst-case |accumulator| from the weights alone.
extern "C" int dpd_q88_int32_safe(const int16_t *h_w1, const int16_t *h_b1, const int16_t *h_w2, const int16_t *h_b2,
                                  const int16_t *h_w3, const int16_t *h_b3) {
    if (!h_w1 || !h_b1 || !h_w2 || !h_b2 || !h_w3 || !h_b3) return 0;
    const long long LIM = (1ll << 31) - (1ll << 15);   // head-room for the rounding constant
    // feature magnitude bounds in Q8.8: |I|,|Q| <= 256, |x| <= 363, |x|^2 <= 512, |x|^4 <= 1024
    long long fmax[30];
    for (int k = 0; k < 30; ++k) fmax[k] = 256;
    for (int g = 0; g < 6; ++g) { fmax[2 + g] = 363; fmax[8 + g] = 512; fmax[14 + g] = 1024; }
    auto absll = [](long long v) { return v < 0 ? -v : v; };
    long long h1[32], h2[16];
    for (int j = 0; j < 32; ++j) {
        long long a = absll(h_b1[j]) << 15;
        for (int k = 0; k < 30; ++k) a += absll(h_w1[j * 30 + k]) * fmax[k];
        if (a >= LIM) return 0;
        long long z = (a >> 15) + 1;
        h1[j] = z > 32768 ? 32768 : z;
    }
    for (int j = 0; j < 16; ++j) {
        long long a = absll(h_b2[j]) << 15;
        for (int k = 0; k < 32; ++k) a += absll(h_w2[j * 32 + k]) * h1[k];
        if (a >= LIM) return 0;
        long long z = (a >> 15) + 1;
        h2[j] = z > 32768 ? 32768 : z;
    }
    for (int j = 0; j < 2; ++j) {
        long long a = absll(h_b3[j]) << 15;
        for (int k = 0; k < 16; ++k) a += absll(h_w3[j * 16 + k]) * h2[k];
        if (a >= LIM) return 0;
    }
    return 1;
}
