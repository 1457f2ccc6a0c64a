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
