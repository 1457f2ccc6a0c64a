// Shared device helpers and host-side error plumbing for libdpd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/dpd_hip.h"

// ---- host side -------------------------------------------------------------------------
void dpd_set_error(const char *fmt, ...);

#define DPD_REQUIRE(cond, code, ...)        \
    do {                                    \
        if (!(cond)) {                      \
            dpd_set_error(__VA_ARGS__);     \
            return (code);                  \
        }                                   \
    } while (0)

#define DPD_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            dpd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return DPD_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)

#define DPD_LAUNCH_CHECK() DPD_HIP(hipGetLastError())

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Set-up that must happen once per GPU (function attributes belong to the device that is current when they are
// set): first_use() is true the first time it is called with a given device current.  Two threads that race on
// the first use of a device both see true; the set-up calls are idempotent.
#include <atomic>
struct PerDeviceOnce {
    std::atomic<unsigned long long> seen{0};
    bool first_use(int *device) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
            *device = -1;
            return true;              // unknown device: do the set-up every time
        }
        *device = dev;
        return (seen.load(std::memory_order_acquire) >> dev & 1ull) == 0;
    }
    void mark(int device) {
        if (device >= 0) seen.fetch_or(1ull << device, std::memory_order_release);
    }
};

// Blocks for a grid-stride kernel: enough to fill 256 CUs several times over, never more than the work.
static inline int grid_for(int64_t items, int per_block, int max_blocks = 256 * 8) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

// ---- generator flat layout --------------------------------------------------------------
constexpr int G_W1 = 0, G_B1 = 960, G_W2 = 992, G_B2 = 1504, G_W3 = 1520, G_B3 = 1552;
constexpr int G_NP = DPD_GEN_NPARAM;
constexpr int G_SLAB = 1560;  // one partial-gradient row: 1554 gradients, [1554] = loss partial, padding

// critic flat layout
constexpr int C_W1 = 0, C_B1 = 256, C_W2 = 320, C_B2 = 2368, C_W3 = 2400, C_B3 = 2912, C_W4 = 2928, C_B4 = 2944;
constexpr int C_NP = DPD_CRITIC_NPARAM;
constexpr int C_SLAB = 2952;  // 2945 gradients + padding

// ---- device helpers ---------------------------------------------------------------------
#define DEV __device__ __forceinline__

DEV float scale_of(float amax, float qmax = 32767.0f) { return fmaxf(amax, 1e-8f) / qmax; }

// A 16-bit quantiser scale with its correctly rounded reciprocal (one double division per kernel).
struct QS {
    float s, inv;
};
DEV QS qs_of(float amax) {
    QS q;
    q.s = scale_of(amax);
    q.inv = (float)(1.0 / (double)q.s);
    return q;
}
DEV QS qs_one() { return QS{1.f, 1.f}; }

// quantise-dequantise with a given scale: clamp(rint(v / s), -2^15, 2^15 - 1) * s, round-half-even: the
// arithmetic of the reference's StraightThroughQuantize.  v / s is formed as q0 = v * inv,
// q = fma(fma(-q0, s, v), inv, q0): with inv the correctly rounded reciprocal this is the correctly
// rounded quotient (Markstein), i.e. bit-identical to the true division at a third of its cost.
DEV float fq16(float v, QS k) {
    const float q0 = v * k.inv;
    float q = rintf(__builtin_fmaf(__builtin_fmaf(-q0, k.s, v), k.inv, q0));
    q = fminf(fmaxf(q, -32768.0f), 32767.0f);
    return q * k.s;
}
DEV float fq16(float v, float s) {   // one-off uses: true division
    float q = rintf(v / s);
    q = fminf(fmaxf(q, -32768.0f), 32767.0f);
    return q * s;
}

DEV float lrelu(float v, float slope) { return v > 0.0f ? v : v * slope; }
DEV float lrelu_mask(float z, float slope) { return z > 0.0f ? 1.0f : slope; }

DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
DEV double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// atomic max for non-negative floats (bit pattern order == value order)
DEV void atomic_max_nonneg(float *addr, float v) {
    atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}
