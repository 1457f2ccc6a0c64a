// C ABI of the generator path: argument checks, kernel-family dispatch, multi-pass QAT driver.
#include <mutex>
#include "gen_internal.h"

int gen_launch_prepare(const float *, const float *, const float *, const float *, const float *, const float *, int,
                       float *, float *, hipStream_t);
int gen_launch_taps(const float *, int64_t, int64_t, float *, hipStream_t);
int gen_launch_tap_adjoint(const float *, int64_t, int64_t, const float *, float *, hipStream_t);
int gen_launch_quantize_output(float *, int64_t, const float *, hipStream_t);

namespace {

// what the generator backward uses of a workspace: 1024 slab rows (neither kernel family launches more
// workgroups) and, right behind them, the chunk counter of the MFMA family
constexpr size_t GEN_WS_BYTES = ((size_t)1024 * G_SLAB + 64) * sizeof(float);
constexpr int64_t MFMA_MIN_SAMPLES = 4096;  // below this the VALU kernels win on launch geometry

int make_problem(GenProblem &p, const float *x, int64_t B, int64_t S, int pre, const float *wq, float slope, int qat,
                 float *absmax) {
    DPD_REQUIRE((x || B == 0) && wq, DPD_ERR_INVALID_ARG, "generator: null input or weight pointer");
    DPD_REQUIRE(B >= 0, DPD_ERR_INVALID_ARG, "generator: negative batch %lld", (long long)B);
    if (!pre) DPD_REQUIRE(S > DPD_GEN_MEMORY, DPD_ERR_SHAPE, "Sequence length %lld must be > memory depth %d", (long long)S, DPD_GEN_MEMORY);
    DPD_REQUIRE(!qat || absmax, DPD_ERR_INVALID_ARG, "generator: QAT needs the absmax buffer");
    p.x = x; p.B = B; p.S = pre ? 0 : S; p.L = pre ? 1 : S - DPD_GEN_MEMORY; p.N = B * p.L;
    p.pre = pre != 0; p.qat = qat != 0; p.wq = wq; p.slope = slope; p.absmax = absmax;
    return DPD_OK;
}

bool use_mfma(const GenProblem &p, int impl) {
    if (p.pre) return false;
    if (impl == DPD_IMPL_MFMA) return true;
    if (impl == DPD_IMPL_VALU) return false;
    return p.N >= MFMA_MIN_SAMPLES;
}

int forward_dispatch(const GenProblem &p, int depth, float *out, int impl, hipStream_t st) {
    if (p.N == 0) return DPD_OK;
    return use_mfma(p, impl) ? gen_mfma_forward(p, depth, out, st) : gen_valu_forward(p, depth, out, st);
}

}  // namespace

extern "C" size_t dpd_workspace_bytes(int64_t n_samples) {
    // partial-gradient slabs (at most 2048 rows of max(G_SLAB, C_SLAB) floats) plus, for the gradient
    // penalty, 28 bytes per sample (input gradient, mask bits, row coefficient)
    if (n_samples < 0) n_samples = 0;
    return (size_t)2048 * C_SLAB * sizeof(float) + (size_t)(1 << 20) + (size_t)n_samples * 28;
}

extern "C" int dpd_tap_assembly(const float *iq, int64_t B, int64_t S, float *feat, void *stream) {
    DPD_REQUIRE((iq && feat) || B == 0, DPD_ERR_INVALID_ARG, "tap_assembly: null pointer");
    DPD_REQUIRE(S > DPD_GEN_MEMORY, DPD_ERR_SHAPE, "Sequence length %lld must be > memory depth %d", (long long)S, DPD_GEN_MEMORY);
    if (B <= 0) return DPD_OK;
    return gen_launch_taps(iq, B, S, feat, as_stream(stream));
}

extern "C" int dpd_tap_assembly_backward(const float *iq, int64_t B, int64_t S, const float *gfeat, float *dx, void *stream) {
    DPD_REQUIRE((iq && gfeat && dx) || B == 0, DPD_ERR_INVALID_ARG, "tap_assembly_backward: null pointer");
    DPD_REQUIRE(S > DPD_GEN_MEMORY, DPD_ERR_SHAPE, "Sequence length %lld must be > memory depth %d", (long long)S, DPD_GEN_MEMORY);
    if (B <= 0) return DPD_OK;
    return gen_launch_tap_adjoint(iq, B, S, gfeat, dx, as_stream(stream));
}

extern "C" int dpd_gen_prepare_weights(const float *w1, const float *b1, const float *w2, const float *b2,
                                       const float *w3, const float *b3, int qat, float *wq, float *absmax,
                                       void *stream) {
    DPD_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && wq && absmax, DPD_ERR_INVALID_ARG, "prepare_weights: null pointer");
    return gen_launch_prepare(w1, b1, w2, b2, w3, b3, qat, wq, absmax, as_stream(stream));
}

extern "C" int dpd_gen_absmax_pass(int stage, const float *x, int64_t B, int64_t S, int pre_assembled,
                                   const float *wq, float slope, float *absmax, float *out, int impl, void *stream) {
    DPD_REQUIRE(stage >= 0 && stage <= 3, DPD_ERR_INVALID_ARG, "absmax_pass: stage %d not in 0..3", stage);
    GenProblem p;
    int rc = make_problem(p, x, B, S, pre_assembled, wq, slope, 1, absmax);
    if (rc) return rc;
    DPD_REQUIRE(stage < 3 || out || use_mfma(p, impl), DPD_ERR_INVALID_ARG,
                "absmax_pass: stage 3 writes the output (optional only for IQ input of >= 4096 samples), out is null");
    hipStream_t st = as_stream(stream);
    DPD_HIP(hipMemsetAsync(absmax + stage, 0, sizeof(float), st));
    return forward_dispatch(p, stage, out, impl, st);
}

extern "C" int dpd_gen_uses_mfma(int64_t B, int64_t S, int pre_assembled, int impl) {
    GenProblem p;
    p.pre = pre_assembled != 0;
    p.N = p.pre ? B : B * (S - DPD_GEN_MEMORY);
    return use_mfma(p, impl) ? 1 : 0;
}

extern "C" int dpd_gen_quantize_output(float *out, int64_t n, const float *absmax, void *stream) {
    DPD_REQUIRE((out || n == 0) && absmax && n >= 0, DPD_ERR_INVALID_ARG, "quantize_output: bad argument");
    if (n == 0) return DPD_OK;
    return gen_launch_quantize_output(out, n, absmax, as_stream(stream));
}

extern "C" int dpd_gen_forward(const float *x, int64_t B, int64_t S, int pre_assembled, const float *wq, float slope,
                               int qat, float *absmax, float *out, int impl, void *stream) {
    GenProblem p;
    int rc = make_problem(p, x, B, S, pre_assembled, wq, slope, qat, absmax);
    if (rc) return rc;
    DPD_REQUIRE(out || B == 0, DPD_ERR_INVALID_ARG, "gen_forward: out is null");
    hipStream_t st = as_stream(stream);
    if (!p.qat) return forward_dispatch(p, 4, out, impl, st);
    DPD_HIP(hipMemsetAsync(absmax, 0, 4 * sizeof(float), st));
    for (int stage = 0; stage < 4; ++stage) {
        rc = forward_dispatch(p, stage, out, impl, st);
        if (rc) return rc;
    }
    if (p.N == 0) return DPD_OK;
    return gen_launch_quantize_output(out, p.N * 2, absmax, st);
}

static int backward_common(const GenProblem &p, const float *grad_out, const float *target, float *grads, float *loss,
                           float *grad_feat, void *ws, size_t ws_bytes, int impl, hipStream_t st,
                           const GenPaChain *chain = nullptr) {
    DPD_REQUIRE(grads, DPD_ERR_INVALID_ARG, "gen_backward: grads is null");
    DPD_REQUIRE(ws && ws_bytes >= GEN_WS_BYTES, DPD_ERR_WORKSPACE,
                "gen_backward: workspace of %zu bytes needed, %zu given", GEN_WS_BYTES, ws_bytes);
    if (p.N == 0) {
        DPD_HIP(hipMemsetAsync(grads, 0, G_NP * sizeof(float), st));
        if (loss) DPD_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
        return DPD_OK;
    }
    float *slab = reinterpret_cast<float *>(ws);
    int n_partials = 0, rc;
    if (use_mfma(p, impl)) {
        DPD_REQUIRE(grad_feat == nullptr, DPD_ERR_INVALID_ARG, "gen_backward: grad_feat needs pre-assembled input");
        rc = gen_mfma_backward(p, grad_out, target, slab, &n_partials, st, chain);
    } else {
        DPD_REQUIRE(chain == nullptr, DPD_ERR_INVALID_ARG, "gen_backward: the fused chain runs on the MFMA kernels only");
        rc = gen_valu_backward(p, grad_out, target, grad_feat, slab, &n_partials, st);
    }
    if (rc) return rc;
    // the slab's loss column holds sums of squared differences: mean over the elements of the compared tensor
    const double n_elem = 2.0 * (double)p.B * (double)(chain ? p.L - (chain->M - 1) : p.L);
    return gen_reduce_slab(slab, n_partials, grads, loss, (float)(1.0 / n_elem), st);
}

extern "C" int dpd_gen_backward(const float *x, int64_t B, int64_t S, int pre_assembled, const float *wq, float slope,
                                int qat, const float *absmax, const float *grad_out, float *grads, float *grad_feat,
                                void *ws, size_t ws_bytes, int impl, void *stream) {
    GenProblem p;
    int rc = make_problem(p, x, B, S, pre_assembled, wq, slope, qat, const_cast<float *>(absmax));
    if (rc) return rc;
    DPD_REQUIRE(grad_out || B == 0, DPD_ERR_INVALID_ARG, "gen_backward: grad_out is null");
    DPD_REQUIRE(!grad_feat || p.pre, DPD_ERR_INVALID_ARG, "gen_backward: grad_feat needs pre-assembled input");
    return backward_common(p, grad_out, nullptr, grads, nullptr, grad_feat, ws, ws_bytes, impl, as_stream(stream));
}

extern "C" int dpd_gen_mse_step(const float *iq, int64_t B, int64_t S, const float *target, const float *wq,
                                float slope, int qat, const float *absmax, float *grads, float *loss, void *ws,
                                size_t ws_bytes, int impl, void *stream) {
    GenProblem p;
    int rc = make_problem(p, iq, B, S, 0, wq, slope, qat, const_cast<float *>(absmax));
    if (rc) return rc;
    DPD_REQUIRE((target || B == 0) && loss, DPD_ERR_INVALID_ARG, "gen_mse_step: null target or loss");
    return backward_common(p, nullptr, target, grads, loss, nullptr, ws, ws_bytes, impl, as_stream(stream));
}

// ---- chain generator -> PA twin with its backward -------------------------------------------------
int pa_launch_mse_grad(const float *y, const float *t, int64_t n, float *gy, double *part, float *loss, hipStream_t st);

namespace {
// One kernel when the batch is large enough for the MFMA family and the PA has the reference's shape
// (5 taps, at most 4 odd orders).  Otherwise the same result from the separate kernels, with the generator
// output, the PA output / its gradient and the gradient w.r.t. the generator output in the tail of the workspace.
int chain_backward(const GenProblem &p, const float *cre, const float *cim, int K, int M, const float *gy,
                   const float *target, float *grads, float *loss, void *ws, size_t ws_bytes, hipStream_t st) {
    DPD_REQUIRE(cre && cim, DPD_ERR_INVALID_ARG, "gen_pa: null coefficient pointer");
    DPD_REQUIRE(K >= 1 && K <= 8 && M >= 1 && M <= 16, DPD_ERR_INVALID_ARG, "gen_pa: K in 1..8, M in 1..16");
    DPD_REQUIRE(p.L >= M, DPD_ERR_SHAPE, "Sequence length %lld must be >= memory depth %d", (long long)p.L, M);
    DPD_REQUIRE(grads, DPD_ERR_INVALID_ARG, "gen_pa: grads is null");
    if (p.N == 0) {
        DPD_HIP(hipMemsetAsync(grads, 0, G_NP * sizeof(float), st));
        if (loss) DPD_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
        return DPD_OK;
    }
    const GenPaChain chain{cre, cim, K, M};
    if (use_mfma(p, DPD_IMPL_AUTO) && M == 5 && K <= 4)
        return backward_common(p, gy, target, grads, loss, nullptr, ws, ws_bytes, DPD_IMPL_MFMA, st, &chain);
    const int64_t OL = p.L - M + 1, NY = p.B * OL;
    DPD_REQUIRE(ws && ws_bytes >= dpd_workspace_bytes(p.N), DPD_ERR_WORKSPACE,
                "gen_pa: this shape runs on the separate kernels and needs a workspace of %zu bytes, %zu given",
                dpd_workspace_bytes(p.N), ws_bytes);
    char *tail = static_cast<char *>(ws) + dpd_workspace_bytes(0);
    double *part = reinterpret_cast<double *>(tail - 8192);                // inside the spare megabyte behind the slabs
    float *dpd = reinterpret_cast<float *>(tail);                           // [B, L, 2]
    float *yb = dpd + 2 * p.N;                                              // [B, OL, 2]: PA output, then its gradient
    float *gx = yb + 2 * p.N;                                               // [B, L, 2]
    int rc = forward_dispatch(p, p.qat ? 3 : 4, dpd, DPD_IMPL_AUTO, st);    // (pass 3 rewrites absmax[3] with the value it holds)
    if (!rc && p.qat) rc = gen_launch_quantize_output(dpd, p.N * 2, p.absmax, st);
    if (!rc) rc = dpd_pa_forward(dpd, p.B, p.L, cre, cim, K, M, yb, st);
    const float *gyp = gy;
    if (!rc && target) {
        rc = pa_launch_mse_grad(yb, target, 2 * NY, yb, part, loss, st);
        gyp = yb;
    }
    if (!rc) rc = dpd_pa_backward(dpd, p.B, p.L, cre, cim, K, M, gyp, gx, st);
    if (rc) return rc;
    return backward_common(p, gx, nullptr, grads, nullptr, nullptr, ws, ws_bytes, DPD_IMPL_AUTO, st);
}
}  // namespace

extern "C" int dpd_gen_pa_backward(const float *iq, int64_t B, int64_t S, const float *wq, float slope, int qat,
                                   const float *absmax, const float *coef_re, const float *coef_im, int K, int M,
                                   const float *gy, float *grads, void *ws, size_t ws_bytes, void *stream) {
    GenProblem p;
    int rc = make_problem(p, iq, B, S, 0, wq, slope, qat, const_cast<float *>(absmax));
    if (rc) return rc;
    DPD_REQUIRE(gy || B == 0, DPD_ERR_INVALID_ARG, "gen_pa_backward: gy is null");
    return chain_backward(p, coef_re, coef_im, K, M, gy, nullptr, grads, nullptr, ws, ws_bytes, as_stream(stream));
}

extern "C" int dpd_gen_pa_mse_step(const float *iq, int64_t B, int64_t S, const float *target, const float *wq,
                                   float slope, int qat, const float *absmax, const float *coef_re,
                                   const float *coef_im, int K, int M, float *grads, float *loss, void *ws,
                                   size_t ws_bytes, void *stream) {
    GenProblem p;
    int rc = make_problem(p, iq, B, S, 0, wq, slope, qat, const_cast<float *>(absmax));
    if (rc) return rc;
    DPD_REQUIRE((target || B == 0) && loss, DPD_ERR_INVALID_ARG, "gen_pa_mse_step: null target or loss");
    return chain_backward(p, coef_re, coef_im, K, M, nullptr, target, grads, loss, ws, ws_bytes, as_stream(stream));
}

// ---- Stage-1 step for callers that cannot allocate ---------------------------------------------
// Same kernels as dpd_gen_mse_step; the partial-gradient slab and the chunk counter live in a
// file-scope device array instead of a caller-provided workspace.  Every GPU holds its own copy of the
// array at its own address: the address is looked up once per device and kept.
constexpr size_t NOALLOC_FLOATS = (size_t)1024 * G_SLAB + 64;        // 1024 slab rows + the chunk counter behind them
static_assert(NOALLOC_FLOATS * sizeof(float) == GEN_WS_BYTES, "scratch covers exactly what the backward uses");
static __device__ __align__(256) float scratch[NOALLOC_FLOATS];      // 6,390,016 bytes per GPU, held until the process ends

namespace {
constexpr int MAX_DEVICES = 64;
float *g_scratch_ptr[MAX_DEVICES];     // device address of `scratch` on each GPU (0 = not looked up yet)
std::mutex g_scratch_mutex;

int scratch_of_current_device(float **ptr) {
    int dev = -1;
    DPD_HIP(hipGetDevice(&dev));
    DPD_REQUIRE(dev >= 0 && dev < MAX_DEVICES, DPD_ERR_INVALID_ARG, "gen_mse_step_noalloc: device index %d not in 0..%d", dev, MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void *p = nullptr;
        DPD_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        DPD_REQUIRE(p != nullptr, DPD_ERR_HIP, "gen_mse_step_noalloc: the scratch array has no address on device %d", dev);
        g_scratch_ptr[dev] = static_cast<float *>(p);
    }
    *ptr = g_scratch_ptr[dev];
    return DPD_OK;
}
}  // namespace

extern "C" int dpd_gen_mse_step_noalloc(const float *iq, int64_t B, int64_t S, const float *target, const float *wq,
                                        float slope, int qat, const float *absmax, float *grads, float *loss,
                                        int impl, void *stream) {
    GenProblem p;
    int rc = make_problem(p, iq, B, S, 0, wq, slope, qat, const_cast<float *>(absmax));
    if (rc) return rc;
    DPD_REQUIRE((target || B == 0) && loss, DPD_ERR_INVALID_ARG, "gen_mse_step_noalloc: null target or loss");
    DPD_REQUIRE(p.N <= DPD_NOALLOC_MAX_SAMPLES, DPD_ERR_WORKSPACE,
                "gen_mse_step_noalloc: %lld samples exceed the maximum of %lld per call; split the batch or use dpd_gen_mse_step",
                (long long)p.N, (long long)DPD_NOALLOC_MAX_SAMPLES);
    float *ptr = nullptr;
    rc = scratch_of_current_device(&ptr);
    if (rc) return rc;
    // the slab rows a launch reads are all written by that launch, and the chunk counter is cleared by the
    // backward launcher on `stream`: nothing of an earlier call's contents is used
    return backward_common(p, nullptr, target, grads, loss, nullptr, ptr, NOALLOC_FLOATS * sizeof(float), impl,
                           as_stream(stream));
}

// Chain iq -> generator -> PA twin with the generator's scales already known (QAT: absmax[0..3] filled
// by the abs-max passes).  Large batches: one kernel (the generator output never makes the round trip
// through HBM; it is written only if the caller asks for it).  Otherwise: forward to the output, the
// output quantiser, then the PA kernel.
extern "C" int dpd_gen_pa_forward(const float *iq, int64_t B, int64_t S, const float *wq, float slope, int qat,
                                  const float *absmax, const float *coef_re, const float *coef_im, int K, int M,
                                  float *dpd_out, float *y, void *stream) {
    DPD_REQUIRE((y || B == 0) && coef_re && coef_im, DPD_ERR_INVALID_ARG, "gen_pa_forward: null pointer");
    DPD_REQUIRE(K >= 1 && K <= 8 && M >= 1 && M <= 16, DPD_ERR_INVALID_ARG, "gen_pa_forward: K in 1..8, M in 1..16");
    GenProblem p;
    int rc = make_problem(p, iq, B, S, 0, wq, slope, qat, const_cast<float *>(absmax));
    if (rc) return rc;
    DPD_REQUIRE(p.L >= M, DPD_ERR_SHAPE, "Sequence length %lld must be >= memory depth %d", (long long)p.L, M);
    if (p.N == 0) return DPD_OK;
    hipStream_t st = as_stream(stream);
    if (use_mfma(p, DPD_IMPL_AUTO)) return gen_mfma_pa_forward(p, coef_re, coef_im, K, M, dpd_out, y, st);
    DPD_REQUIRE(dpd_out, DPD_ERR_INVALID_ARG, "gen_pa_forward: small batches need dpd_out as the intermediate buffer");
    if (!p.qat) {
        rc = forward_dispatch(p, 4, dpd_out, DPD_IMPL_AUTO, st);
    } else {
        // the pass-3 kernel rewrites absmax[3] with the same value it already holds (max is idempotent)
        rc = forward_dispatch(p, 3, dpd_out, DPD_IMPL_AUTO, st);
        if (!rc) rc = gen_launch_quantize_output(dpd_out, p.N * 2, absmax, st);
    }
    if (rc) return rc;
    return dpd_pa_forward(dpd_out, B, p.L, coef_re, coef_im, K, M, y, stream);
}
