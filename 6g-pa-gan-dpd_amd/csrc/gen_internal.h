// Internal launch interface between the generator C ABI (gen_api.hip) and its two kernel families.
#pragma once
#include "common.h"

struct GenProblem {
    const float *x;      // iq [B,S,2] or features [B,30]
    int64_t B, S;        // S unused when pre
    int64_t L;           // outputs per row (S-5, or 1 when pre)
    int64_t N;           // B*L samples
    bool pre;            // features are pre-assembled
    bool qat;
    const float *wq;     // flat prepared weights [1554]
    float slope;
    float *absmax;       // [8]; read (and for passes: written at [stage])
};

// depth: 0..3 = QAT abs-max passes, 4 = full forward writing `out`.
int gen_valu_forward(const GenProblem &p, int depth, float *out, hipStream_t st);
// gradient source: grad_out != NULL -> upstream gradient; else MSE against target.
int gen_valu_backward(const GenProblem &p, const float *grad_out, const float *target, float *grad_feat,
                      float *slab, int *n_partials, hipStream_t st);

int gen_mfma_forward(const GenProblem &p, int depth, float *out, hipStream_t st);
// one kernel: generator forward (final scales in p.absmax when p.qat) + memory-polynomial PA; out may be NULL
int gen_mfma_pa_forward(const GenProblem &p, const float *cre, const float *cim, int K, int M, float *out, float *y,
                        hipStream_t st);
// chain != NULL: the loss / upstream gradient lives on the PA output [B, L - 4, 2] (generator -> PA twin, fused)
struct GenPaChain {
    const float *cre, *cim;   // [K, M]
    int K, M;
};
int gen_mfma_backward(const GenProblem &p, const float *grad_out, const float *target,
                      float *slab, int *n_partials, hipStream_t st, const GenPaChain *chain = nullptr);

// grads[1554] (+ loss) = sum over n_partials slab rows (double accumulation, deterministic order)
int gen_reduce_slab(const float *slab, int n_partials, float *grads, float *loss, float loss_scale, hipStream_t st);
