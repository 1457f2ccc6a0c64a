/*
 * dpd_hip.h - C ABI of libdpd_hip.so: the batched DPD hot path on MI355X (gfx950).
 *
 * Every entry point takes plain device pointers and sizes; no framework types.  All tensor
 * arguments are DEVICE (HBM) pointers to contiguous fp32 (or int16 where stated) unless the
 * name starts with h_.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Calls are asynchronous on `stream`; no entry point synchronises the device or allocates
 * memory: scratch is caller-provided (`ws`, size from dpd_workspace_bytes()).
 *
 * Empty batches (B, N or n equal to 0) are valid everywhere: the data pointers may then be NULL and
 * outputs that are sums or gradients over the batch are zero-filled.
 *
 * Return value: DPD_OK or a DPD_ERR_* code; dpd_last_error() gives the text for the calling
 * thread.  DPD_ERR_SHAPE marks the conditions on which the reference raises ValueError.
 *
 * The reference (orpheus016/6g-pa-gan-dpd) has no FFI: its boundary is the Python API of
 * models/ and utils/.  Each entry below names the reference function it replaces.
 */
#ifndef DPD_HIP_H
#define DPD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPD_OK 0
#define DPD_ERR_INVALID_ARG 1 /* null pointer, negative size, unsupported option */
#define DPD_ERR_SHAPE 2       /* reference raises ValueError (sequence too short, bad state) */
#define DPD_ERR_HIP 3         /* a HIP runtime call failed */
#define DPD_ERR_WORKSPACE 4   /* workspace missing or too small */

/* Generator geometry fixed by the reference defaults (models/tdnn_generator.py:218-245):
 * memory depth 5, features 30, hidden 32 and 16, outputs 2. */
#define DPD_GEN_MEMORY 5
#define DPD_GEN_NFEAT 30
#define DPD_GEN_NPARAM 1554 /* flat order: W1[32x30] b1[32] W2[16x32] b2[16] W3[2x16] b3[2] */
#define DPD_GEN_NSCALE 8    /* absmax[0..3]: features, act1, act2, tanh out; [4..6]: W1, W2, W3 */

/* Critic geometry (models/discriminator.py:59-93 defaults): 4 -> 64 -> 32 -> 16 -> 1. */
#define DPD_CRITIC_NPARAM 2945 /* flat: W1[64x4] b1[64] W2[32x64] b2[32] W3[16x32] b3[16] W4[1x16] b4[1] */

/* kernel selection for the generator entry points */
#define DPD_IMPL_AUTO 0
#define DPD_IMPL_VALU 1 /* one sample per lane, weights as scalar operands */
#define DPD_IMPL_MFMA 2 /* fp32 MFMA tiles of 16 samples (IQ-row input only) */

const char *dpd_last_error(void);
int dpd_version(void);

/* Process-wide knobs.  "critic_impl": 0 = choose by size, 1 = VALU kernels, 2 = MFMA kernels.
 * "gen_deterministic": 1 = the MFMA generator backward assigns chunks to waves in a fixed order, which
 * makes its gradients bitwise reproducible from run to run (default 0: chunks are handed out first
 * come, first served, ~3 % faster, gradients reproducible to fp32 summation order).
 * "critic_deterministic": the same choice for the MFMA critic backward and gradient-penalty kernels
 * (default 0: tiles first come, first served, ~4 % faster; 1: fixed order, bitwise reproducible).
 * "chain_carry": how dpd_gen_pa_forward walks the rows when K = 4 and M = 5.  1 = every wave takes whole rows and
 * carries the last four generator outputs of a chunk over to the next, so no output is computed twice; 0 = chunks of 60
 * PA outputs that overlap by four generator outputs (any wave can take any chunk); 2 (default) = 1 when the batch has
 * enough rows to fill the waves evenly, else 0.  The results agree to fp32 rounding (same arithmetic per output).
 * "gp_fused": 1 (default) = dpd_critic_gp on rows of a multiple of 256 samples runs forward, input gradient and the
 * reverse chain in one pass over the samples and applies the row coefficients to per-unit partial sums afterwards;
 * 0 = the three-pass form for every shape.  Same results to fp32 summation order. */
int dpd_set_option(const char *name, int value);

/* Scratch size in bytes for the ws-taking entry points: dpd_workspace_bytes(0) suffices for every
 * call except dpd_critic_gp, which needs dpd_workspace_bytes(B * L). */
size_t dpd_workspace_bytes(int64_t n_samples);

/* ---------------------------------------------------------------- generator ------------- */

/* MemoryTapAssembly.forward (models/tdnn_generator.py:158-195): iq [B,S,2] -> feat [B,S-5,30].
 * DPD_ERR_SHAPE if S <= 5. */
int dpd_tap_assembly(const float *iq, int64_t B, int64_t S, float *feat, void *stream);

/* Its adjoint (the reference's MemoryTapAssembly is differentiable torch, models/tdnn_generator.py:158-195):
 * gfeat [B,S-5,30] -> dx [B,S,2], every point written.  The envelope columns contribute nothing where |x| = 0. */
int dpd_tap_assembly_backward(const float *iq, int64_t B, int64_t S, const float *gfeat, float *dx, void *stream);

/* Weight preparation shared by every generator call: packs the six parameter tensors into the
 * flat layout; with qat != 0 the weights are per-row fake-quantised (16 bit, scale = rowmax/32767,
 * models/tdnn_generator.py:394-403 with :74-103); biases stay fp32.  absmax[4..6] receive the
 * per-tensor |W| maxima used by the running-scale bookkeeping (:122-130). */
int dpd_gen_prepare_weights(const float *w1, const float *b1, const float *w2, const float *b2,
                            const float *w3, const float *b3, int qat, float *wq, float *absmax,
                            void *stream);

/* One abs-max pass of the QAT forward (the four per-tensor activation scales of
 * TDNNGeneratorQAT.forward, models/tdnn_generator.py:412-456, are global reductions that each
 * depend on the previous one).  stage 0: features -> absmax[0]; 1: act1 -> absmax[1];
 * 2: act2 -> absmax[2]; 3: tanh output -> absmax[3] and the unquantised output into `out`.
 * x is iq [B,S,2] (pre_assembled == 0) or features [B,30] (pre_assembled != 0, S ignored).
 * absmax[stage] is overwritten; earlier entries are read.  Between passes a data-parallel caller
 * may max-reduce absmax[stage] across ranks.  `out` is used by stage 3 only; there it may be NULL
 * for IQ input of >= 4096 samples (a caller that continues with dpd_gen_pa_forward needs no copy). */
int dpd_gen_absmax_pass(int stage, const float *x, int64_t B, int64_t S, int pre_assembled,
                        const float *wq, float slope, float *absmax, float *out, int impl, void *stream);

/* 1 if a generator call of this shape runs on the MFMA kernels, 0 if on the VALU kernels (the rule behind
 * DPD_IMPL_AUTO; with it a caller knows whether stage 3 of dpd_gen_absmax_pass may be given out == NULL).  Part of
 * the dispatch of TDNNGeneratorQAT.forward (models/tdnn_generator.py:412-456), which has no such choice. */
int dpd_gen_uses_mfma(int64_t B, int64_t S, int pre_assembled, int impl);

/* Final step of the QAT forward: out = fake_quant(out, absmax[3]) in place over n floats. */
int dpd_gen_quantize_output(float *out, int64_t n, const float *absmax, void *stream);

/* TDNNGenerator.forward / TDNNGeneratorQAT.forward (models/tdnn_generator.py:257-287, :412-456).
 * qat == 0: one fused kernel.  qat != 0: runs the four abs-max passes and the output quantiser on
 * `stream` (single-GPU semantics: scales from this call's batch).  out [B,S-5,2] or [B,2]. */
int dpd_gen_forward(const float *x, int64_t B, int64_t S, int pre_assembled, const float *wq, float slope,
                    int qat, float *absmax, float *out, int impl, void *stream);

/* Backward of the generator for an upstream gradient grad_out (same shape as out): recomputes the
 * forward from x with the scales in absmax (straight-through estimator, tdnn_generator.py:68-71)
 * and writes the flat parameter gradient grads[1554].  grad_feat (optional, pre_assembled only)
 * receives d loss / d features [B,30]. */
int dpd_gen_backward(const float *x, int64_t B, int64_t S, int pre_assembled, const float *wq, float slope,
                     int qat, const float *absmax, const float *grad_out, float *grads, float *grad_feat,
                     void *ws, size_t ws_bytes, int impl, void *stream);

/* Stage-1 supervised step core (training_colab.ipynb code cell 14, "STAGE 1"): forward,
 * MSE(out, target) and backward fused; nothing but grads[1554] and loss[0] (the mean squared
 * error) is written.  With qat != 0 absmax must already hold the four scales (run
 * dpd_gen_absmax_pass 0..3 first). */
int dpd_gen_mse_step(const float *iq, int64_t B, int64_t S, const float *target, const float *wq, float slope,
                     int qat, const float *absmax, float *grads, float *loss,
                     void *ws, size_t ws_bytes, int impl, void *stream);

/* dpd_gen_mse_step for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same
 * arguments minus the workspace: the partial-gradient slab and the chunk counter live in a device array
 * inside the library (6,390,016 bytes per GPU, held until the process ends), so the call makes no
 * allocation of any kind.  At most DPD_NOALLOC_MAX_SAMPLES = B * (S - 5) samples per call
 * (DPD_ERR_WORKSPACE above that).  All calls on one GPU share that GPU's array: calls that overlap on one
 * GPU (different streams or threads) race; calls on one stream are ordered and need nothing cleared. */
#define DPD_NOALLOC_MAX_SAMPLES ((int64_t)1 << 30)
int dpd_gen_mse_step_noalloc(const float *iq, int64_t B, int64_t S, const float *target, const float *wq,
                             float slope, int qat, const float *absmax, float *grads, float *loss, int impl,
                             void *stream);

/* ---------------------------------------------------------------- PA twin --------------- */

/* VolterraPA.forward (models/pa_digital_twin.py:125-170): x [B,S,2] -> y [B,S-M+1,2] with
 * K odd orders (1,3,..,2K-1) and M taps; coef_re/coef_im [K,M] on the device.
 * DPD_ERR_SHAPE if S < M.  K <= 8, M <= 16. */
int dpd_pa_forward(const float *x, int64_t B, int64_t S, const float *coef_re, const float *coef_im,
                   int K, int M, float *y, void *stream);

/* The generator's output quantiser (TDNNGeneratorQAT.forward, models/tdnn_generator.py:452-454) fused
 * into the PA forward: x [B,S,2] holds the unquantised generator output (dpd_gen_absmax_pass stage 3)
 * and absmax3 its abs-max; x is quantised in place and y = PA(quantised x). */
int dpd_quant_pa_forward(float *x, int64_t B, int64_t S, const float *absmax3, const float *coef_re,
                         const float *coef_im, int K, int M, float *y, void *stream);

/* Gradient of the above w.r.t. x for an upstream gy [B,S-M+1,2]; gx [B,S,2] is overwritten.
 * d|x|/dx at x == 0 is 0, as in torch. */
int dpd_pa_backward(const float *x, int64_t B, int64_t S, const float *coef_re, const float *coef_im,
                    int K, int M, const float *gy, float *gx, void *stream);

/* Chain iq -> generator -> PA twin (the validation chain of demo/video_demo.py:221-248):
 * y [B, S-5-M+1, 2] receives the PA output, dpd_out [B,S-5,2] the generator output (optional for
 * B*(S-5) >= 4096, where the whole chain is one kernel and the generator output stays on chip;
 * required below that).  With qat != 0 absmax[0..3] must already hold the four scales
 * (dpd_gen_absmax_pass 0..3).  DPD_ERR_SHAPE if S <= 5 or S-5 < M. */
int dpd_gen_pa_forward(const float *iq, int64_t B, int64_t S, const float *wq, float slope, int qat,
                       const float *absmax, const float *coef_re, const float *coef_im, int K, int M,
                       float *dpd_out, float *y, void *stream);

/* Backward of that chain for an upstream gradient gy [B, S-5-M+1, 2] on the PA output: grads[1554] w.r.t.
 * the generator parameters (VolterraPA.forward, models/pa_digital_twin.py:125-170, composed with
 * TDNNGeneratorQAT.forward, models/tdnn_generator.py:412-456, under autograd; the output quantiser passes
 * the gradient through unchanged, :68-71).  For B*(S-5) >= 4096, M == 5 and K <= 4 (the reference's PA)
 * one kernel recomputes the forward and runs PA adjoint and generator backward per tile pair: neither the
 * generator output nor its gradient goes through HBM, and dpd_workspace_bytes(0) suffices.  Other shapes
 * run on the separate kernels and need dpd_workspace_bytes(B * (S-5)).  absmax as for dpd_gen_pa_forward. */
int dpd_gen_pa_backward(const float *iq, int64_t B, int64_t S, const float *wq, float slope, int qat,
                        const float *absmax, const float *coef_re, const float *coef_im, int K, int M,
                        const float *gy, float *grads, void *ws, size_t ws_bytes, void *stream);

/* The same with the loss inside: loss[0] = mean((PA(G(iq)) - target)^2) over target [B, S-5-M+1, 2] and
 * grads[1554] = its gradient; nothing else is written.  This is torch.nn.functional.mse_loss on
 * PADigitalTwin.forward(generator(iq), add_noise=False) (models/pa_digital_twin.py:311-338) under autograd:
 * the twin is a differentiable module, although the reference's own scripts chain the two modules for
 * evaluation only (validate.py:91-119, demo/video_demo.py:221-248). */
int dpd_gen_pa_mse_step(const float *iq, int64_t B, int64_t S, const float *target, const float *wq, float slope,
                        int qat, const float *absmax, const float *coef_re, const float *coef_im, int K, int M,
                        float *grads, float *loss, void *ws, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------- signal impairments ---- */

/* Bytes of scratch for the two calls below on [B, S, 2] rows. */
size_t dpd_impair_workspace_bytes(int64_t B, int64_t S);

/* Training-time augmentation of IQ rows x [B,S,2] -> out [B,S,2] (may be x): `augment_signal` of the reference's
 * notebook (training_colab.ipynb, data-loading cell) with one set of draws per row.  params [B,8] per row:
 * 0 noise_rel and 1 noise_mul (noise std = sqrt(mean(row^2) * noise_rel) * noise_mul; the notebook's
 * 1 / 10^(snr_db/10) and `strength`; noise_mul = 0: no noise), 2, 3 cos / sin of the phase offset, 4 gain,
 * 5 thermal gain shift (0: no thermal stage), 6 thermal phase shift in rad, 7 AM-AM compression: the stages run in
 * the notebook's order (noise, phase, gain, thermal with gain_shift * (1 - 0.1 |v|^2 compression)).
 * noise [B,S,2]: standard normal draws, or NULL: drawn in the kernel from a counter-based generator keyed by
 * `seed` (sample index as the counter; same seed, same draws). */
int dpd_augment_rows(const float *x, int64_t B, int64_t S, const float *params, const float *noise, uint64_t seed,
                     float *out, void *ws, size_t ws_bytes, void *stream);

/* PADigitalTwin.add_noise (models/pa_digital_twin.py:280-309) on x [B,S,2] -> out: every sample rotated by
 * phase_noise_rad * (a standard normal draw), plus white noise of std sqrt(mean(row^2) * params[b,0]) * params[b,1]
 * (params [B,8] as above, entries 2..7 unused).  The reference takes the power over the whole tensor: call with
 * B = 1 and S = all samples for that.  noise [B,S,2] and phase_noise [B,S]: the draws, or both NULL: drawn in the
 * kernel from `seed`. */
int dpd_pa_add_noise(const float *x, int64_t B, int64_t S, const float *params, float phase_noise_rad,
                     const float *noise, const float *phase_noise, uint64_t seed, float *out, void *ws,
                     size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------- critic ---------------- */

/* Spectral normalisation of the four critic layers (SpectralNormLinear, models/discriminator.py:28-42;
 * torch.nn.utils.spectral_norm, legacy hook form) in one launch.  w_orig, bias, u, v: host arrays of
 * four device pointers (layers 4->64, 64->32, 32->16, 16->1).  With power_iteration != 0 (training-mode
 * forward) u and v are advanced in place by one iteration first.  wb [2945] receives the flat operand of
 * the critic entry points (W_l / sigma_l and b_l); uv_saved [229] the u, v actually used and sigma [4]
 * the spectral norms, both needed by the backward. */
int dpd_critic_sn_forward(const float *const *w_orig, const float *const *bias, float *const *u,
                          float *const *v, int power_iteration, float eps, float *wb, float *uv_saved,
                          float *sigma, void *stream);

/* Gradient w.r.t. weight_orig / bias (flat, the layout of wb) from the gradient gwb [2945] w.r.t. wb, with
 * u, v held constant as in the reference: dW = (G - <G, W/sigma> u v^T) / sigma. */
int dpd_critic_sn_backward(const float *const *w_orig, const float *gwb, const float *uv_saved,
                           const float *sigma, float *grads, void *stream);

/* The three evaluations of one critic update (WassersteinLoss.discriminator_loss, models/discriminator.py:204-244:
 * D(real), D(fake), D(interpolate), each a forward with its own power iteration) in one launch each way:
 * dpd_critic_sn_forward3 = three successive dpd_critic_sn_forward, evaluation k into wb + k * 2945,
 * uv_saved + k * 229, sigma + k * 4; dpd_critic_sn_backward3: grads [2945] = (g_0 + g_1) + coef2 * g_2 with g_k
 * what dpd_critic_sn_backward gives for gwb + k * 2945 and the saved u, v, sigma of evaluation k. */
int dpd_critic_sn_forward3(const float *const *w_orig, const float *const *bias, float *const *u,
                           float *const *v, int power_iteration, float eps, float *wb, float *uv_saved,
                           float *sigma, void *stream);
int dpd_critic_sn_backward3(const float *const *w_orig, const float *gwb, const float *uv_saved,
                            const float *sigma, float coef2, float *grads, void *stream);

/* Discriminator.forward (models/discriminator.py:103-136) given the already spectrally
 * normalised weights wb (flat, DPD_CRITIC_NPARAM): score[n] for n < N rows of x [N,2], cond [N,2].
 * score_sum (optional) receives sum_n score[n]. */
int dpd_critic_forward(const float *x, const float *cond, int64_t N, const float *wb, float slope,
                       float *score, float *score_sum, void *stream);

/* Backward for an upstream gscore[N] (or, if gscore is NULL, the constant gconst for every row):
 * grads[2945] w.r.t. wb and/or gx [N,2] w.r.t. x (either may be NULL, not both; ws is only needed
 * when grads is requested).  score_sum (optional) receives sum_n score[n] of the recomputed forward, so
 * a loss that is linear in the scores (the Wasserstein terms, models/discriminator.py:219-226, 259-260)
 * needs no separate forward call. */
int dpd_critic_backward(const float *x, const float *cond, int64_t N, const float *wb, float slope,
                        const float *gscore, float gconst, float *grads, float *gx, float *score_sum,
                        void *ws, size_t ws_bytes, void *stream);

/* compute_gradient_penalty (models/discriminator.py:139-191) with its double backward fused:
 * x_hat = alpha[b]*real + (1-alpha[b])*fake per batch row b (rows of L samples; L = 1 for [B,2]
 * inputs), g = d sum D(x_hat, cond) / d x_hat, gp = mean_b (||g_b||_2 - 1)^2.
 * Writes gp[0], the row norms rownorm[B] and grads[2945] = d gp / d wb (bias entries are zero). */
int dpd_critic_gp(const float *real, const float *fake, const float *cond, const float *alpha,
                  int64_t B, int64_t L, const float *wb, float slope, float *gp, float *rownorm,
                  float *grads, void *ws, size_t ws_bytes, void *stream);

/* ---------------------------------------------------------------- spectral loss --------- */

/* Per-row statistics behind compute_evm / compute_acpr / compute_nmse / SpectralLoss
 * (utils/spectral_loss.py:24-167, :209-257) for pred, target [B,L,2]; L a power of two, 2..4096.
 * band [3,L] uint8 masks (main, lower, upper) as produced from fftfreq (:103-110).
 * stats [B,8]: sum|p-t|, sum|p-t|^2, sum|t|^2, P_main, P_lower, P_upper (of pred, |X|^2/L), 0, 0.
 * target may be NULL (ACPR only): the first three entries are then 0.  band may be NULL (EVM / NMSE /
 * L1 only): no transform is done, any L >= 1 is accepted and the band powers are 0. */
int dpd_spectral_stats(const float *pred, const float *target, int64_t B, int64_t L,
                       const uint8_t *band, float *stats, void *stream);

/* Backward for gstats [B,8] = d loss / d stats (entries 2, 6, 7 ignored):
 * dpred [B,L,2] = gstats[b,0] * sign(p-t) + 2 * gstats[b,1] * (p-t)
 *               + (2/L) * IDFT_unnormalised((gstats[b,3]*main + gstats[b,4]*lower + gstats[b,5]*upper) * X),
 * X = FFT(pred row b).  target may be NULL (the first two terms are then dropped). */
int dpd_spectral_backward(const float *pred, const float *target, int64_t B, int64_t L,
                          const uint8_t *band, const float *gstats, float *dpred, void *stream);

/* The loss arithmetic of SpectralLoss.forward (utils/spectral_loss.py:209-257) on the row statistics,
 * split so that a data-parallel caller can all-reduce `sums` between the first two calls.
 * dpd_spectral_rowdb: rowdb [B,4] (optional) = per-row EVM_dB, ACPR_lower_dB, ACPR_upper_dB, NMSE_dB
 * (compute_evm :24-64, compute_acpr :67-132, compute_nmse :135-167); sums[8] = (sum|p-t|, sum EVM_dB,
 * sum lower, sum upper, sum NMSE_dB, B, 0, 0), reduced in a fixed order.
 * dpd_spectral_loss_final: out[8] = (total, L1, EVM term, ACPR term, d total / d sums[0..3]) with
 * sums[5] as the (global) row count.
 * dpd_spectral_terms_backward: gstats [B,8] = gscale[0] (device scalar, NULL = 1) * d total / d stats,
 * the input of dpd_spectral_backward; coef = out + 4. */
int dpd_spectral_rowdb(const float *stats, int64_t B, int64_t L, float *rowdb, float *sums, void *stream);
int dpd_spectral_loss_final(const float *sums, int64_t L, float w_l1, float w_evm, float w_acpr, float *out,
                            void *stream);
int dpd_spectral_terms_backward(const float *stats, const float *coef, const float *gscale, int64_t B,
                                int64_t L, float *gstats, void *stream);

/* ---------------------------------------------------------------- fixed point ----------- */

/* quantize_to_fixed_point / fake_quantize (utils/quantization.py:88-161): q = clamp(rint(x*scale),
 * qmin, qmax); out = q (dequant == 0) or q/scale (dequant != 0). */
int dpd_quantize_fixed(const float *x, int64_t n, float scale, int bits, int is_signed, int dequant,
                       float *out, void *stream);

/* QuantizedLinear.forward (utils/quantization.py:213-233): out [N, out] = fq_act(fq_act(x) fq_w(w)^T + bias)
 * with fq(v) = clamp(rint(v * scale), qmin, qmax) / scale for the given scales and bit widths; bias (fp32,
 * may be NULL) is not quantised.  The sum is formed exactly (fp64 over exact products), so for the Q8.8 /
 * Q1.15 power-of-two scales the result is that of the integer datapath, independent of summation order. */
int dpd_quantized_linear(const float *x, const float *w, const float *bias, int64_t N, int in_features,
                         int out_features, float act_scale, int act_bits, int act_signed, float w_scale, int w_bits,
                         int w_signed, float *out, void *stream);

/* Its backward with the straight-through estimator of FakeQuantizeFunction (utils/quantization.py:164-173):
 * gx [N, in] = gy fq_w(w), gw [out, in] = gy^T fq_act(x), gb [out] = column sums of gy; each may be NULL. */
int dpd_quantized_linear_backward(const float *x, const float *w, const float *gy, int64_t N, int in_features,
                                  int out_features, float act_scale, int act_bits, int act_signed, float w_scale,
                                  int w_bits, int w_signed, float *gx, float *gw, float *gb, void *stream);

/* fake_quantize with a data-derived scale (models/tdnn_generator.py:74-103): x [rows, cols];
 * per_row == 0: one scale for the tensor, else one per row (rows <= 65535).  absmax_out [1] or [rows]
 * is required: it is the reduction scratch and returns the abs-max values. */
int dpd_fake_quantize_dynamic(const float *x, int64_t rows, int64_t cols, int per_row, int n_bits,
                              float *out, float *absmax_out, void *stream);

/* FakeQuantizeModule.observe (models/tdnn_generator.py:121-130) for n <= 64 quantisers at once, `times` times over
 * with the same observation: running[i] <- (1 - momentum) * running[i] + momentum * max(absmax[i], 1e-8) / q_max,
 * q_max = 2^(n_bits-1) - 1 (a caller that reuses one forward for several steps repeats the bookkeeping only). */
int dpd_qat_observe(const float *absmax, float *running, int n, float momentum, int n_bits, int times,
                    void *stream);

/* Bit-exact Q8.8 inference: iq int16 Q1.15 [B,S,2] -> out int16 Q1.15 [B,S-5,2].
 * Weights int16 Q1.15 (w1 [32,30], w2 [16,32], w3 [2,16]), biases int16 Q8.8, accumulators Q9.23,
 * requantisation round-half-even with saturation (utils/quantization.py:88-120, :213-256),
 * tanh through the 256-entry table of generate_tanh_lut (:365-393).  Integer VALU kernels: acc64 != 0 forces
 * 64-bit accumulators; acc64 == 0 (v_dot2_i32_i16, 32-bit accumulators) requires dpd_q88_int32_safe() to hold
 * for the weights. */
int dpd_q88_infer(const int16_t *iq, int64_t B, int64_t S, const int16_t *w1, const int16_t *b1,
                  const int16_t *w2, const int16_t *b2, const int16_t *w3, const int16_t *b3,
                  const int16_t *lut, int acc64, int16_t *out, void *stream);

/* Host-side interval check (h_ = host pointers): 1 if 32-bit accumulation is provably exact for
 * every input, 0 otherwise. */
int dpd_q88_int32_safe(const int16_t *h_w1, const int16_t *h_b1, const int16_t *h_w2, const int16_t *h_b2,
                       const int16_t *h_w3, const int16_t *h_b3);

/* The same inference on the int8 matrix cores (v_mfma_i32_16x16x64_i8): every int16 as two 8-bit digits, three
 * int32 accumulators per product, bit-identical to dpd_q88_infer.  Allowed when dpd_q88_int32_safe() holds and every
 * weight is <= 32639 (dpd_q88_digits_fit(), host pointers: a larger weight has no signed 8-bit high digit).
 * dpd_q88_matrix_prepare builds the weight image (dpd_q88_matrix_image_bytes() bytes of device memory, 16-byte
 * aligned, caller-owned) from the device weights once per set of weights; dpd_q88_matrix_infer then serves any number
 * of batches from it.  Same reference contract as dpd_q88_infer (it stands for the same composition of
 * utils/quantization.py:88-120, :213-256, :365-393). */
int dpd_q88_digits_fit(const int16_t *h_w1, const int16_t *h_w2, const int16_t *h_w3);
size_t dpd_q88_matrix_image_bytes(void);
int dpd_q88_matrix_prepare(const int16_t *w1, const int16_t *b1, const int16_t *w2, const int16_t *b2,
                           const int16_t *w3, const int16_t *b3, void *image, void *stream);
int dpd_q88_matrix_infer(const int16_t *iq, int64_t B, int64_t S, const void *image, const int16_t *lut,
                         int16_t *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DPD_HIP_H */
