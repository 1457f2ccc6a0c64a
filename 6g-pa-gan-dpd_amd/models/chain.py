"""Chain iq -> DPD generator -> PA twin as one call (demo/video_demo.py:221-248, validate.py:91-119: the
generator output feeds the PA model and the PA output is compared with the input), with its backward:
the PA output carries an autograd node whose gradient reaches the generator parameters through the fused
chain-backward kernel (PA adjoint and generator backward per tile pair; neither the generator output nor
its gradient goes through HBM)."""
from typing import Optional, Tuple

import torch

from .. import _lib as L
from .pa_digital_twin import PADigitalTwin, VolterraPA
from .tdnn_generator import TDNNGenerator, TDNNGeneratorQAT, iq_rows_backward

_SIZES = [960, 32, 512, 16, 32, 2]


class _ChainFn(torch.autograd.Function):
    """(y, dpd) = (PA(G(iq)), G(iq)); gradients flow from y to the six generator parameters."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, cre, cim, slope, qat, absmax_sync, want_dpd):
        B, S, _ = x.shape
        K, M = cre.shape
        dev = x.device
        st = L.stream()
        wq = torch.empty(L.GEN_NPARAM, dtype=torch.float32, device=dev)
        absmax = torch.zeros(L.GEN_NSCALE, dtype=torch.float32, device=dev)
        L.call("dpd_gen_prepare_weights", *(L.fptr(L.as_f32(t.detach())) for t in (w1, b1, w2, b2, w3, b3)), int(qat),
               L.fptr(wq), L.fptr(absmax), st)
        y = torch.empty(B, S - 5 - M + 1, 2, dtype=torch.float32, device=dev)
        if qat:
            # the fourth abs-max pass already produces the (unquantised) output: quantiser + PA finish it
            # in one bandwidth-bound kernel, cheaper than recomputing the generator inside a fused kernel
            dpd = torch.empty(B, S - 5, 2, dtype=torch.float32, device=dev)
            for stage in range(4):
                L.call("dpd_gen_absmax_pass", stage, L.fptr(x), B, S, 0, L.fptr(wq), slope, L.fptr(absmax),
                       L.fptr(dpd) if stage == 3 else L.c_p(0), 0, st)
                if absmax_sync is not None:
                    absmax_sync(absmax[stage:stage + 1])
            L.call("dpd_quant_pa_forward", L.fptr(dpd), B, S - 5, L.fptr(absmax[3:4]), L.fptr(cre), L.fptr(cim), K, M,
                   L.fptr(y), st)
        else:
            fused = B * (S - 5) >= 4096
            dpd = torch.empty(B, S - 5, 2, dtype=torch.float32, device=dev) if (want_dpd or not fused) else None
            L.call("dpd_gen_pa_forward", L.fptr(x), B, S, L.fptr(wq), slope, 0, L.fptr(absmax), L.fptr(cre), L.fptr(cim),
                   K, M, L.fptr(dpd), L.fptr(y), st)
        ctx.save_for_backward(x, wq, absmax, cre, cim)
        ctx.cfg = (B, S, K, M, slope, bool(qat))
        ctx.shapes = [t.shape for t in (w1, b1, w2, b2, w3, b3)]
        if dpd is None:
            dpd = torch.empty(0, dtype=torch.float32, device=dev)
        ctx.mark_non_differentiable(dpd)
        return y, dpd

    @staticmethod
    def backward(ctx, gy, _gdpd):
        x, wq, absmax, cre, cim = ctx.saved_tensors
        B, S, K, M, slope, qat = ctx.cfg
        grads = torch.empty(L.GEN_NPARAM, dtype=torch.float32, device=x.device)
        if ctx.needs_input_grad[0]:
            # the IQ rows ask for a gradient: the separate kernels (generator output recomputed with the scales of the
            # forward, PA adjoint, generator backward on pre-assembled features, adjoint of the tap assembly)
            st = L.stream()
            dpd = torch.empty(B, S - 5, 2, dtype=torch.float32, device=x.device)
            if qat:
                tmp = absmax.clone()
                L.call("dpd_gen_absmax_pass", 3, L.fptr(x), B, S, 0, L.fptr(wq), slope, L.fptr(tmp), L.fptr(dpd), 0, st)
                L.call("dpd_gen_quantize_output", L.fptr(dpd), dpd.numel(), L.fptr(absmax), st)
            else:
                L.call("dpd_gen_forward", L.fptr(x), B, S, 0, L.fptr(wq), slope, 0, L.fptr(absmax), L.fptr(dpd), 0, st)
            gd = torch.empty_like(dpd)
            L.call("dpd_pa_backward", L.fptr(dpd), B, S - 5, L.fptr(cre), L.fptr(cim), K, M, L.fptr(L.as_f32(gy)), L.fptr(gd), st)
            gx = iq_rows_backward(x, B, S, wq, slope, qat, absmax, gd, grads)
            parts = [g.view(s) for g, s in zip(torch.split(grads, _SIZES), ctx.shapes)]
            return (gx, *parts, None, None, None, None, None, None)
        _, wsp, wsn = L.workspace(x.device, B * (S - 5))
        L.call("dpd_gen_pa_backward", L.fptr(x), B, S, L.fptr(wq), slope, int(qat), L.fptr(absmax), L.fptr(cre),
               L.fptr(cim), K, M, L.fptr(L.as_f32(gy)), L.fptr(grads), wsp, wsn, L.stream())
        parts = [g.view(s) for g, s in zip(torch.split(grads, _SIZES), ctx.shapes)]
        return (None, *parts, None, None, None, None, None, None)


def dpd_pa_chain(generator, pa, iq: torch.Tensor, return_dpd: bool = True,
                 absmax_sync=None) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """(generator(iq), pa(generator(iq))) for iq [B, S, 2]; the PA runs without noise.

    Float generator, B*(S-5) >= 4096: one kernel; with `return_dpd=False` the generator output never
    leaves the chip and None is returned in its place.  QAT generator: the four abs-max passes, then the
    output quantiser fused into the PA kernel.  The PA output is differentiable w.r.t. the generator
    parameters (the returned generator output is detached).  Does not touch the generator's
    running-scale buffers (an evaluation path, like `generator.eval()`)."""
    L.need_gpu(iq)
    vpa = pa.pa_model if isinstance(pa, PADigitalTwin) else pa
    if not isinstance(vpa, VolterraPA):
        raise TypeError("pa must be a PADigitalTwin or a VolterraPA")
    if not isinstance(generator, (TDNNGenerator, TDNNGeneratorQAT)):
        raise TypeError("generator must be a TDNNGenerator or a TDNNGeneratorQAT")
    x = L.as_f32(iq)
    if x.dim() != 3 or x.shape[-1] != 2:
        raise RuntimeError(f"iq must be [B, S, 2], got {tuple(x.shape)}")
    B, S, _ = x.shape
    if S <= 5:
        raise ValueError(f"Sequence length {S} must be > memory depth 5")
    K, M = vpa.coeffs_real.shape
    if S - 5 < M:
        raise ValueError(f"Sequence length {S - 5} must be >= memory depth {M}")
    qat = bool(getattr(generator, "qat_enabled", False))
    cre, cim = L.as_f32(vpa.coeffs_real), L.as_f32(vpa.coeffs_imag)
    y, dpd = _ChainFn.apply(x, generator.fc1.weight, generator.fc1.bias, generator.fc2.weight, generator.fc2.bias,
                            generator.fc3.weight, generator.fc3.bias, cre, cim, float(generator.leaky_slope), qat,
                            absmax_sync, bool(return_dpd))
    return (dpd if (return_dpd and dpd.numel()) else None), y
