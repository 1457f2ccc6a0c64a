"""Validation chain iq -> DPD generator -> PA twin as one call (demo/video_demo.py:221-248,
validate.py: the generator output feeds the PA model and the PA output is compared with the input).

Inference only (no autograd graph): for training use the modules, whose ops carry gradients."""
from typing import Optional, Tuple

import torch

from .. import _lib as L
from .pa_digital_twin import PADigitalTwin, VolterraPA
from .tdnn_generator import TDNNGenerator, TDNNGeneratorQAT


@torch.no_grad()
def dpd_pa_chain(generator, pa, iq: torch.Tensor, return_dpd: bool = True,
                 absmax_sync=None) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """(generator(iq), pa(generator(iq))) for iq [B, S, 2]; the PA runs without noise.

    Float generator, B*(S-5) >= 4096: one kernel; with `return_dpd=False` the generator output never
    leaves the chip and None is returned in its place.  QAT generator: the four abs-max passes, then the
    output quantiser fused into the PA kernel.  Does not touch the
    generator's running-scale buffers (an evaluation path, like `generator.eval()`)."""
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
    dev = x.device
    st = L.stream()
    wq = torch.empty(L.GEN_NPARAM, dtype=torch.float32, device=dev)
    absmax = torch.zeros(L.GEN_NSCALE, dtype=torch.float32, device=dev)
    params = (generator.fc1.weight, generator.fc1.bias, generator.fc2.weight, generator.fc2.bias, generator.fc3.weight,
              generator.fc3.bias)
    L.call("dpd_gen_prepare_weights", *(L.fptr(L.as_f32(t.detach())) for t in params), int(qat), L.fptr(wq),
           L.fptr(absmax), st)
    slope = float(generator.leaky_slope)
    cre, cim = L.as_f32(vpa.coeffs_real), L.as_f32(vpa.coeffs_imag)
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
        return (dpd if return_dpd else None), y
    fused = B * (S - 5) >= 4096
    dpd = torch.empty(B, S - 5, 2, dtype=torch.float32, device=dev) if (return_dpd or not fused) else None
    L.call("dpd_gen_pa_forward", L.fptr(x), B, S, L.fptr(wq), slope, 0, L.fptr(absmax), L.fptr(cre), L.fptr(cim), K, M,
           L.fptr(dpd), L.fptr(y), st)
    return (dpd if return_dpd else None), y
