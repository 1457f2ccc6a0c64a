"""TDNN generator 30->32->16->2, float and dynamic-range fake-quant (QAT) forms (oracle).

Follows reference models/tdnn_generator.py:
  * dynamic fake quantisation  :46-103  (scale = max(absmax, 1e-8)/32767, round-half-even,
    clamp to [-32768, 32767], straight-through gradient)
  * running-scale EMA          :122-132 (momentum 0.1, per-tensor abs-max even for the per-row
    weight quantisers; never used for quantising)
  * float forward              :257-287
  * QAT forward                :394-456 (input quant -> [row-quantised W, fp32 bias, LeakyReLU,
    activation quant] x2 -> row-quantised W3 -> tanh -> output quant)
Parameters are passed as a dict {fc1.weight, fc1.bias, fc2.weight, fc2.bias, fc3.weight, fc3.bias}.
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .taps import tap_assembly

PARAM_KEYS = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias")
ACT_QUANT_KEYS = ("input_quant", "fc1_act_quant", "fc2_act_quant", "output_quant")
WEIGHT_QUANT_KEYS = ("fc1_weight_quant", "fc2_weight_quant", "fc3_weight_quant")


def dyn_scale(x: torch.Tensor, n_bits: int = 16, per_row: bool = False) -> torch.Tensor:
    q_max = 2 ** (n_bits - 1) - 1
    if per_row:
        amax = x.abs().amax(dim=list(range(1, x.dim())), keepdim=True)
    else:
        amax = x.abs().max()
    return torch.clamp(amax, min=1e-8) / q_max


class _STE(torch.autograd.Function):
    """Returns `xq` in the forward pass and hands the incoming gradient to `x` unchanged."""

    @staticmethod
    def forward(ctx, x, xq):
        return xq.clone()

    @staticmethod
    def backward(ctx, g):
        return g, None


def fake_quant_dyn(x: torch.Tensor, n_bits: int = 16, per_row: bool = False) -> torch.Tensor:
    """Quantise-dequantise with a scale taken from the data; identity gradient."""
    q_min, q_max = -(2 ** (n_bits - 1)), 2 ** (n_bits - 1) - 1
    xd = x.detach()
    scale = dyn_scale(xd, n_bits, per_row)
    xq = torch.clamp(torch.round(xd / scale), q_min, q_max) * scale
    return _STE.apply(x, xq)


def ema_scale(running: torch.Tensor, x: torch.Tensor, n_bits: int = 16, momentum: float = 0.1) -> torch.Tensor:
    """running_scale update done by every quantiser module in training mode (:123-130)."""
    return (1 - momentum) * running + momentum * dyn_scale(x.detach(), n_bits, per_row=False)


def gen_forward(x: torch.Tensor, p: Dict[str, torch.Tensor], slope: float = 0.2, qat: bool = False,
                pre_assembled: bool = False, n_bits: int = 16, memory_depth: int = 5,
                running: Optional[Dict[str, torch.Tensor]] = None,
                absmax_out: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """x: IQ rows [B, S, 2] (-> [B, S-M, 2]) or, if pre_assembled, features [N, 30] (-> [N, 2]).

    If `running` is given (training mode) it is updated in place with the seven EMA scales.
    If `absmax_out` is given (QAT) it receives the per-tensor abs-max behind each of the seven quantiser scales.
    """
    if pre_assembled:
        f, shape = x, None
    else:
        f = tap_assembly(x, memory_depth)
        shape = f.shape[:2]
        f = f.reshape(-1, f.shape[-1])
    w1, b1, w2, b2, w3, b3 = (p[k] for k in PARAM_KEYS)
    if not qat:
        h = F.leaky_relu(F.linear(f, w1, b1), slope)
        h = F.leaky_relu(F.linear(h, w2, b2), slope)
        y = torch.tanh(F.linear(h, w3, b3))
    else:
        def act_q(t, key):
            if running is not None:
                running[key] = ema_scale(running[key], t, n_bits)
            if absmax_out is not None:
                absmax_out[key] = t.detach().abs().max()
            return fake_quant_dyn(t, n_bits)

        def w_q(w, key):
            if running is not None:
                running[key] = ema_scale(running[key], w, n_bits)
            if absmax_out is not None:
                absmax_out[key] = w.detach().abs().max()
            return fake_quant_dyn(w, n_bits, per_row=True)

        xq = act_q(f, "input_quant")
        h = act_q(F.leaky_relu(F.linear(xq, w_q(w1, "fc1_weight_quant"), b1), slope), "fc1_act_quant")
        h = act_q(F.leaky_relu(F.linear(h, w_q(w2, "fc2_weight_quant"), b2), slope), "fc2_act_quant")
        y = act_q(torch.tanh(F.linear(h, w_q(w3, "fc3_weight_quant"), b3)), "output_quant")
    if shape is not None:
        y = y.reshape(shape[0], shape[1], 2)
    return y


def gen_mse_grads(x, target, p, **kw) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor]]:
    """MSE(gen_forward(x), target) with its gradients for the six parameters (autograd)."""
    pp = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    out = gen_forward(x, pp, **kw)
    loss = F.mse_loss(out, target)
    grads = torch.autograd.grad(loss, [pp[k] for k in PARAM_KEYS])
    return out.detach(), loss.detach(), dict(zip(PARAM_KEYS, grads))


def export_int16_weights(p: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Per-tensor symmetric int16 export (tdnn_generator.py:458-488); biases stay float."""
    out = {}
    for i in (1, 2, 3):
        w = p[f"fc{i}.weight"]
        scale = w.abs().max() / (2 ** 15 - 1)
        out[f"fc{i}_weight"] = torch.round(w / scale).to(torch.int16)
        out[f"fc{i}_weight_scale"] = scale
        out[f"fc{i}_bias"] = p[f"fc{i}.bias"]
    return out
