"""Memory-polynomial PA twin (oracle; test infrastructure only).

Follows reference models/pa_digital_twin.py:
  * coefficient initialisation  :80-114
  * model                       :125-170   y[n] = sum_{k in 1,3,5,7} sum_{m<M} h[k,m] x[n+M-1-m] |x[n+M-1-m]|^(k-1)
  * thermal coefficient sets    :220-278
The reference accumulates the 20 terms per output in complex64, term order k-major then m.
"""
import math
from typing import Tuple

import numpy as np
import torch

TEMP_TABLE = {  # pa_digital_twin.py:220-239
    "cold": dict(gain_scale=1.02, phase_offset_deg=-5, am_am_scale=1.05, am_pm_scale=0.95),
    "normal": dict(gain_scale=1.0, phase_offset_deg=0, am_am_scale=1.0, am_pm_scale=1.0),
    "hot": dict(gain_scale=0.95, phase_offset_deg=10, am_am_scale=0.92, am_pm_scale=1.15),
}


def init_coeffs(memory_depth: int = 5, nonlinear_order: int = 7, gain_db: float = 30.0) -> Tuple[torch.Tensor, torch.Tensor]:
    orders = list(range(1, nonlinear_order + 1, 2))
    re = torch.zeros(len(orders), memory_depth)
    im = torch.zeros(len(orders), memory_depth)
    g = 10 ** (gain_db / 20)
    re[0, 0] = g
    for m in range(1, memory_depth):
        d = 0.1 * np.exp(-m / 2)
        re[0, m] = d * g
        im[0, m] = 0.02 * d * g
    for ki, k in enumerate(orders[1:], start=1):
        am_am = -0.05 * g / (k ** 1.5)
        am_pm = 0.02 * g / (k ** 1.5)
        re[ki, 0] = am_am
        im[ki, 0] = am_pm
        for m in range(1, memory_depth):
            d = 0.3 * np.exp(-m / 1.5)
            re[ki, m] = am_am * d
            im[ki, m] = am_pm * d * 0.5
    return re, im


def thermal_coeffs(base_re: torch.Tensor, base_im: torch.Tensor, state: str) -> Tuple[torch.Tensor, torch.Tensor]:
    if state not in TEMP_TABLE:
        raise ValueError(f"Invalid temperature state: {state}")
    t = TEMP_TABLE[state]
    ph = t["phase_offset_deg"] * np.pi / 180
    re, im = base_re.clone(), base_im.clone()
    # linear row: both parts are built from the base REAL part (:269-270)
    re[0] = base_re[0] * t["gain_scale"] * np.cos(ph)
    im[0] = base_re[0] * t["gain_scale"] * np.sin(ph)
    for k in range(1, base_re.shape[0]):
        re[k] = base_re[k] * t["am_am_scale"]
        im[k] = base_im[k] * t["am_pm_scale"]
    return re, im


def volterra_forward(x: torch.Tensor, coef_re: torch.Tensor, coef_im: torch.Tensor) -> torch.Tensor:
    """x [B, S, 2] (or complex [B, S]) -> y [B, S-M+1, 2]; ValueError if S < M."""
    xc = torch.complex(x[..., 0], x[..., 1]) if (x.dim() == 3 and x.shape[-1] == 2) else x
    B, S = xc.shape
    K, M = coef_re.shape
    if S < M:
        raise ValueError(f"Sequence length {S} must be >= memory depth {M}")
    out_len = S - M + 1
    h = torch.complex(coef_re, coef_im)
    mag = xc.abs()
    y = torch.zeros(B, out_len, dtype=torch.complex64)
    for ki in range(K):
        k = 2 * ki + 1
        basis = xc * (mag ** (k - 1) if k > 1 else torch.ones_like(mag))
        for m in range(M):
            y = y + h[ki, m] * basis[:, M - 1 - m:M - 1 - m + out_len]
    return torch.stack([y.real, y.imag], dim=-1)
