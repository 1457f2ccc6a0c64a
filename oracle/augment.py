"""Signal impairments (oracle; test infrastructure only).

Follows the reference's
  * notebook augmentation `augment_signal`   training_colab.ipynb, data-loading code cell (noise, phase, gain, thermal)
  * PA twin noise model `add_noise`          models/pa_digital_twin.py:280-309
with the random numbers passed in (the reference takes them from torch's global generator, in this order:
snr = randn(1), noise = randn_like(signal), phase = rand(1), gain = rand(1), thermal = rand(1);
add_noise: awgn = randn_like(x), phase = randn(B, S, 1)).
"""
import math

import torch

THERMAL = {"cold": (1.05, -0.02, 0.98), "hot": (0.95, 0.03, 1.02)}   # gain shift, phase shift, compression


def augment_signal(signal: torch.Tensor, augment_type: str, strength: float, snr: float, noise: torch.Tensor,
                   phase: float, gain: float, thermal: float) -> torch.Tensor:
    """signal [n, 2] -> [n, 2]; the stages and their order are the notebook's."""
    a = signal.clone()
    if augment_type in ("noise", "all"):
        snr_db = 40 + snr * 5
        noise_power = torch.mean(signal ** 2) / (10 ** (snr_db / 10))
        a = a + noise * torch.sqrt(noise_power) * strength
    if augment_type in ("phase", "all"):
        off = torch.tensor((phase - 0.5) * 0.1 * strength)
        c, s = torch.cos(off), torch.sin(off)
        a = torch.stack([a[:, 0] * c - a[:, 1] * s, a[:, 0] * s + a[:, 1] * c], dim=1)
    if augment_type in ("gain", "all"):
        a = a * (1.0 + (gain - 0.5) * 0.2 * strength)
    if augment_type in ("thermal", "all"):
        if 0.33 <= thermal < 0.67:
            return a
        gs, ps, comp = THERMAL["cold" if thermal < 0.33 else "hot"]
        mag = torch.sqrt(a[:, 0] ** 2 + a[:, 1] ** 2)
        ph = torch.atan2(a[:, 1], a[:, 0]) + ps
        g = gs * (1.0 - 0.1 * mag ** 2 * comp)
        a = torch.stack([g * mag * torch.cos(ph), g * mag * torch.sin(ph)], dim=1)
    return a


def pa_add_noise(x: torch.Tensor, awgn: torch.Tensor, phase: torch.Tensor, noise_floor_db: float = -60.0,
                 phase_noise_deg: float = 0.5) -> torch.Tensor:
    """x [B, S, 2], awgn [B, S, 2] and phase [B, S] standard normal draws."""
    noise_std = torch.sqrt((x ** 2).mean() * (10 ** (noise_floor_db / 10)))
    n = awgn * noise_std
    if phase_noise_deg > 0:
        pn = phase * (phase_noise_deg * math.pi / 180)
        xc = torch.complex(x[..., 0], x[..., 1]) * torch.exp(1j * pn)
        x = torch.stack([xc.real, xc.imag], dim=-1)
    return x + n
