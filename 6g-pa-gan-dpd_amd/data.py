"""Synthetic OFDM-like IQ rows generated on the device (shape and spectrum of the benchmark input).

Mirrors the reference's own synthetic generators (utils/dataset.py:359-386 SyntheticDPDDataset,
training_colab.ipynb generate_ofdm_signal): an IFFT of random QPSK sub-carriers, peak-normalised.
Here: L-point IFFT with L/4 QPSK carriers centred on DC (inside the +-50 MHz main channel at
fs = 200 MHz), a 5-sample cyclic prefix so that every output sample has its 5 taps of history inside
its own row, each row scaled to peak |x| = 0.7.
"""
import math

import torch


def ofdm_rows(batch: int, seq_len: int, seed: int, device, peak: float = 0.7) -> torch.Tensor:
    """-> [batch, seq_len, 2] fp32 on `device`; seq_len = L + 5."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    L = seq_len - 5
    nsc = max(L // 4, 1)
    bits = torch.randint(0, 4, (batch, nsc), generator=g, device=dev)
    phase = math.pi / 4 + (math.pi / 2) * bits.to(torch.float32)
    spec = torch.zeros(batch, L, dtype=torch.complex64, device=dev)
    idx = (torch.arange(nsc, device=dev) - nsc // 2) % L
    spec[:, idx] = torch.polar(torch.ones_like(phase), phase)
    x = torch.fft.ifft(spec, dim=-1)
    x = torch.cat([x[:, -5:], x], dim=1)
    x = x / x.abs().amax(dim=1, keepdim=True) * peak
    return torch.view_as_real(x).to(torch.float32).contiguous()
