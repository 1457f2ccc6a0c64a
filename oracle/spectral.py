"""EVM / ACPR / NMSE metrics and the combined spectral loss (oracle; test infrastructure only).

Follows reference utils/spectral_loss.py:
  * compute_evm   :24-64     sqrt(mean|m-r|^2 / clamp(mean|r|^2, 1e-10)); dB = 20 log10(. + 1e-10)
  * compute_acpr  :67-132    one full-length complex FFT per row, |X|^2/L, three inclusive masks
                             from fftfreq (boundary bins are counted in both bands)
  * compute_nmse  :135-167
  * SpectralLoss  :209-293   50*L1 + 20*clamp((mean EVM_dB+40)/40, 0, 2)
                             + 10*clamp((max(mean lo_dB, mean up_dB)+50)/50, 0, 2)
"""
from typing import Dict, Tuple

import torch


def _c(x):
    return torch.complex(x[..., 0], x[..., 1]) if (x.dim() == 3 and x.shape[-1] == 2) else x


def band_masks(L: int, sample_rate=200e6, channel_bw=100e6, adjacent_offset=100e6):
    f = torch.fft.fftfreq(L, d=1 / sample_rate)
    main = torch.abs(f) <= channel_bw / 2
    lower = (f >= -(adjacent_offset + channel_bw / 2)) & (f <= -(adjacent_offset - channel_bw / 2))
    upper = (f >= (adjacent_offset - channel_bw / 2)) & (f <= (adjacent_offset + channel_bw / 2))
    return main, lower, upper


def evm(measured, reference, db=True):
    err = _c(measured) - _c(reference)
    ep = (err.abs() ** 2).mean(dim=-1)
    rp = torch.clamp((_c(reference).abs() ** 2).mean(dim=-1), min=1e-10)
    lin = torch.sqrt(ep / rp)
    return 20 * torch.log10(lin + 1e-10) if db else lin


def acpr(signal, sample_rate=200e6, channel_bw=100e6, adjacent_offset=100e6, db=True):
    x = _c(signal)
    L = x.shape[-1]
    ps = (torch.fft.fft(x, dim=-1).abs() ** 2) / L
    main, lower, upper = band_masks(L, sample_rate, channel_bw, adjacent_offset)
    mp = torch.clamp((ps * main).sum(-1), min=1e-10)
    lo, up = (ps * lower).sum(-1) / mp, (ps * upper).sum(-1) / mp
    if db:
        lo, up = 10 * torch.log10(lo + 1e-10), 10 * torch.log10(up + 1e-10)
    return lo, up


def nmse(measured, reference, db=True):
    m = measured.reshape(measured.shape[0], -1)
    r = reference.reshape(reference.shape[0], -1)
    v = ((m - r) ** 2).sum(-1) / torch.clamp((r ** 2).sum(-1), min=1e-10)
    return 10 * torch.log10(v + 1e-10) if db else v


def spectral_loss(pred, target, sample_rate=200e6, channel_bw=100e6, adjacent_offset=100e6,
                  evm_weight=20.0, acpr_weight=10.0, l1_weight=50.0) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    l1 = (pred - target).abs().mean()
    evm_term = torch.clamp((evm(pred, target).mean() + 40) / 40, 0, 2)
    lo, up = acpr(pred, sample_rate, channel_bw, adjacent_offset)
    acpr_term = torch.clamp((torch.max(lo.mean(), up.mean()) + 50) / 50, 0, 2)
    total = l1_weight * l1 + evm_weight * evm_term + acpr_weight * acpr_term
    return total, dict(l1=l1, evm=evm_term, acpr=acpr_term)


def metrics(pred, target, sample_rate=200e6, channel_bw=100e6, adjacent_offset=100e6) -> Dict[str, float]:
    with torch.no_grad():
        lo, up = acpr(pred, sample_rate, channel_bw, adjacent_offset)
        m = dict(evm_db=evm(pred, target).mean().item(), nmse_db=nmse(pred, target).mean().item(),
                 acpr_lower_db=lo.mean().item(), acpr_upper_db=up.mean().item())
        m["acpr_max_db"] = max(m["acpr_lower_db"], m["acpr_upper_db"])
        m["l1_error"] = (pred - target).abs().mean().item()
        return m
