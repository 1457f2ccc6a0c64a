"""Host-side mirror of the reference's utils/spectral_loss.py (compute_evm :24-64, compute_acpr
:67-132, compute_nmse :135-167, SpectralLoss :170-293, EVMLoss :296-310, ACPRLoss :313-339,
create_spectral_loss :342-354) on top of libdpd_hip.so.

All per-sample work (error sums, the row FFT, band powers, and their gradient through an inverse
transform) runs in csrc/spectral.hip and produces six statistics per row; the scalar arithmetic on
the [B] statistics (dB conversions, clamps, means, max) is plain torch on the device for the
stand-alone metrics and three small kernels for SpectralLoss itself (spectral_total).  The reference signal is treated as a
constant (no gradient), as in every use the reference makes of these functions.
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib as L

_mask_cache = {}


def band_masks(n: int, sample_rate: float, channel_bw: float, adjacent_offset: float, device) -> torch.Tensor:
    """uint8 [3, n] (main, lower, upper) from fp32 fftfreq exactly as the reference builds them
    (inclusive bounds, so boundary bins belong to two bands).  Host-side, cached per configuration."""
    key = (n, float(sample_rate), float(channel_bw), float(adjacent_offset), str(device))
    m = _mask_cache.get(key)
    if m is None:
        f = torch.fft.fftfreq(n, d=1 / sample_rate)
        main = torch.abs(f) <= channel_bw / 2
        lower = (f >= -(adjacent_offset + channel_bw / 2)) & (f <= -(adjacent_offset - channel_bw / 2))
        upper = (f >= (adjacent_offset - channel_bw / 2)) & (f <= (adjacent_offset + channel_bw / 2))
        m = torch.stack([main, lower, upper]).to(torch.uint8).contiguous().to(device)
        _mask_cache[key] = m
    return m


def _check_row_length(n: int, spectrum: bool) -> None:
    """Row lengths the kernels transform: powers of two in 2..4096 by FFT, any other length up to 8192 by a direct DFT
    (forward and gradient; e.g. the 4,087 PA outputs of a 4,096-sample slice)."""
    pow2 = not (n & (n - 1) or n < 2 or n > 4096)
    if spectrum and not pow2 and (n < 2 or n > 8192):
        raise NotImplementedError(f"ACPR on the HIP path needs a row length of a power of two in 2..4096 or any length up to 8192, got {n}")


class _RowStatsFn(torch.autograd.Function):
    """stats [B, 8] = (sum|p-t|, sum|p-t|^2, sum|t|^2, P_main, P_lower, P_upper, 0, 0) per row."""

    @staticmethod
    def forward(ctx, pred, target, band):
        L.need_gpu(pred, target, band)
        p = L.as_f32(pred)
        t = L.as_f32(target) if target is not None else None
        B, n = p.shape[0], p.shape[1]
        stats = torch.zeros(B, 8, dtype=torch.float32, device=p.device)
        _check_row_length(n, band is not None)
        L.call("dpd_spectral_stats", L.fptr(p), L.fptr(t), B, n, L.fptr(band, torch.uint8), L.fptr(stats), L.stream())
        ctx.save_for_backward(p, t if t is not None else p.new_empty(0), band if band is not None else p.new_empty(0))
        ctx.flags = (t is not None, band is not None)
        return stats

    @staticmethod
    def backward(ctx, gstats):
        p, t, band = ctx.saved_tensors
        has_t, has_band = ctx.flags
        B, n = p.shape[0], p.shape[1]
        dpred = torch.empty_like(p)
        L.call("dpd_spectral_backward", L.fptr(p), L.fptr(t if has_t else None), B, n,
               L.fptr(band if has_band else None, torch.uint8), L.fptr(L.as_f32(gstats)), L.fptr(dpred), L.stream())
        return dpred, None, None


class _SpectralTotalFn(torch.autograd.Function):
    """(total, l1, evm_term, acpr_term) of SpectralLoss.forward in five launches: row statistics, dB
    conversion + row sums, the scalar terms (forward); d total / d statistics, then the statistics'
    gradient through the inverse transform (backward).  `sum_fn(t)` (optional) all-reduces the
    six-entry sums tensor in place across data-parallel ranks, which makes every term that of the
    global batch.  Only `total` carries a gradient."""

    @staticmethod
    def forward(ctx, pred, target, band, w_l1, w_evm, w_acpr, sum_fn):
        L.need_gpu(pred, target, band)
        p, t = L.as_f32(pred), L.as_f32(target)
        B, n = p.shape[0], p.shape[1]
        _check_row_length(n, True)
        st = L.stream()
        stats = torch.empty(B, 8, dtype=torch.float32, device=p.device)
        sums = torch.empty(8, dtype=torch.float32, device=p.device)
        out = torch.empty(8, dtype=torch.float32, device=p.device)
        L.call("dpd_spectral_stats", L.fptr(p), L.fptr(t), B, n, L.fptr(band, torch.uint8), L.fptr(stats), st)
        L.call("dpd_spectral_rowdb", L.fptr(stats), B, n, L.c_p(0), L.fptr(sums), st)
        if sum_fn is not None:
            sum_fn(sums)
        L.call("dpd_spectral_loss_final", L.fptr(sums), n, float(w_l1), float(w_evm), float(w_acpr), L.fptr(out), st)
        ctx.save_for_backward(p, t, band, stats, out)
        total, l1, evm, acpr = out[0], out[1], out[2], out[3]
        ctx.mark_non_differentiable(l1, evm, acpr)
        return total, l1, evm, acpr

    @staticmethod
    def backward(ctx, g_total, _g1, _g2, _g3):
        p, t, band, stats, out = ctx.saved_tensors
        B, n = p.shape[0], p.shape[1]
        st = L.stream()
        gstats = torch.empty_like(stats)
        g = L.as_f32(g_total).reshape(1)
        L.call("dpd_spectral_terms_backward", L.fptr(stats), L.fptr(out[4:]), L.fptr(g), B, n, L.fptr(gstats), st)
        dpred = torch.empty_like(p)
        L.call("dpd_spectral_backward", L.fptr(p), L.fptr(t), B, n, L.fptr(band, torch.uint8), L.fptr(gstats),
               L.fptr(dpred), st)
        return dpred, None, None, None, None, None, None


def spectral_total(pred, target, sample_rate, channel_bw, adjacent_offset, w_l1, w_evm, w_acpr, sum_fn=None):
    """w_l1 * L1 + w_evm * EVM term + w_acpr * ACPR term and the three terms (0-d device tensors)."""
    p, t = _as_iq(pred), _as_iq(target)
    band = band_masks(p.shape[1], sample_rate, channel_bw, adjacent_offset, p.device)
    return _SpectralTotalFn.apply(p, t.detach(), band, w_l1, w_evm, w_acpr, sum_fn)


def _as_iq(x: torch.Tensor) -> torch.Tensor:
    """[B, L, 2] view of an IQ tensor given as [B, L, 2] or complex [B, L]."""
    if x.dim() == 3 and x.shape[-1] == 2:
        return x
    return torch.view_as_real(x.to(torch.complex64))


def compute_evm(measured: torch.Tensor, reference: torch.Tensor = None, return_db: bool = True) -> torch.Tensor:
    m, r = _as_iq(measured), _as_iq(reference)
    n = m.shape[1]
    st = _RowStatsFn.apply(m, r.detach(), None)
    ref_power = torch.clamp(st[:, 2] / n, min=1e-10)
    evm = torch.sqrt((st[:, 1] / n) / ref_power)
    return 20 * torch.log10(evm + 1e-10) if return_db else evm


def compute_acpr(signal: torch.Tensor, sample_rate: float = 200e6, channel_bw: float = 100e6,
                 adjacent_offset: float = 100e6, return_db: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    s = _as_iq(signal)
    band = band_masks(s.shape[1], sample_rate, channel_bw, adjacent_offset, s.device)
    st = _RowStatsFn.apply(s, None, band)
    main = torch.clamp(st[:, 3], min=1e-10)
    lo, up = st[:, 4] / main, st[:, 5] / main
    if return_db:
        lo, up = 10 * torch.log10(lo + 1e-10), 10 * torch.log10(up + 1e-10)
    return lo, up


def compute_nmse(measured: torch.Tensor, reference: torch.Tensor, return_db: bool = True) -> torch.Tensor:
    m = measured.reshape(measured.shape[0], -1)
    r = reference.reshape(reference.shape[0], -1)
    if m.shape[1] % 2:
        # the kernel sums over IQ pairs: a zero appended to both rows adds nothing to either sum
        m = torch.nn.functional.pad(m, (0, 1))
        r = torch.nn.functional.pad(r, (0, 1))
    st = _RowStatsFn.apply(m.reshape(m.shape[0], -1, 2), r.detach().reshape(r.shape[0], -1, 2), None)
    nmse = st[:, 1] / torch.clamp(st[:, 2], min=1e-10)
    return 10 * torch.log10(nmse + 1e-10) if return_db else nmse


def _loss_terms(pred, target, sample_rate, channel_bw, adjacent_offset):
    """One statistics pass shared by all terms: (l1, evm_db [B], acpr_lower_db [B], acpr_upper_db [B], nmse_db [B])."""
    p, t = _as_iq(pred), _as_iq(target)
    B, n = p.shape[0], p.shape[1]
    band = band_masks(n, sample_rate, channel_bw, adjacent_offset, p.device)
    st = _RowStatsFn.apply(p, t.detach(), band)
    l1 = st[:, 0].sum() / (B * n * 2)
    ref_power = torch.clamp(st[:, 2] / n, min=1e-10)
    evm_db = 20 * torch.log10(torch.sqrt((st[:, 1] / n) / ref_power) + 1e-10)
    main = torch.clamp(st[:, 3], min=1e-10)
    lo_db = 10 * torch.log10(st[:, 4] / main + 1e-10)
    up_db = 10 * torch.log10(st[:, 5] / main + 1e-10)
    nmse_db = 10 * torch.log10(st[:, 1] / torch.clamp(st[:, 2], min=1e-10) + 1e-10)
    return l1, evm_db, lo_db, up_db, nmse_db


class SpectralLoss(nn.Module):
    """l1_weight * L1 + evm_weight * clamp((mean EVM_dB + 40) / 40, 0, 2)
    + acpr_weight * clamp((max(mean ACPR_lower_dB, mean ACPR_upper_dB) + 50) / 50, 0, 2)."""

    def __init__(self, sample_rate: float = 200e6, channel_bw: float = 100e6, adjacent_offset: float = 100e6,
                 evm_weight: float = 20.0, acpr_weight: float = 10.0, l1_weight: float = 50.0):
        super().__init__()
        self.sample_rate, self.channel_bw, self.adjacent_offset = sample_rate, channel_bw, adjacent_offset
        self.evm_weight, self.acpr_weight, self.l1_weight = evm_weight, acpr_weight, l1_weight
        self.l1_loss = nn.L1Loss()

    def forward(self, predicted: torch.Tensor, target: torch.Tensor, return_components: bool = False):
        total, l1, evm_loss, acpr_loss = spectral_total(predicted, target, self.sample_rate, self.channel_bw,
                                                        self.adjacent_offset, self.l1_weight, self.evm_weight,
                                                        self.acpr_weight)
        if return_components:
            return total, {"l1": l1, "evm": evm_loss, "acpr": acpr_loss}
        return total

    def metrics_tensor(self, predicted: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """[evm_db, nmse_db, acpr_lower_db, acpr_upper_db, l1_error] as one device tensor (no host synchronisation)."""
        with torch.no_grad():
            l1, evm_db, lo_db, up_db, nmse_db = _loss_terms(predicted, target, self.sample_rate, self.channel_bw,
                                                           self.adjacent_offset)
            return torch.stack([evm_db.mean(), nmse_db.mean(), lo_db.mean(), up_db.mean(), l1])

    def compute_metrics(self, predicted: torch.Tensor, target: torch.Tensor) -> Dict[str, float]:
        vals = self.metrics_tensor(predicted, target).tolist()  # one sync
        m = dict(evm_db=vals[0], nmse_db=vals[1], acpr_lower_db=vals[2], acpr_upper_db=vals[3])
        m["acpr_max_db"] = max(m["acpr_lower_db"], m["acpr_upper_db"])
        m["l1_error"] = vals[4]
        return m


class EVMLoss(nn.Module):
    def __init__(self, target_evm_db: float = -35.0):
        super().__init__()
        self.target_evm_db = target_evm_db

    def forward(self, predicted: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return torch.relu(compute_evm(predicted, target, return_db=True).mean() - self.target_evm_db)


class ACPRLoss(nn.Module):
    def __init__(self, sample_rate: float = 200e6, channel_bw: float = 100e6, adjacent_offset: float = 100e6,
                 target_acpr_db: float = -45.0):
        super().__init__()
        self.sample_rate, self.channel_bw, self.adjacent_offset, self.target_acpr_db = sample_rate, channel_bw, adjacent_offset, target_acpr_db

    def forward(self, signal: torch.Tensor) -> torch.Tensor:
        lo, up = compute_acpr(signal, self.sample_rate, self.channel_bw, self.adjacent_offset)
        return torch.relu(torch.max(lo.mean(), up.mean()) - self.target_acpr_db)


def create_spectral_loss(config: dict) -> SpectralLoss:
    system = config.get("system", {})
    loss = config.get("training", {}).get("loss", {})
    fs = float(system.get("sample_rate", 200e6))          # YAML gives "200e6" as a string
    return SpectralLoss(sample_rate=fs, channel_bw=fs / 2, adjacent_offset=fs / 2,
                        evm_weight=float(loss.get("spectral_evm", 20.0)), acpr_weight=float(loss.get("spectral_acpr", 10.0)),
                        l1_weight=float(loss.get("reconstruction_l1", 50.0)))
