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


TEMPERATURE_C = {"cold": -20.0, "normal": 25.0, "hot": 70.0}   # the reference's three training variants (train.py:497-513)


def apply_thermal_drift(y: torch.Tensor, temperature: float, reference_temp: float = 25.0) -> torch.Tensor:
    """Thermal drift of a PA output signal [..., 2] on the device: the model of the reference's
    apply_thermal_drift (train.py:92-126).  With dT = temperature - reference_temp: gain 1 - 0.005 dT / 10,
    phase shift 0.003 dT / 10 rad, AM/AM compression 1 - (0.01 dT / 50) |y|^2; y' = y * gain * compression * e^(j phase)."""
    dt = float(temperature) - float(reference_temp)
    gain = 1.0 + (-0.005) * (dt / 10.0)
    phase = 0.003 * (dt / 10.0)
    a_amam = 0.01 * (dt / 50.0)
    yc = torch.view_as_complex(y.to(torch.float32).contiguous())
    env2 = yc.real ** 2 + yc.imag ** 2
    out = yc * (gain * (1.0 - a_amam * env2)) * complex(math.cos(phase), math.sin(phase))
    return torch.view_as_real(out).contiguous()


def augment_rows(rows: torch.Tensor, strength: float = 0.1, seed: int = 0) -> torch.Tensor:
    """Training-time augmentation of IQ rows [B, S, 2] on the device, one random draw per row and effect.

    The effects and their ranges are those of the reference notebook's `augment_signal`
    (training_colab.ipynb, data-loading cell): additive white noise at an SNR ~ N(40, 5) dB scaled by
    `strength`; a phase offset of +-0.05 * strength rad; a gain of 1 +- 0.1 * strength; and a thermal
    drift (one third of the rows each: cold = gain 1.05, phase -0.02 rad, compression 0.98; unchanged;
    hot = 0.95, +0.03, 1.02) with AM-AM compression gain * (1 - 0.1 |x|^2 * compression).  The
    reference draws one set of random numbers per call on the host; here every row gets its own draw
    from a device generator (no host round trip), so parity is at the level of the distributions.
    """
    dev = rows.device
    g = torch.Generator(device=dev).manual_seed(seed)
    B = rows.shape[0]
    x = torch.view_as_complex(rows.contiguous())
    power = (rows ** 2).mean(dim=(1, 2), keepdim=False)
    snr_db = 40.0 + 5.0 * torch.randn(B, generator=g, device=dev)
    noise_std = torch.sqrt(power / torch.pow(10.0, snr_db / 10.0)) * strength
    noise = torch.randn(rows.shape, generator=g, device=dev) * noise_std.view(B, 1, 1)
    x = x + torch.view_as_complex(noise)
    phase = (torch.rand(B, generator=g, device=dev) - 0.5) * 0.1 * strength
    gain = 1.0 + (torch.rand(B, generator=g, device=dev) - 0.5) * 0.2 * strength
    x = x * torch.polar(gain, phase).view(B, 1)
    # thermal drift: 0 = cold, 1 = unchanged, 2 = hot
    state = torch.randint(0, 3, (B,), generator=g, device=dev)
    t_gain = torch.tensor([1.05, 1.0, 0.95], device=dev)[state]
    t_phase = torch.tensor([-0.02, 0.0, 0.03], device=dev)[state]
    t_comp = torch.tensor([0.98, 0.0, 1.02], device=dev)[state]
    mag2 = x.real ** 2 + x.imag ** 2
    am = t_gain.view(B, 1) * (1.0 - 0.1 * mag2 * t_comp.view(B, 1))
    x = x * am * torch.polar(torch.ones_like(t_phase), t_phase).view(B, 1)
    return torch.view_as_real(x).to(torch.float32).contiguous()
