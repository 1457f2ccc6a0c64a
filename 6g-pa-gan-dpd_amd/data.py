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


THERMAL_AUGMENT = {"cold": (1.05, -0.02, 0.98), "normal": (0.0, 0.0, 0.0), "hot": (0.95, 0.03, 1.02)}   # gain shift, phase shift, compression


def _impair_ws(device, B, S):
    from . import _lib as L
    n = int(L.lib().dpd_impair_workspace_bytes(int(B), int(S)))
    return torch.empty(n, dtype=torch.uint8, device=device)


def _augment_launch(x: torch.Tensor, params: torch.Tensor, noise, seed: int) -> torch.Tensor:
    """x [B, S, 2] through dpd_augment_rows with per-row parameters [B, 8]."""
    from . import _lib as L
    L.need_gpu(x, params, noise)
    x = L.as_f32(x)
    B, S = x.shape[0], x.shape[1]
    out = torch.empty_like(x)
    ws = _impair_ws(x.device, B, S)
    L.call("dpd_augment_rows", L.fptr(x), B, S, L.fptr(L.as_f32(params)), L.fptr(None if noise is None else L.as_f32(noise)),
           int(seed) & 0xFFFFFFFFFFFFFFFF, L.fptr(out), L.c_p(ws.data_ptr()), L.c_sz(ws.numel()), L.stream())
    return out


def augment_signal(signal: torch.Tensor, augment_type: str = "all", strength: float = 0.1, draws: dict = None,
                   seed: int = 0) -> torch.Tensor:
    """The notebook's `augment_signal(signal [n, 2], augment_type, strength)` (training_colab.ipynb, data-loading
    cell) on the device, one kernel pass: additive noise at snr_db = 40 + 5 * draws['snr'] dB relative to the power of
    the whole signal, scaled by `strength`; a rotation by (draws['phase'] - 0.5) * 0.1 * strength rad; a gain of
    1 + (draws['gain'] - 0.5) * 0.2 * strength; and, by draws['thermal'] (< 0.33 cold, < 0.67 unchanged, else hot), the
    thermal stage gain_shift * (1 - 0.1 |v|^2 compression) with a phase shift.  `draws` holds the random numbers the
    notebook takes from torch's global generator (snr: one standard normal; phase, gain, thermal: one uniform each;
    noise: standard normals [n, 2], optional -- without it the noise comes from the kernel's own generator, keyed by
    `seed`); draws=None draws them here from a generator seeded with `seed`."""
    if augment_type not in ("noise", "phase", "gain", "thermal", "all"):
        raise ValueError(f"unknown augment_type {augment_type!r}")
    if draws is None:
        g = torch.Generator().manual_seed(int(seed))
        draws = dict(snr=float(torch.randn(1, generator=g)), phase=float(torch.rand(1, generator=g)),
                     gain=float(torch.rand(1, generator=g)), thermal=float(torch.rand(1, generator=g)))
    on = lambda kind: augment_type in (kind, "all")
    p = [0.0, 0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0]
    if on("noise"):
        p[0], p[1] = 1.0 / (10.0 ** ((40.0 + float(draws["snr"]) * 5.0) / 10.0)), float(strength)
    if on("phase"):
        off = torch.tensor((float(draws["phase"]) - 0.5) * 0.1 * strength)        # fp32, as the notebook forms it
        p[2], p[3] = float(torch.cos(off)), float(torch.sin(off))
    if on("gain"):
        p[4] = 1.0 + (float(draws["gain"]) - 0.5) * 0.2 * strength
    if on("thermal"):
        t = float(draws["thermal"])
        p[5], p[6], p[7] = THERMAL_AUGMENT["cold" if t < 0.33 else ("normal" if t < 0.67 else "hot")]
    params = torch.tensor([p], dtype=torch.float32, device=signal.device)
    noise = draws.get("noise") if on("noise") else None
    out = _augment_launch(signal.reshape(1, -1, 2), params, None if noise is None else noise.reshape(1, -1, 2), seed)
    return out.reshape(signal.shape)


def augment_rows(rows: torch.Tensor, strength: float = 0.1, seed: int = 0) -> torch.Tensor:
    """Training-time augmentation of IQ rows [B, S, 2] on the device: the four effects of the notebook's
    `augment_signal` (above) with one set of draws per ROW instead of one per call -- the per-row draws (B numbers
    each) come from a device generator seeded with `seed`, the noise itself from the kernel's counter-based generator,
    so the whole augmentation is one pass over the rows (8 B in, 8 B out per sample) with no host round trip.
    Parity with the notebook is exact for given draws (`augment_signal`); for the rows it is that of the distributions."""
    dev = rows.device
    g = torch.Generator(device=dev).manual_seed(int(seed))
    B = rows.shape[0]
    snr_db = 40.0 + 5.0 * torch.randn(B, generator=g, device=dev)
    phase = (torch.rand(B, generator=g, device=dev) - 0.5) * 0.1 * strength
    gain = 1.0 + (torch.rand(B, generator=g, device=dev) - 0.5) * 0.2 * strength
    state = torch.randint(0, 3, (B,), generator=g, device=dev)          # 0 = cold, 1 = unchanged, 2 = hot
    table = torch.tensor([THERMAL_AUGMENT["cold"], THERMAL_AUGMENT["normal"], THERMAL_AUGMENT["hot"]], device=dev)
    params = torch.cat([torch.stack([torch.pow(10.0, -snr_db / 10.0), torch.full_like(snr_db, float(strength)),
                                     torch.cos(phase), torch.sin(phase), gain], dim=1), table[state]], dim=1)
    return _augment_launch(rows, params.contiguous(), None, seed)
