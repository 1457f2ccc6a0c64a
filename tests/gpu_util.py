"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch

from conftest import load_golden  # noqa: F401

DEV = "cuda:0"


def T(a, dev=DEV):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def sd(g, prefix):
    return {k[len(prefix):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def close_q(a, b, frac=0.95):
    """Fake-quantised tensors: every entry within one 16-bit step of the tensor's scale, and at least
    `frac` of them equal to fp32 rounding (the scale itself may differ in its last bit, so a
    bit-for-bit comparison is not meaningful across tanh / GEMM implementations)."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    lsb = np.abs(b).max() / 32767
    assert np.abs(a - b).max() <= lsb + 1e-5 * np.abs(b).max(), (np.abs(a - b).max(), lsb)  # one step + fp32 tolerance on the scale
    same = np.abs(a - b) <= 2e-6 * np.abs(b) + 1e-9
    assert same.mean() >= frac, same.mean()


def ofdm_rows(B, S, seed, peak=0.7):
    """Synthetic OFDM-like IQ rows [B, S, 2] (CPU tensor): QPSK on L/4 bins around DC, 5-sample prefix."""
    g = torch.Generator().manual_seed(seed)
    L = S - 5
    nsc = max(L // 4, 1)
    bits = torch.randint(0, 4, (B, nsc), generator=g)
    sym = torch.exp(1j * (np.pi / 4 + np.pi / 2 * bits.to(torch.float32)))
    spec = torch.zeros(B, L, dtype=torch.complex64)
    idx = (torch.arange(nsc) - nsc // 2) % L
    spec[:, idx] = sym.to(torch.complex64)
    x = torch.fft.ifft(spec, dim=-1)
    x = torch.cat([x[:, -5:], x], dim=1)
    x = x / x.abs().amax(dim=1, keepdim=True) * peak
    return torch.stack([x.real, x.imag], dim=-1).to(torch.float32).contiguous()
