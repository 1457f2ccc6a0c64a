"""Helpers shared by the GPU parity tests."""
import os

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
    if os.environ.get("DPD_CLOSE_REPORT"):
        # how much of the allowance a comparison uses: max |a - b| / (atol + rtol |b|), with the caller's line
        import inspect
        f = inspect.stack()[1]
        d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
        use = float(np.max(d / np.maximum(atol + rtol * np.abs(b), 1e-300))) if d.size else 0.0
        print(f"close: {os.path.basename(f.filename)}:{f.lineno} used {use:.4f} of rtol={rtol:g} atol={atol:g} "
              f"max|d|={float(d.max()) if d.size else 0:.3e} max|b|={float(np.abs(b).max()) if d.size else 0:.3e}")
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def close_q(a, b, frac=None):
    """Fake-quantised tensors: every entry within one 16-bit step of the tensor's scale, and all but a few
    equal to fp32 rounding.  A pre-activation that differs in its last bits between two GEMM summation
    orders (or a tanh that differs by an ulp) can land on the other side of a rounding boundary and move an
    entry by one whole step; measured on this suite: at most 3.1 % of a 128-entry tensor (4 entries), 0.4-1.1 %
    of larger ones.  Allowed: 2 % of the entries, or 5 entries for small tensors.  The scales themselves are
    compared with the oracle's separately (test_qat_scales_match_oracle)."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    lsb = np.abs(b).max() / 32767
    assert np.abs(a - b).max() <= lsb + 1e-5 * np.abs(b).max(), (np.abs(a - b).max(), lsb)  # one step + fp32 tolerance on the scale
    same = np.abs(a - b) <= 2e-6 * np.abs(b) + 1e-9
    if os.environ.get("DPD_CLOSEQ_REPORT"):
        print(f"close_q: n={a.size} equal fraction {same.mean():.6f} max|d|/lsb {np.abs(a - b).max() / lsb:.4f}")
    allowed = max(5, int(np.ceil(0.02 * a.size))) if frac is None else int(np.floor((1 - frac) * a.size))
    assert (~same).sum() <= allowed, ((~same).sum(), a.size)


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
