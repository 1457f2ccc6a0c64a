"""Loop-for-loop restatement of the two Python loops that dominate the reference on a CPU (oracle; test
infrastructure only -- it exists to be timed as the "literal algorithm" CPU baseline and to be checked
against the vectorised oracle).

  * tap assembly: one Python iteration per output sample, as reference
    models/tdnn_generator.py:176-193 (for n in range(M, S): gather current IQ, the M+1 envelope taps,
    their 2nd and 4th powers, the M previous IQ pairs; concatenate);
  * memory polynomial: the triple loop over output sample, order and tap of reference
    models/pa_digital_twin.py:157-165 (y[:, n] += h[k, m] * x[:, n+M-1-m] * |x[:, n+M-1-m]|^(k-1)).

Both give the same values as oracle/taps.py and oracle/pa.py (tests/test_oracle_goldens.py).
"""
import torch


def tap_assembly_loop(iq: torch.Tensor, memory_depth: int = 5) -> torch.Tensor:
    B, S, _ = iq.shape
    M = memory_depth
    if S <= M:
        raise ValueError(f"Sequence length {S} must be > memory depth {M}")
    env = torch.sqrt(iq[..., 0] ** 2 + iq[..., 1] ** 2)
    rows = []
    for n in range(M, S):
        cur = iq[:, n, :]
        e = torch.flip(env[:, n - M:n + 1], dims=[1])                 # e(n), e(n-1), ..., e(n-M)
        mem = torch.flip(iq[:, n - M:n, :], dims=[1]).reshape(B, -1)  # I(n-1), Q(n-1), ..., I(n-M), Q(n-M)
        rows.append(torch.cat([cur, e, e ** 2, e ** 4, mem], dim=-1))
    return torch.stack(rows, dim=1)


def volterra_forward_loop(x: torch.Tensor, coef_re: torch.Tensor, coef_im: torch.Tensor) -> torch.Tensor:
    xc = torch.complex(x[..., 0], x[..., 1])
    B, S = xc.shape
    K, M = coef_re.shape
    if S < M:
        raise ValueError(f"Sequence length {S} must be >= memory depth {M}")
    out_len = S - M + 1
    h = torch.complex(coef_re, coef_im)
    y = torch.zeros(B, out_len, dtype=torch.complex64)
    for n in range(out_len):
        for ki in range(K):
            k = 2 * ki + 1
            for m in range(M):
                xm = xc[:, n + M - 1 - m]
                y[:, n] += h[ki, m] * xm * (xm.abs() ** (k - 1))
    return torch.stack([y.real, y.imag], dim=-1)
