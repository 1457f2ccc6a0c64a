"""Memory-tap feature assembly, vectorised (oracle; test infrastructure only).

Follows reference models/tdnn_generator.py:158-195 (MemoryTapAssembly.forward).  For output
sample n (n = M .. S-1) the feature vector is, for memory depth M:

    [ I(n), Q(n),
      e(n), e(n-1), ..., e(n-M),            e = sqrt(I^2 + Q^2)      (tdnn_generator.py:173)
      e(n)^2, ..., e(n-M)^2,                                          (:184)
      e(n)^4, ..., e(n-M)^4,                                          (:185)
      I(n-1), Q(n-1), ..., I(n-M), Q(n-M) ]                           (:189)

i.e. the envelope powers are stored in three blocks, not interleaved per tap.
"""
import torch


def feature_dim(memory_depth: int = 5) -> int:
    return 2 + 3 * (memory_depth + 1) + 2 * memory_depth


def tap_assembly(iq: torch.Tensor, memory_depth: int = 5) -> torch.Tensor:
    """iq [B, S, 2] -> features [B, S-M, 2 + 3(M+1) + 2M]; ValueError if S <= M (:169-170)."""
    B, S, _ = iq.shape
    M = memory_depth
    if S <= M:
        raise ValueError(f"Sequence length {S} must be > memory depth {M}")
    L = S - M
    env = torch.sqrt(iq[..., 0] ** 2 + iq[..., 1] ** 2)                   # [B, S]
    # column g of a window holds time n - g
    e = torch.stack([env[:, M - g:S - g] for g in range(M + 1)], dim=-1)  # [B, L, M+1]
    cur = iq[:, M:, :]                                                    # [B, L, 2]
    mem = torch.stack([iq[:, M - g:S - g, :] for g in range(1, M + 1)], dim=2).reshape(B, L, 2 * M)
    return torch.cat([cur, e, e ** 2, e ** 4, mem], dim=-1)
