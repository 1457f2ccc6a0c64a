"""Single-node data parallelism for the DPD training steps: one process per GPU, torch.distributed
over RCCL (backend "nccl" on ROCm) or gloo in CPU tests.

The hot path shards by IQ row: every sample's 5 taps of history live in its own row, the row FFT
is per row, so ranks never exchange samples.  What crosses the xGMI links per optimiser step is
tiny (generator 1,554 + critic 2,945 fp32 gradients, a handful of scalars), i.e. latency-bound:
the design minimises the NUMBER of collectives (one flat buffer per backward) rather than bytes.

Exact-global semantics (results equal to one process on the concatenated batch):
  * QAT activation scales are global abs-max reductions  -> all_reduce(MAX) between passes;
  * losses that are non-linear in batch means (clamped EVM / ACPR terms) are formed from
    all-reduced per-row sums via GlobalSum, whose backward is the identity;
  * every loss term is then a function of global sums, so gradients are all-reduced with SUM.
"""
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def _all_reduce(t: torch.Tensor, op, group=None) -> torch.Tensor:
    """all_reduce in place.  RCCL (backend "nccl") reduces device tensors directly.  Under gloo (CPU tests and the
    one-GPU rehearsal of the multi-rank path) a device tensor is reduced through a host copy, and the payloads here
    are a few kilobytes.

    Why: handed device tensors, gloo stalls when the work that produces the tensor is still in flight on the stream.
    Two ranks on one GPU, Stage-2 steps, DPD_GLOO_DEVICE_TENSORS=1: after a few dozen collectives both ranks sit in
    the SAME dist.all_reduce (the gradient all-reduce of the critic update; stack dump of both ranks after 45 s), so
    the ranks agree on the order of collectives and the wait is inside gloo's device path.  The same run with a
    torch.cuda.synchronize() in front of every all_reduce (DPD_GLOO_DEVICE_TENSORS=2) completes.  The host copy below
    contains that synchronisation, so it is the fix and not a cover: nothing of ours is left in flight when gloo gets
    the data.  RCCL orders its kernels on the stream itself and is not affected."""
    mode = os.environ.get("DPD_GLOO_DEVICE_TENSORS", "0")     # experiment switch: 1 = device tensors to gloo, 2 = after a synchronise
    if t.is_cuda and mode == "2":
        torch.cuda.synchronize()
    if t.is_cuda and dist.get_backend(group) == "gloo" and mode == "0":
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


class _GlobalSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        y = x.clone()
        _all_reduce(y, dist.ReduceOp.SUM, group)
        return y

    @staticmethod
    def backward(ctx, g):
        # every rank evaluates the same downstream loss on the global sum; the contribution of this
        # rank's addend to it has derivative 1.  (Gradients are summed over ranks afterwards.)
        return g, None


class DataParallel:
    """Thin wrapper over a process group; with world == 1 every method is a no-op."""

    def __init__(self, group=None):
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.group = group
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0

    # ---- sharding -------------------------------------------------------------------------
    def shard_rows(self, n_rows: int):
        """Contiguous block of rows for this rank (the last ranks get one row less when uneven)."""
        base, rem = divmod(n_rows, self.world)
        start = self.rank * base + min(self.rank, rem)
        return start, start + base + (1 if self.rank < rem else 0)

    # ---- collectives ----------------------------------------------------------------------
    def max_(self, t: torch.Tensor) -> torch.Tensor:
        if self.enabled:
            _all_reduce(t, dist.ReduceOp.MAX, self.group)
        return t

    def sum_(self, t: torch.Tensor) -> torch.Tensor:
        if self.enabled:
            _all_reduce(t, dist.ReduceOp.SUM, self.group)
        return t

    def mean_(self, t: torch.Tensor) -> torch.Tensor:
        if self.enabled:
            _all_reduce(t, dist.ReduceOp.SUM, self.group)
            t.div_(self.world)
        return t

    def global_sum(self, x: torch.Tensor) -> torch.Tensor:
        """Differentiable all-reduce(SUM) (see module docstring)."""
        return _GlobalSumFn.apply(x, self.group) if self.enabled else x

    def sum_grads_(self, params: Iterable[torch.nn.Parameter]) -> None:
        """One all-reduce(SUM) over the gradients of `params`, packed into a flat buffer."""
        if not self.enabled:
            return
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        _all_reduce(flat, dist.ReduceOp.SUM, self.group)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
