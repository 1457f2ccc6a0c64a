"""Training-step harness in sequence mode ([B, L+5, 2] -> [B, L, 2]).

Counterpart of the reference's train.py:257-351 (train_step) and of the two-stage schedule of
training_colab.ipynb (code cell 14), written for the sequence formulation the reference's modules
were designed for (utils/dataset.py:262-320); the reference's own train_step cannot run for
B > 1 (shape mismatch in its gradient penalty).  Optimiser settings follow train.py:206-218.
"""
from typing import Callable, Dict, Optional

import torch

from . import _lib as L

ADAM = dict(lr=1e-4, betas=(0.0, 0.9), weight_decay=1e-5)
GEN_SIZES = [960, 32, 512, 16, 32, 2]


def _gen_params(gen):
    return [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]


class Stage1Step:
    """Supervised pre-training step: loss = MSE(G(iq), target); Adam update of the generator.

    Forward, loss and backward run in one fused kernel (dpd_gen_mse_step): neither the generator
    output nor any activation is written to HBM.  `grad_sync` (optional) is called with the flat
    gradient buffer before the optimiser step (data-parallel all-reduce); `absmax_sync` with each
    QAT abs-max scalar between passes.
    """

    def __init__(self, generator, optimizer: Optional[torch.optim.Optimizer] = None,
                 grad_sync: Optional[Callable[[torch.Tensor], None]] = None,
                 absmax_sync: Optional[Callable[[torch.Tensor], None]] = None, impl: int = L.IMPL_AUTO):
        self.gen = generator
        self.params = _gen_params(generator)
        dev = self.params[0].device
        self.opt = optimizer or torch.optim.Adam(self.params, **ADAM)
        self.grad_sync, self.absmax_sync, self.impl = grad_sync, absmax_sync, impl
        self.wq = torch.empty(L.GEN_NPARAM, dtype=torch.float32, device=dev)
        self.absmax = torch.zeros(L.GEN_NSCALE, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(L.GEN_NPARAM + 2, dtype=torch.float32, device=dev)   # [.., loss, weight]
        self.scratch_out = None
        self._bind_grads()

    def _bind_grads(self):
        """Parameter .grad tensors are views of one flat buffer (one all-reduce, no packing copies)."""
        for p, g in zip(self.params, torch.split(self.grads[:L.GEN_NPARAM], GEN_SIZES)):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g.view_as(p)

    def __call__(self, iq: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        gen = self.gen
        self._bind_grads()
        qat = bool(getattr(gen, "qat_enabled", False))
        B, S = iq.shape[0], iq.shape[1]
        st = L.stream()
        L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in self.params), int(qat), L.fptr(self.wq),
               L.fptr(self.absmax), st)
        if qat:
            if self.scratch_out is None or self.scratch_out.shape[0] != B or self.scratch_out.shape[1] != S - 5:
                self.scratch_out = torch.empty(B, S - 5, 2, dtype=torch.float32, device=iq.device)
            for stage in range(4):
                L.call("dpd_gen_absmax_pass", stage, L.fptr(iq), B, S, 0, L.fptr(self.wq), float(gen.leaky_slope),
                       L.fptr(self.absmax), L.fptr(self.scratch_out), self.impl, st)
                if self.absmax_sync is not None:
                    self.absmax_sync(self.absmax[stage:stage + 1])
        _, wsp, wsn = L.workspace(iq.device)
        loss = self.grads[L.GEN_NPARAM:L.GEN_NPARAM + 1]
        L.call("dpd_gen_mse_step", L.fptr(iq), B, S, L.fptr(target), L.fptr(self.wq), float(gen.leaky_slope), int(qat),
               L.fptr(self.absmax), L.fptr(self.grads), L.fptr(loss), wsp, wsn, self.impl, st)
        if self.grad_sync is not None:
            self.grad_sync(self.grads)          # gradients and the loss in one buffer: one all-reduce
        if qat and gen.training:
            for i, name in enumerate(gen.ACT_QUANT + gen.WEIGHT_QUANT):
                getattr(gen, name).observe(self.absmax[i])
        self.opt.step()
        return loss
