"""Training-step harness in sequence mode ([B, L+5, 2] -> [B, L, 2]).

Counterpart of the reference's train.py:257-351 (train_step) and of the two-stage schedule of
training_colab.ipynb (code cell 14), written for the sequence formulation the reference's modules
were designed for (utils/dataset.py:262-320); the reference's own train_step cannot run for
B > 1 (shape mismatch in its gradient penalty).  Optimiser settings follow train.py:206-218.
"""
from typing import Callable, Dict, Optional

import torch

from . import _lib as L

# fused: the whole update of all parameter tensors in one launch (same arithmetic as the reference's default Adam)
ADAM = dict(lr=1e-4, betas=(0.0, 0.9), weight_decay=1e-5, fused=True)
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

    def _core(self, iq, target, B, S, qat, st):
        _, wsp, wsn = L.workspace(iq.device)
        loss = self.grads[L.GEN_NPARAM:L.GEN_NPARAM + 1]
        L.call("dpd_gen_mse_step", L.fptr(iq), B, S, L.fptr(target), L.fptr(self.wq), float(self.gen.leaky_slope), int(qat),
               L.fptr(self.absmax), L.fptr(self.grads), L.fptr(loss), wsp, wsn, self.impl, st)
        return loss

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
        loss = self._core(iq, target, B, S, qat, st)
        if self.grad_sync is not None:
            self.grad_sync(self.grads)          # gradients and the loss in one buffer: one all-reduce
        if qat and gen.training:
            for i, name in enumerate(gen.ACT_QUANT + gen.WEIGHT_QUANT):
                getattr(gen, name).observe(self.absmax[i])
        self.opt.step()
        return loss


class ChainStep(Stage1Step):
    """Supervised step through the PA twin: loss = MSE(PA(G(iq)), target) with target [B, L-4, 2]; Adam update
    of the generator.  Generator forward, output quantiser, PA twin, loss, PA adjoint and generator backward run
    in one fused kernel (dpd_gen_pa_mse_step); the QAT abs-max passes precede it as in Stage1Step."""

    def __init__(self, generator, pa, **kw):
        super().__init__(generator, **kw)
        vpa = getattr(pa, "pa_model", pa)
        self.cre, self.cim = L.as_f32(vpa.coeffs_real), L.as_f32(vpa.coeffs_imag)

    def _core(self, iq, target, B, S, qat, st):
        K, M = self.cre.shape
        _, wsp, wsn = L.workspace(iq.device, B * (S - 5))
        loss = self.grads[L.GEN_NPARAM:L.GEN_NPARAM + 1]
        L.call("dpd_gen_pa_mse_step", L.fptr(iq), B, S, L.fptr(target), L.fptr(self.wq), float(self.gen.leaky_slope),
               int(qat), L.fptr(self.absmax), L.fptr(self.cre), L.fptr(self.cim), K, M, L.fptr(self.grads), L.fptr(loss),
               wsp, wsn, st)
        return loss


class Stage2Step:
    """CWGAN-GP step: n_critic critic updates (Wasserstein + gradient penalty), then one generator
    update on w_adv * adversarial + w_l1 * L1 + SpectralLoss (the combination of train.py:336-340).

    `alphas` (optional, [n_critic, B, 1]) fixes the interpolation coefficients; by default they are
    drawn on the device.  With a DataParallel context every loss term is formed from global sums
    and gradients are all-reduced once per backward (see parallel.py).
    """

    def __init__(self, generator, discriminator, g_opt=None, d_opt=None, n_critic: int = 5, gp_weight: float = 10.0,
                 w_adv: float = 1.0, w_l1: float = 50.0, spectral_loss=None, dp=None):
        from .models.discriminator import WassersteinLoss
        from .utils.spectral_loss import SpectralLoss
        from .parallel import DataParallel
        self.gen, self.disc = generator, discriminator
        self.g_opt = g_opt or torch.optim.Adam(generator.parameters(), **ADAM)
        self.d_opt = d_opt or torch.optim.Adam(discriminator.parameters(), **ADAM)
        self.n_critic, self.w_adv, self.w_l1 = n_critic, w_adv, w_l1
        self.wl = WassersteinLoss(gp_weight)
        self.sl = spectral_loss or SpectralLoss()
        self.dp = dp or DataParallel()
        if self.dp.enabled:
            generator.absmax_sync = self.dp.max_

    def _critic_loss(self, target, fake, cond, alpha):
        dev = target.device
        total, info = self.wl.discriminator_loss(self.disc, target, fake, cond, dev, alpha=alpha, sync_info=False)
        # means over the local shard; with equal shards the global mean is their average
        return total / self.dp.world, info

    def _generator_loss(self, fake, target, cond):
        from .utils.spectral_loss import spectral_total
        dp, sl = self.dp, self.sl
        g_adv, _ = self.wl.generator_loss(self.disc, fake, cond, sync_info=False)
        adv_g = dp.global_sum(g_adv / dp.world)

        def sum_fn(sums):                      # row sums and row count of all ranks -> terms of the global batch
            dp.sum_(sums)
        # the reconstruction term w_l1 * L1 rides on the spectral loss's own L1 (one pass over the samples)
        fused, l1_g, evm_loss, acpr_loss = spectral_total(fake, target, sl.sample_rate, sl.channel_bw, sl.adjacent_offset,
                                                         sl.l1_weight + self.w_l1, sl.evm_weight, sl.acpr_weight,
                                                         sum_fn if dp.enabled else None)
        total = self.w_adv * adv_g + fused
        spectral = fused.detach() - self.w_l1 * l1_g
        return total, dict(g_total=total.detach(), g_adv=adv_g.detach(), recon=l1_g, spectral=spectral,
                           evm=evm_loss, acpr=acpr_loss)

    def __call__(self, iq: torch.Tensor, target: torch.Tensor, cond: torch.Tensor, alphas=None) -> Dict[str, torch.Tensor]:
        gen, disc, dp = self.gen, self.disc, self.dp
        info = {}
        d_params = list(disc.parameters())
        # The generator does not change until its own update at the end of the step, so one forward serves
        # the whole step: its detached output feeds the critic updates, its autograd node the generator
        # update.  The reference recomputes an identical forward n_critic + 1 times; only the
        # running-scale bookkeeping of those forwards has to be repeated.
        fake = gen(iq)
        fake_d = fake.detach()
        for it in range(self.n_critic):
            self.d_opt.zero_grad(set_to_none=True)
            if it > 0 and hasattr(gen, "repeat_observation"):
                gen.repeat_observation()
            alpha = alphas[it] if alphas is not None else None
            d_total, d_info = self._critic_loss(target, fake_d, cond, alpha)
            d_total.backward()
            dp.sum_grads_(d_params)
            self.d_opt.step()
            info[f"d_total_{it}"] = dp.sum_(d_total.detach().clone())
        self.g_opt.zero_grad(set_to_none=True)
        # the critic is frozen for the generator update: its parameter gradients would be discarded
        for p in d_params:
            p.requires_grad_(False)
        try:
            if hasattr(gen, "repeat_observation"):
                gen.repeat_observation()
            g_total, g_info = self._generator_loss(fake, target, cond)
            g_total.backward()
        finally:
            for p in d_params:
                p.requires_grad_(True)
        dp.sum_grads_(gen.parameters())
        self.g_opt.step()
        info.update(g_info)
        return info
