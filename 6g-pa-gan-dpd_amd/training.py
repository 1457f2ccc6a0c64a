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
            # the last pass writes the generator output unless the MFMA family runs it (the library says which
            # family takes this shape): the step has no use for that output, so it is not written where it can
            # be left out
            mfma = bool(L.lib().dpd_gen_uses_mfma(B, S, 0, self.impl))
            outp = L.fptr(None)
            if not mfma:
                if self.scratch_out is None or self.scratch_out.shape[0] != B or self.scratch_out.shape[1] != S - 5:
                    self.scratch_out = torch.empty(B, S - 5, 2, dtype=torch.float32, device=iq.device)
                outp = L.fptr(self.scratch_out)
            for stage in range(4):
                L.call("dpd_gen_absmax_pass", stage, L.fptr(iq), B, S, 0, L.fptr(self.wq), float(gen.leaky_slope),
                       L.fptr(self.absmax), outp, self.impl, st)
                if self.absmax_sync is not None:
                    self.absmax_sync(self.absmax[stage:stage + 1])
        loss = self._core(iq, target, B, S, qat, st)
        if self.grad_sync is not None:
            self.grad_sync(self.grads)          # gradients and the loss in one buffer: one all-reduce
        if qat and gen.training:
            gen.observe_all(self.absmax)
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


class CriticUpdate:
    """One critic update of the CWGAN-GP step without autograd in between: the three critic evaluations of
    WassersteinLoss.discriminator_loss (real, fake, interpolate; each with its own spectral-norm power
    iteration, in that order, as in the reference's models/discriminator.py:204-244) write their gradients
    into rows of one buffer, the spectral-norm backward maps each row to the parameters, and the rows are
    combined into one flat gradient whose slices are the parameters' .grad (one all-reduce, one fused Adam).
    Same kernels and the same arithmetic per term as the autograd path, about a third of its launches."""

    SIZES = [256, 64, 2048, 32, 512, 16, 16, 1]      # W1, b1, ..., W4, b4: the layout of the flat operand

    def __init__(self, disc, opt, dp, gp_weight: float):
        self.disc, self.opt, self.dp, self.gp_weight = disc, opt, dp, float(gp_weight)
        self.layers = list(disc._layers())
        dev = self.layers[0].weight_orig.device
        n = L.CRITIC_NPARAM
        f = dict(dtype=torch.float32, device=dev)
        self.wb, self.uv, self.sigma = torch.empty(3, n, **f), torch.empty(3, 229, **f), torch.empty(3, 4, **f)
        self.gwb = torch.empty(3, n, **f)
        self.total = torch.zeros(n, **f)
        self.sums = torch.zeros(16, 4, **f)            # per update of a step: sum D(real), sum D(fake), gradient penalty, 0
        self.rownorm = None
        self.coef, self.coef_n = None, None

    @staticmethod
    def supported(disc) -> bool:
        from .models.discriminator import _SNLinearCore
        layers = list(disc._layers())
        return (disc.input_dim == 4 and list(disc.hidden_dims) == [64, 32, 16]
                and all(isinstance(lin, _SNLinearCore) for lin in layers) and layers[0].weight_orig.is_cuda
                and all(lin.weight_orig.dtype == torch.float32 for lin in layers))

    def _bind_grads(self):
        params = []
        for lin in self.layers:
            params += [lin.weight_orig, lin.bias]
        for p, g in zip(params, torch.split(self.total, self.SIZES)):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g.view_as(p)

    def __call__(self, real, fake, cond, alpha=None, slot: int = 0) -> torch.Tensor:
        """One update; the three sums it leaves go to row `slot` of self.sums (shares() turns the rows into losses)."""
        from .models.discriminator import _ptr_array
        disc, dp = self.disc, self.dp
        B = real.shape[0]
        Lrow = real.shape[1] if real.dim() == 3 else 1
        N = B * Lrow
        dev = real.device
        st = L.stream()
        ws = _ptr_array([lin.weight_orig for lin in self.layers])
        bs = _ptr_array([lin.bias for lin in self.layers])
        us = _ptr_array([lin.weight_u for lin in self.layers])
        vs = _ptr_array([lin.weight_v for lin in self.layers])
        eps, slope, power = float(self.layers[0].eps), float(disc.leaky_slope), int(disc.training)
        _, wsp, wsn = L.workspace(dev, N)
        if alpha is None:
            alpha = torch.rand(B, 1, device=dev)
        if self.rownorm is None or self.rownorm.numel() != B:
            self.rownorm = torch.empty(B, dtype=torch.float32, device=dev)
        if slot >= self.sums.shape[0]:
            self.sums = torch.zeros(2 * slot + 2, 4, dtype=torch.float32, device=dev)
        sums = self.sums[slot]
        cond_c = L.as_f32(cond)
        # the three evaluations' spectral normalisation (three successive power iterations, in the reference's order
        # real, fake, interpolate) in one launch: nothing between them touches the weights or u, v
        L.call("dpd_critic_sn_forward3", ws, bs, us, vs, power, eps, L.fptr(self.wb), L.fptr(self.uv), L.fptr(self.sigma), st)
        # d_loss = -(mean D(real) - mean D(fake)); every rank contributes its shard's mean / world
        for k, (x, sign) in enumerate(((real, -1.0), (fake, 1.0))):
            L.call("dpd_critic_backward", L.fptr(L.as_f32(x)), L.fptr(cond_c), N, L.fptr(self.wb[k]), slope, L.c_p(0),
                   sign / (N * dp.world), L.fptr(self.gwb[k]), L.c_p(0), L.fptr(sums[k:k + 1]), wsp, wsn, st)
        L.call("dpd_critic_gp", L.fptr(L.as_f32(real)), L.fptr(L.as_f32(fake)), L.fptr(cond_c),
               L.fptr(L.as_f32(alpha).reshape(-1)), B, Lrow, L.fptr(self.wb[2]), slope, L.fptr(sums[2:3]),
               L.fptr(self.rownorm), L.fptr(self.gwb[2]), wsp, wsn, st)
        # spectral-norm backward of the three evaluations and their combination (g_real + g_fake) + w * g_penalty
        L.call("dpd_critic_sn_backward3", ws, L.fptr(self.gwb), L.fptr(self.uv), L.fptr(self.sigma),
               self.gp_weight / dp.world, L.fptr(self.total), st)
        self._bind_grads()
        dp.sum_(self.total)
        self.opt.step()
        self._n = N
        return sums

    def shares(self, n_updates: int) -> torch.Tensor:
        """This rank's share of the critic loss of updates 0..n_updates-1, (sum D(fake) - sum D(real)) / N +
        gp_weight * gp, over the number of ranks (the caller sums the shares of all ranks, once per step): one
        matrix-vector product with the sums the updates left."""
        N, dp, dev = self._n, self.dp, self.sums.device
        if self.coef is None or self.coef_n != (N, dp.world):
            self.coef = torch.tensor([-1.0 / N, 1.0 / N, self.gp_weight, 0.0], dtype=torch.float32, device=dev) / dp.world
            self.coef_n = (N, dp.world)
        return torch.mv(self.sums[:n_updates], self.coef)


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
        # critic updates without autograd between the kernels where the critic has the shape they are built for
        self.critic_update = CriticUpdate(discriminator, self.d_opt, self.dp, gp_weight) if CriticUpdate.supported(discriminator) else None

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
        # the running-scale bookkeeping of the n_critic forwards the reference repeats (n_critic - 1 more in the critic
        # loop, one for the generator update): nothing reads the running scales in between, so all at once
        if hasattr(gen, "repeat_observation"):
            gen.repeat_observation(self.n_critic)
        fast = self.critic_update is not None and target.dim() == 3
        if alphas is None and fast and self.n_critic > 0:
            alphas = torch.rand(self.n_critic, iq.shape[0], 1, device=iq.device)      # one launch for the step's draws
        d_shares = []
        for it in range(self.n_critic):
            alpha = alphas[it] if alphas is not None else None
            if fast:
                self.critic_update(target, fake_d, cond, alpha, slot=it)
                continue
            self.d_opt.zero_grad(set_to_none=True)
            d_total, d_info = self._critic_loss(target, fake_d, cond, alpha)
            d_total.backward()
            dp.sum_grads_(d_params)
            self.d_opt.step()
            d_shares.append(d_total.detach())
        # the critic losses of the step are only reported: one all-reduce for all of them
        if self.n_critic > 0:
            d_all = dp.sum_(self.critic_update.shares(self.n_critic) if fast else torch.stack(d_shares))
            for it in range(self.n_critic):
                info[f"d_total_{it}"] = d_all[it]
        self.g_opt.zero_grad(set_to_none=True)
        # the critic is frozen for the generator update: its parameter gradients would be discarded
        for p in d_params:
            p.requires_grad_(False)
        try:
            g_total, g_info = self._generator_loss(fake, target, cond)
            g_total.backward()
        finally:
            for p in d_params:
                p.requires_grad_(True)
        dp.sum_grads_(gen.parameters())
        self.g_opt.step()
        info.update(g_info)
        return info
