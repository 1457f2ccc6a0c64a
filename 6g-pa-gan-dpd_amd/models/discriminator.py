"""Host-side mirror of the reference's models/discriminator.py (SpectralNormLinear :28-42,
Discriminator :45-136, compute_gradient_penalty :139-191, WassersteinLoss :194-267,
create_discriminator :270-281) on top of libdpd_hip.so.

The critic MLP, its backward and the gradient penalty (with its double backward) run in the HIP
kernels of csrc/critic.hip and csrc/critic_mfma.hip.  Spectral normalisation (one power iteration per
training-mode forward on 64x4 .. 1x16 matrices, sigma, W / sigma and the backward to `weight_orig`) is one
launch each way in csrc/critic_sn.hip; nothing of the critic is computed by torch ops.
State-dict keys match the reference: features.{0,2,4}.linear.{bias,weight_orig,weight_u,weight_v}
and output.linear.*.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L

_SIZES = [256, 64, 2048, 32, 512, 16, 16, 1]


class _SNLinearCore(nn.Module):
    """Parameters and buffers of a legacy-hook spectral-norm Linear, with the same RNG consumption at
    construction (Linear init, then u ~ N(0,1)^out and v ~ N(0,1)^in, normalised)."""

    def __init__(self, in_features: int, out_features: int, eps: float = 1e-12):
        super().__init__()
        lin = nn.Linear(in_features, out_features)
        self.in_features, self.out_features, self.eps = in_features, out_features, eps
        self.bias = nn.Parameter(lin.bias.detach().clone())
        self.weight_orig = nn.Parameter(lin.weight.detach().clone())
        with torch.no_grad():
            u = F.normalize(lin.weight.new_empty(out_features).normal_(0, 1), dim=0, eps=eps)
            v = F.normalize(lin.weight.new_empty(in_features).normal_(0, 1), dim=0, eps=eps)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _SNFlatFn(torch.autograd.Function):
    """wb[2945] = [W1/sigma1 | b1 | ... | W4/sigma4 | b4] for four spectrally normalised layers: one
    launch forward (power iteration in place on the u, v buffers when `power_iteration`), one backward."""

    @staticmethod
    def forward(ctx, power_iteration, eps, us, vs, *wbs):
        ws, bs = [L.as_f32(t) for t in wbs[0::2]], [L.as_f32(t) for t in wbs[1::2]]
        L.need_gpu(*ws)
        dev = ws[0].device
        wb = torch.empty(L.CRITIC_NPARAM, dtype=torch.float32, device=dev)
        uv = torch.empty(229, dtype=torch.float32, device=dev)
        sigma = torch.empty(4, dtype=torch.float32, device=dev)
        L.call("dpd_critic_sn_forward", _ptr_array(ws), _ptr_array(bs), _ptr_array(us), _ptr_array(vs),
               int(power_iteration), float(eps), L.fptr(wb), L.fptr(uv), L.fptr(sigma), L.stream())
        ctx.save_for_backward(uv, sigma, *ws)
        return wb

    @staticmethod
    def backward(ctx, gwb):
        uv, sigma, *ws = ctx.saved_tensors
        g = torch.empty(L.CRITIC_NPARAM, dtype=torch.float32, device=uv.device)
        L.call("dpd_critic_sn_backward", _ptr_array(ws), L.fptr(L.as_f32(gwb)), L.fptr(uv), L.fptr(sigma), L.fptr(g),
               L.stream())
        out, off = [], 0
        for n in _SIZES:
            out.append(g[off:off + n])
            off += n
        shaped = [o.view_as(ws[i // 2]) if i % 2 == 0 else o for i, o in enumerate(out)]
        return (None, None, None, None, *shaped)


class SpectralNormLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        if not bias:
            raise NotImplementedError("the HIP critic expects biased layers (the reference default)")
        self.linear = _SNLinearCore(in_features, out_features)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # The critic never calls its layers one by one: Discriminator.forward / mean_score run the four layers fused
        # in the HIP kernels.  A single spectral-norm layer on its own has no kernel here, and this package has no
        # torch fallback.
        raise NotImplementedError("SpectralNormLinear runs only as part of Discriminator (the four layers are fused in "
                                  "the HIP critic kernels); a stand-alone layer is not provided")


class _CriticFn(torch.autograd.Function):
    """score[N] = critic(x[N,2], cond[N,2]; wb[2945]) and its backward w.r.t. wb and x."""

    @staticmethod
    def forward(ctx, x, cond, wb, slope):
        L.need_gpu(x, cond, wb)
        xc, cc, wbc = L.as_f32(x), L.as_f32(cond), L.as_f32(wb)
        N = xc.shape[0]
        score = torch.empty(N, dtype=torch.float32, device=xc.device)
        L.call("dpd_critic_forward", L.fptr(xc), L.fptr(cc), N, L.fptr(wbc), float(slope), L.fptr(score), L.c_p(0),
               L.stream())
        ctx.save_for_backward(xc, cc, wbc)
        ctx.slope = float(slope)
        ctx.need_x = x.requires_grad
        return score

    @staticmethod
    def backward(ctx, gscore):
        xc, cc, wbc = ctx.saved_tensors
        N = xc.shape[0]
        need_w = ctx.needs_input_grad[2]
        if not (need_w or ctx.need_x):
            return None, None, None, None
        grads = torch.empty(L.CRITIC_NPARAM, dtype=torch.float32, device=xc.device) if need_w else None
        gx = torch.empty_like(xc) if ctx.need_x else None
        _, wsp, wsn = L.workspace(xc.device)
        L.call("dpd_critic_backward", L.fptr(xc), L.fptr(cc), N, L.fptr(wbc), ctx.slope, L.fptr(L.as_f32(gscore)), 0.0,
               L.fptr(grads), L.fptr(gx), L.c_p(0), wsp, wsn, L.stream())
        return gx, None, grads, None


class _CriticMeanFn(torch.autograd.Function):
    """mean_n critic(x, cond; wb) with its gradients formed in the same kernel pass: the mean is linear in
    the scores, so the upstream gradient of every score is the constant 1/N and the backward of this
    function only scales the saved gradients."""

    @staticmethod
    def forward(ctx, x, cond, wb, slope):
        L.need_gpu(x, cond, wb)
        xc, cc, wbc = L.as_f32(x), L.as_f32(cond), L.as_f32(wb)
        N = xc.shape[0]
        need_w, need_x = wb.requires_grad, x.requires_grad
        ssum = torch.zeros(1, dtype=torch.float32, device=xc.device)
        if not (need_w or need_x):
            L.call("dpd_critic_forward", L.fptr(xc), L.fptr(cc), N, L.fptr(wbc), float(slope), L.c_p(0), L.fptr(ssum),
                   L.stream())
            return ssum[0] / N
        grads = torch.empty(L.CRITIC_NPARAM, dtype=torch.float32, device=xc.device) if need_w else None
        gx = torch.empty_like(xc) if need_x else None
        _, wsp, wsn = L.workspace(xc.device)
        L.call("dpd_critic_backward", L.fptr(xc), L.fptr(cc), N, L.fptr(wbc), float(slope), L.c_p(0), 1.0 / N,
               L.fptr(grads), L.fptr(gx), L.fptr(ssum), wsp, wsn, L.stream())
        ctx.save_for_backward(*(t for t in (grads, gx) if t is not None))
        ctx.flags = (need_w, need_x)
        return ssum[0] / N

    @staticmethod
    def backward(ctx, g):
        need_w, need_x = ctx.flags
        saved = list(ctx.saved_tensors)
        grads = saved.pop(0) if need_w else None
        gx = saved.pop(0) if need_x else None
        return (gx * g if need_x else None), None, (grads * g if need_w else None), None


class _GradPenaltyFn(torch.autograd.Function):
    """gp = mean_b (||d sum D(x_hat)/d x_hat||_2 - 1)^2 with d gp / d wb computed in the same call."""

    @staticmethod
    def forward(ctx, real, fake, cond, alpha, wb, slope, B, Lrow):
        L.need_gpu(real, fake, cond, alpha, wb)
        dev = real.device
        gp = torch.zeros(1, dtype=torch.float32, device=dev)
        rownorm = torch.empty(B, dtype=torch.float32, device=dev)
        grads = torch.empty(L.CRITIC_NPARAM, dtype=torch.float32, device=dev)
        _, wsp, wsn = L.workspace(dev, B * Lrow)
        L.call("dpd_critic_gp", L.fptr(L.as_f32(real)), L.fptr(L.as_f32(fake)), L.fptr(L.as_f32(cond)),
               L.fptr(L.as_f32(alpha).reshape(-1)), B, Lrow, L.fptr(L.as_f32(wb)), float(slope), L.fptr(gp),
               L.fptr(rownorm), L.fptr(grads), wsp, wsn, L.stream())
        ctx.save_for_backward(grads)
        return gp[0]

    @staticmethod
    def backward(ctx, g):
        (grads,) = ctx.saved_tensors
        # the penalty is piecewise constant in real / fake / cond (masks only), so those get no gradient
        return None, None, None, None, grads * g, None, None, None


class Discriminator(nn.Module):
    """Conditional critic: cat([x, condition]) -> (Linear + LeakyReLU) x3 -> Linear -> score."""

    def __init__(self, input_dim: int = 4, hidden_dims: List[int] = [64, 32, 16], leaky_slope: float = 0.2,
                 use_spectral_norm: bool = True):
        super().__init__()
        self.input_dim, self.hidden_dims, self.leaky_slope, self.use_spectral_norm = input_dim, hidden_dims, leaky_slope, use_spectral_norm
        layers, in_dim = [], input_dim
        for h in hidden_dims:
            layers.append(SpectralNormLinear(in_dim, h) if use_spectral_norm else nn.Linear(in_dim, h))
            layers.append(nn.LeakyReLU(negative_slope=leaky_slope))
            in_dim = h
        self.features = nn.Sequential(*layers)
        self.output = SpectralNormLinear(hidden_dims[-1], 1) if use_spectral_norm else nn.Linear(hidden_dims[-1], 1)
        for m in self.modules():
            if isinstance(m, _SNLinearCore):
                nn.init.kaiming_normal_(m.weight_orig, a=self.leaky_slope)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight, a=self.leaky_slope)
                nn.init.zeros_(m.bias)

    def _layers(self):
        mods = [self.features[i] for i in range(0, len(self.features), 2)] + [self.output]
        return [m.linear if isinstance(m, SpectralNormLinear) else m for m in mods]

    def flat_weights(self) -> torch.Tensor:
        """[W1, b1, ..., W4, b4] flattened, W spectrally normalised (one power iteration per layer
        when training); differentiable w.r.t. weight_orig / bias."""
        if self.input_dim != 4 or list(self.hidden_dims) != [64, 32, 16]:
            raise NotImplementedError("the HIP critic is built for 4 -> 64 -> 32 -> 16 -> 1 (the reference default)")
        layers = list(self._layers())
        if all(isinstance(lin, _SNLinearCore) for lin in layers):
            L.need_gpu(layers[0].weight_orig)       # the spectral normalisation is a HIP kernel: no CPU path
            args = []
            for lin in layers:
                args += [lin.weight_orig, lin.bias]
            return _SNFlatFn.apply(self.training, layers[0].eps, [lin.weight_u for lin in layers],
                                   [lin.weight_v for lin in layers], *args)
        # use_spectral_norm=False: plain Linear layers, whose parameters are only packed into the flat operand
        L.need_gpu(layers[0].weight)
        parts = []
        for lin in layers:
            parts += [lin.weight.reshape(-1), lin.bias]
        return torch.cat(parts)

    def mean_score(self, x: torch.Tensor, condition: torch.Tensor) -> torch.Tensor:
        """mean over all samples of forward(x, condition), fused with its backward (one kernel pass).
        Counts as one forward for the spectral-norm power iteration."""
        if x.dim() == 3:
            x, condition = x.reshape(-1, 2), condition.reshape(-1, 2)
        return _CriticMeanFn.apply(x, condition, self.flat_weights(), self.leaky_slope)

    def forward(self, x: torch.Tensor, condition: torch.Tensor) -> torch.Tensor:
        shape = None
        if x.dim() == 3:
            shape = x.shape[:2]
            x, condition = x.reshape(-1, 2), condition.reshape(-1, 2)
        score = _CriticFn.apply(x, condition, self.flat_weights(), self.leaky_slope).unsqueeze(-1)
        return score.reshape(shape[0], shape[1], 1) if shape is not None else score


def compute_gradient_penalty(discriminator: Discriminator, real_samples: torch.Tensor, fake_samples: torch.Tensor,
                             condition: torch.Tensor, device: torch.device,
                             alpha: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Same call as the reference; `alpha` ([B, 1], optional) fixes the interpolation coefficients,
    otherwise they are drawn with torch.rand(B, 1, device=device) as the reference does."""
    B = real_samples.size(0)
    if alpha is None:
        alpha = torch.rand(B, 1, device=device)
    Lrow = real_samples.shape[1] if real_samples.dim() == 3 else 1
    wb = discriminator.flat_weights()       # one power iteration, as one critic forward in the reference
    return _GradPenaltyFn.apply(real_samples.detach(), fake_samples.detach(), condition.detach(), alpha, wb,
                                discriminator.leaky_slope, B, Lrow)


class WassersteinLoss:
    """Critic loss E[D(fake)] - E[D(real)] + gp_weight * GP and generator loss -E[D(fake)]."""

    def __init__(self, gp_weight: float = 10.0):
        self.gp_weight = gp_weight

    def discriminator_loss(self, discriminator, real_samples, fake_samples, condition, device,
                           alpha: Optional[torch.Tensor] = None, sync_info: bool = True) -> Tuple[torch.Tensor, dict]:
        real_mean = discriminator.mean_score(real_samples, condition)
        fake_mean = discriminator.mean_score(fake_samples.detach(), condition)
        w_distance = real_mean - fake_mean
        d_loss = -w_distance
        gp = compute_gradient_penalty(discriminator, real_samples, fake_samples, condition, device, alpha=alpha)
        total = d_loss + self.gp_weight * gp
        vals = dict(d_loss=d_loss, gp=gp, w_distance=w_distance, real_score=real_mean, fake_score=fake_mean)
        # the reference returns python floats (one host sync each); sync_info=False keeps device scalars
        info = {k: (v.item() if sync_info else v.detach()) for k, v in vals.items()}
        return total, info

    def generator_loss(self, discriminator, fake_samples, condition, sync_info: bool = True) -> Tuple[torch.Tensor, dict]:
        fake_mean = discriminator.mean_score(fake_samples, condition)
        g_loss = -fake_mean
        vals = dict(g_loss=g_loss, fake_score=fake_mean)
        return g_loss, {k: (v.item() if sync_info else v.detach()) for k, v in vals.items()}


def create_discriminator(config: dict) -> Discriminator:
    d = config["model"]["discriminator"]
    return Discriminator(input_dim=d.get("input_dim", 4), hidden_dims=d.get("hidden_dims", [64, 32, 16]),
                         leaky_slope=d.get("leaky_slope", 0.2), use_spectral_norm=d.get("use_spectral_norm", True))
