"""Stage-1 (supervised MSE) and Stage-2 (CWGAN-GP) training steps in sequence mode (oracle; test infrastructure only).

Sequence formulation [B, L+5, 2] -> [B, L, 2] as used by reference utils/dataset.py:262-320.
Step structure follows reference train.py:257-351 (n_critic critic updates with
WassersteinLoss.discriminator_loss, then one generator update on
1.0*adv + 50*L1 + SpectralLoss, train.py:336-340) and the two-stage schedule of
training_colab.ipynb (code cell 14: Stage 1 is MSE only).  Optimiser: Adam(lr 1e-4,
betas (0, 0.9), weight_decay 1e-5), train.py:206-218.
"""
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

from . import critic as oc
from . import generator as og
from . import spectral as osp

ADAM = dict(lr=1e-4, betas=(0.0, 0.9), weight_decay=1e-5)


def new_running() -> Dict[str, torch.Tensor]:
    return {k: torch.tensor(1.0) for k in og.ACT_QUANT_KEYS + og.WEIGHT_QUANT_KEYS}


class GenState:
    def __init__(self, params: Dict[str, torch.Tensor]):
        self.p = {k: params[k].detach().clone().requires_grad_(True) for k in og.PARAM_KEYS}
        self.running = new_running()
        self.opt = torch.optim.Adam([self.p[k] for k in og.PARAM_KEYS], **ADAM)


class CriticState:
    def __init__(self, layers: List[Dict[str, torch.Tensor]]):
        self.layers = layers
        for l in self.layers:
            l["weight_orig"] = l["weight_orig"].detach().clone().requires_grad_(True)
            l["bias"] = l["bias"].detach().clone().requires_grad_(True)
        # parameter order of the reference module: per layer bias, weight_orig
        self.opt = torch.optim.Adam([t for l in self.layers for t in (l["bias"], l["weight_orig"])], **ADAM)


def stage1_step(g: GenState, iq, target, qat: bool) -> torch.Tensor:
    g.opt.zero_grad()
    out = og.gen_forward(iq, g.p, qat=qat, running=g.running if qat else None)
    loss = F.mse_loss(out, target)
    loss.backward()
    g.opt.step()
    return loss.detach()


def stage2_step(g: GenState, d: CriticState, iq, target, cond, alphas: Sequence[torch.Tensor], qat: bool = True,
                n_critic: int = 5, gp_weight: float = 10.0, w_adv: float = 1.0, w_l1: float = 50.0):
    info = {}
    for it in range(n_critic):
        d.opt.zero_grad()
        with torch.no_grad():
            fake = og.gen_forward(iq, g.p, qat=qat, running=g.running if qat else None)
        d_total, d_info = oc.discriminator_loss(d.layers, target, fake, cond, alphas[it], training=True, gp_weight=gp_weight)
        d_total.backward()
        d.opt.step()
        info[f"d_total_{it}"] = d_total.detach()
    g.opt.zero_grad()
    fake = og.gen_forward(iq, g.p, qat=qat, running=g.running if qat else None)
    g_adv = oc.generator_loss(d.layers, fake, cond, training=True)
    recon = (fake - target).abs().mean()
    spectral, comps = osp.spectral_loss(fake, target)
    g_total = w_adv * g_adv + w_l1 * recon + spectral
    g_total.backward()
    g.opt.step()
    info.update(g_total=g_total.detach(), g_adv=g_adv.detach(), recon=recon.detach(), spectral=spectral.detach(),
                evm=comps["evm"].detach(), acpr=comps["acpr"].detach())
    return info
