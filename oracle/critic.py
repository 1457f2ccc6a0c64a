"""Conditional spectral-norm critic 4->64->32->16->1 and WGAN-GP terms (oracle; test infrastructure only).

Follows reference models/discriminator.py:
  * critic forward          :103-136  cat([x, cond]) -> (SN-Linear + LeakyReLU 0.2) x3 -> SN-Linear
  * gradient penalty        :139-191  alpha per batch row, x_hat = a*real + (1-a)*fake,
                                      g = d sum D(x_hat, c) / d x_hat, mean_b (||g_b||_2 - 1)^2
  * Wasserstein losses      :204-267
Spectral normalisation is the legacy torch.nn.utils.spectral_norm hook used at :37-39: in
training mode every forward runs one power iteration in place on (u, v) without gradient
(v = normalize(W^T u), u = normalize(W v), eps 1e-12), then sigma = u . (W v) and
W_bar = W / sigma with u, v held constant for the gradient.

A critic is a list of four layer dicts {weight_orig, bias, u, v}; u/v are updated in place.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

LAYER_PREFIXES = ("features.0.linear", "features.2.linear", "features.4.linear", "output.linear")


def layers_from_state_dict(sd: Dict[str, torch.Tensor]) -> List[Dict[str, torch.Tensor]]:
    out = []
    for p in LAYER_PREFIXES:
        out.append(dict(weight_orig=sd[p + ".weight_orig"].clone(), bias=sd[p + ".bias"].clone(),
                        u=sd[p + ".weight_u"].clone(), v=sd[p + ".weight_v"].clone()))
    return out


def sn_weight(layer: Dict[str, torch.Tensor], power_iteration: bool, eps: float = 1e-12) -> torch.Tensor:
    w = layer["weight_orig"]
    if power_iteration:
        with torch.no_grad():
            layer["v"] = F.normalize(torch.mv(w.t(), layer["u"]), dim=0, eps=eps)
            layer["u"] = F.normalize(torch.mv(w, layer["v"]), dim=0, eps=eps)
    sigma = torch.dot(layer["u"], torch.mv(w, layer["v"]))
    return w / sigma


def critic_forward(layers, x: torch.Tensor, cond: torch.Tensor, training: bool, slope: float = 0.2,
                   spectral_norm: bool = True) -> torch.Tensor:
    """x, cond: [B, 2] or [B, L, 2] -> score [B, 1] or [B, L, 1]."""
    shape = None
    if x.dim() == 3:
        shape = x.shape[:2]
        x, cond = x.reshape(-1, 2), cond.reshape(-1, 2)
    h = torch.cat([x, cond], dim=-1)
    for i, layer in enumerate(layers):
        w = sn_weight(layer, training) if spectral_norm else layer["weight_orig"]
        h = F.linear(h, w, layer["bias"])
        if i + 1 < len(layers):
            h = F.leaky_relu(h, slope)
    return h.reshape(shape[0], shape[1], 1) if shape is not None else h


def gradient_penalty(layers, real, fake, cond, alpha: torch.Tensor, training: bool, slope: float = 0.2) -> torch.Tensor:
    """alpha [B, 1] is explicit (the reference draws torch.rand(B, 1) inside, :163)."""
    a = alpha.reshape(-1, 1, 1) if real.dim() == 3 else alpha.reshape(-1, 1)
    x_hat = (a * real + (1 - a) * fake).detach().requires_grad_(True)
    d = critic_forward(layers, x_hat, cond, training, slope)
    (g,) = torch.autograd.grad(d, x_hat, torch.ones_like(d), create_graph=True, retain_graph=True)
    g = g.reshape(g.shape[0], -1)
    return ((g.norm(2, dim=1) - 1) ** 2).mean()


def discriminator_loss(layers, real, fake, cond, alpha, training: bool, gp_weight: float = 10.0,
                       slope: float = 0.2) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    rs = critic_forward(layers, real, cond, training, slope)
    fs = critic_forward(layers, fake.detach(), cond, training, slope)
    w_distance = rs.mean() - fs.mean()
    gp = gradient_penalty(layers, real, fake, cond, alpha, training, slope)
    total = -w_distance + gp_weight * gp
    info = dict(d_loss=-w_distance, gp=gp, w_distance=w_distance, real_score=rs.mean(), fake_score=fs.mean())
    return total, info


def generator_loss(layers, fake, cond, training: bool, slope: float = 0.2) -> torch.Tensor:
    return -critic_forward(layers, fake, cond, training, slope).mean()
