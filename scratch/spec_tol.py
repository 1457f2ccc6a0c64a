"""Measured error of the SpectralLoss gradient against the oracle and against an fp64 evaluation (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpd_amd
from dpd_amd.utils import SpectralLoss
from oracle import spectral as osp
for n, rows in ((256, 7), (1024, 9), (4096, 3), (64, 3)):
    gen = torch.Generator().manual_seed(n)
    tgt = torch.randn(rows, n, 2, generator=gen) * 0.3
    pred = (tgt + 0.05 * torch.randn(rows, n, 2, generator=gen))
    p32 = pred.clone().requires_grad_(True)
    ref, _ = osp.spectral_loss(p32, tgt)
    (g32,) = torch.autograd.grad(ref, p32)
    p64 = pred.double().requires_grad_(True)
    ref64, _ = osp.spectral_loss(p64, tgt.double())
    (g64,) = torch.autograd.grad(ref64, p64)
    p = pred.to("cuda").requires_grad_(True)
    SpectralLoss()(p, tgt.to("cuda")).backward()
    s = g64.abs().max()
    print(n, "hip vs fp64 %.2e  oracle32 vs fp64 %.2e  hip vs oracle32 %.2e (of max |g|)" % (
        (p.grad.cpu().double() - g64).abs().max() / s, (g32.double() - g64).abs().max() / s, (p.grad.cpu() - g32).abs().max() / s))
