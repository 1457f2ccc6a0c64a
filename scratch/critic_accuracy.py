"""Error of the MFMA critic kernels of a library build (DPD_HIP_LIBRARY) against an fp64 evaluation on the CPU:
scores, input gradient and flat weight gradient of dpd_critic_backward for N samples."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpd_amd
from dpd_amd import _lib as L
dev = torch.device('cuda:0'); torch.manual_seed(0)
N = 1 << 16
x = (torch.randn(N, 2) * 0.5); c = (torch.randn(N, 2) * 0.5); gs = torch.randn(N)
sizes = [(64, 4), (32, 64), (16, 32), (1, 16)]
Ws = [torch.randn(o, i) * (1.5 / i ** 0.5) for o, i in sizes]; bs = [torch.randn(o) * 0.2 for o, _ in sizes]
flat = torch.cat([t.reshape(-1) for W, b in zip(Ws, bs) for t in (W, b)])
assert flat.numel() == 2945
# fp64 reference
xd = x.double().requires_grad_(True)
Wd = [W.double().requires_grad_(True) for W in Ws]; bd = [b.double().requires_grad_(True) for b in bs]
h = torch.cat([xd, c.double()], 1)
for k in range(4):
    h = h @ Wd[k].T + bd[k]
    if k < 3: h = torch.nn.functional.leaky_relu(h, 0.2)
score = h[:, 0]
(score * gs.double()).sum().backward()
ref_g = torch.cat([t.grad.reshape(-1) for W, b in zip(Wd, bd) for t in (W, b)])
xg, cg, gsg, wb = x.to(dev).contiguous(), c.to(dev).contiguous(), gs.to(dev).contiguous(), flat.to(dev)
out = torch.empty(N, device=dev); grads = torch.empty(2945, device=dev); gx = torch.empty(N, 2, device=dev)
ws, wsp, wsn = L.workspace(dev); st = L.stream()
L.call("dpd_set_option", b"critic_impl", 2)
L.call("dpd_critic_forward", L.fptr(xg), L.fptr(cg), N, L.fptr(wb), 0.2, L.fptr(out), L.c_p(0), st)
L.call("dpd_critic_backward", L.fptr(xg), L.fptr(cg), N, L.fptr(wb), 0.2, L.fptr(gsg), 0.0, L.fptr(grads), L.fptr(gx), L.c_p(0), wsp, wsn, st)
torch.cuda.synchronize()
def err(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
name = os.path.basename(os.environ.get("DPD_HIP_LIBRARY", "default"))
print("%-16s score %.2e   input gradient %.2e   weight gradient %.2e   (max error / max magnitude, against fp64)"
      % (name, err(out, score.detach()), err(gx, xd.grad), err(grads, ref_g)))
