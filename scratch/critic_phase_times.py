"""Per-phase cycle sums of an ABL_TIMING build of k_critic_bwd_mfma<true,false> (library given by DPD_HIP_LIBRARY)."""
import sys, torch
sys.path.insert(0, '.')
import dpd_amd
from dpd_amd import _lib as L
dev = torch.device('cuda:0'); torch.manual_seed(0)
N = 1 << 24
x = torch.randn(N, 2, device=dev) * 0.3; c = torch.randn(N, 2, device=dev) * 0.3
wb = torch.randn(2945, device=dev) * 0.2
grads = torch.empty(2945, device=dev); ssum = torch.zeros(1, device=dev)
ws, wsp, wsn = L.workspace(dev)
off = 2048 * 2952 * 4
def go():
    L.call("dpd_critic_backward", L.fptr(x), L.fptr(c), N, L.fptr(wb), 0.2, L.c_p(0), 1.0 / N, L.fptr(grads), L.c_p(0), L.fptr(ssum), wsp, wsn, L.stream())
go(); torch.cuda.synchronize()
ws[off:off + 160].zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); go(); e1.record(); torch.cuda.synchronize()
t = ws[off:off + 160].view(torch.int64).cpu().tolist()
ft = t[10:20]; t = t[:6]
tiles = N // 16
names = ["loop/prefetch", "forward (48 MFMA)", "masks + chain_down (40)", "outer p1 (16)", "outer p2 (32)", "outer p3 (8)"]
mf = [0, 48, 40, 16, 32, 8]
tot = sum(t)
print("kernel %.3f ms" % e0.elapsed_time(e1))
for n, v, m in zip(names, t, mf):
    print("%-26s %8.1f cycles/tile  %5.1f%%   fair share (2 waves) %d" % (n, v / tiles, 100 * v / tot, m * 64))
print("total %.1f cycles/tile per wave" % (tot / tiles))
fn = ["(before)", "b1 reads + FC1 issue (4)", "FC2 group 0 (8)", "FC2 group 1 (8)", "FC2 group 2 (8)", "FC2 group 3 (8) + pre", "h2 activation", "p1 MFMAs (16 small)", "FC3 (8) + pre", "h3, last layer, group sum"]
for n, v in zip(fn, ft):
    print("  forward: %-30s %8.1f cycles/tile" % (n, v / tiles))
