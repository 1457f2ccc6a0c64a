"""Read the per-phase cycle sums of an ABL_TIMING build of the fused Stage-1 kernel."""
import sys, os, torch
sys.path.insert(0, '.')
import dpd_amd
from dpd_amd import _lib as L
from dpd_amd.data import ofdm_rows
from dpd_amd.models import TDNNGeneratorQAT
dev = torch.device('cuda:0'); torch.manual_seed(0)
gen = TDNNGeneratorQAT().to(dev)
B, S = 16384, 1029
iq = ofdm_rows(B, S, seed=1234, device=dev); tgt = iq[:, 5:, :].contiguous()
params = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
wq = torch.empty(1554, device=dev); absmax = torch.zeros(8, device=dev)
grads = torch.empty(1554, device=dev); loss = torch.zeros(1, device=dev)
st = L.stream()
L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in params), 0, L.fptr(wq), L.fptr(absmax), st)
ws, wsp, wsn = L.workspace(dev)
off = 1024 * 1560 * 4 + 64
def go():
    L.call("dpd_gen_mse_step", L.fptr(iq), B, S, L.fptr(tgt), L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(grads), L.fptr(loss), wsp, wsn, 2, st)
go(); torch.cuda.synchronize()
ws[off:off + 176].zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); go(); e1.record(); torch.cuda.synchronize()
print("step %.3f ms (kernel + slab reduce)" % e0.elapsed_time(e1))
t = ws[off:off + 176].view(torch.int64).cpu().tolist()
clk, real = t[20], t[21]; t = t[:20]
print('shader clock while the waves ran: %.3f GHz; wave alive %.3f ms on average' % (clk / real * 0.1, real / 2048 / 1e5))
pairs = B * 16 * 2   # tile pairs in the launch
names = ["prefetch+div", "F1", "S1A", "S1B+V1A", "S2A+V1B", "S2B+V2A", "V2B+V3+V4A", "S5A+V4B", "S5B+V5A", "S6A+V5B", "S6B", "-", "-", "-", "-", "-", "commit", "vmwait", "looptail", "-"]
tot = sum(t)
for n, v in zip(names, t):
    print("%-10s %8.1f cycles/pair  %5.1f%%" % (n, v / pairs, 100 * v / tot))
print("counter rate if the waves run the whole kernel: %.2f GHz" % (tot / (B * 16 * 2) * (B * 16 * 2 / 2048) / (e0.elapsed_time(e1) * 1e-3) / 1e9))
print("total %.1f cycles/pair (100 MHz counter units if s_memtime is the constant-rate clock)" % (tot / pairs))
