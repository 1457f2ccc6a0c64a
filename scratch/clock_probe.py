"""Shader clock inside the fused Stage-1 kernel and the critic backward (library built with -DABL_CLOCK,
given by DPD_HIP_LIBRARY): every wave adds its shader-clock cycles and its 100 MHz ticks to two counters."""
import sys, torch
sys.path.insert(0, '.')
import dpd_amd
from dpd_amd import _lib as L
from dpd_amd.data import ofdm_rows
from dpd_amd.models import TDNNGeneratorQAT
dev = torch.device('cuda:0'); torch.manual_seed(0)
ws, wsp, wsn = L.workspace(dev)
st = L.stream()

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

# ---- generator
gen = TDNNGeneratorQAT().to(dev)
B, S = 16384, 1029
iq = ofdm_rows(B, S, seed=1234, device=dev); tgt = iq[:, 5:, :].contiguous()
params = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
wq = torch.empty(1554, device=dev); absmax = torch.zeros(8, device=dev)
grads = torch.empty(1554, device=dev); loss = torch.zeros(1, device=dev)
L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in params), 0, L.fptr(wq), L.fptr(absmax), st)
off = 1024 * 1560 * 4 + 64
def go():
    L.call("dpd_gen_mse_step", L.fptr(iq), B, S, L.fptr(tgt), L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(grads), L.fptr(loss), wsp, wsn, 2, st)
go(); torch.cuda.synchronize()
ws[off:off + 176].zero_()
reps = 20
ms = timed(go, reps)
t = ws[off:off + 176].view(torch.int64).cpu().tolist()
clk, real = t[20], t[21]
ghz = clk / real * 0.1
pairs_per_wave = B * 16 * 2 / 2048
cyc_pair = clk / (reps + 1) / 2048 / pairs_per_wave
print("generator Stage-1 step %.3f ms; shader clock %.3f GHz; %.0f cycles per tile pair per wave; MFMA pipe busy %.3f (112 MFMAs x 32 cycles x 2 waves)"
      % (ms, ghz, cyc_pair, 112 * 32 * 2 / cyc_pair))

# ---- critic
N = 1 << 24
x = torch.randn(N, 2, device=dev) * 0.3; c = torch.randn(N, 2, device=dev) * 0.3
wb = torch.randn(2945, device=dev) * 0.2
cg = torch.empty(2945, device=dev); ssum = torch.zeros(1, device=dev)
offc = 2048 * 2952 * 4
def goc():
    L.call("dpd_critic_backward", L.fptr(x), L.fptr(c), N, L.fptr(wb), 0.2, L.c_p(0), 1.0 / N, L.fptr(cg), L.c_p(0), L.fptr(ssum), wsp, wsn, st)
goc(); torch.cuda.synchronize()
ws[offc:offc + 80].zero_(); ws[offc + 72:offc + 80] = 127
ms = timed(goc, reps)
t = ws[offc:offc + 80].view(torch.int64).cpu().tolist()
print('critic wave alive: mean %.3f ms, max %.3f ms, min %.3f ms' % (t[7] / (reps + 1) / int(__import__("os").environ.get("CRITIC_WAVES", "2048")) / 1e5, t[8] / 1e5, t[9] / 1e5))
clk, real = t[6], t[7]
ghz = clk / real * 0.1
nwv = int(__import__("os").environ.get("CRITIC_WAVES", "2048")); cyc_tile = clk / (reps + 1) / nwv / (N / 16 / nwv)
print("critic backward %.3f ms; shader clock %.3f GHz; %.0f cycles per tile per wave; MFMA pipe busy %.3f (124 MFMAs x 32 + 16 x 8 cycles, x 2 waves)"
      % (ms, ghz, cyc_tile, (124 * 32 + 16 * 8) * 2 / cyc_tile))
