"""Time the fused Stage-1 kernel (dpd_gen_mse_step) of the library named by DPD_HIP_LIBRARY."""
import sys, os, torch
sys.path.insert(0, '.')
import dpd_amd
from dpd_amd import _lib as L
from dpd_amd.data import ofdm_rows
from dpd_amd.models import TDNNGeneratorQAT
dev = torch.device('cuda:0')
torch.manual_seed(0)
gen = TDNNGeneratorQAT().to(dev)
B, S = 16384, 1029
iq = ofdm_rows(B, S, seed=1234, device=dev)
tgt = iq[:, 5:, :].contiguous()
params = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
wq = torch.empty(1554, device=dev); absmax = torch.zeros(8, device=dev)
grads = torch.empty(1554, device=dev); loss = torch.zeros(1, device=dev)
st = L.stream()
qat = int(os.environ.get("QAT", "0"))
L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in params), 0, L.fptr(wq), L.fptr(absmax), st)
_, wsp, wsn = L.workspace(dev)
def go():
    L.call("dpd_gen_mse_step", L.fptr(iq), B, S, L.fptr(tgt), L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(grads), L.fptr(loss), wsp, wsn, 2, st)
for _ in range(3): go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n): go()
e1.record(); torch.cuda.synchronize()
print("%-28s %.4f ms  loss %.6f" % (os.path.basename(os.environ.get("DPD_HIP_LIBRARY", "default")), e0.elapsed_time(e1) / n, loss.item()))
