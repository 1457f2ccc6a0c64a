"""Launch the fused Stage-1 kernel a few times (for counter collection)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in params), 0, L.fptr(wq), L.fptr(absmax), st)
_, wsp, wsn = L.workspace(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    L.call("dpd_gen_mse_step", L.fptr(iq), B, S, L.fptr(tgt), L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(grads), L.fptr(loss), wsp, wsn, 2, st)
torch.cuda.synchronize()
print(loss.item())
