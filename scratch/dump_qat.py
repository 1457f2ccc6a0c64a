"""QAT generator outputs of the library named by DPD_HIP_LIBRARY into a file: dump_qat.py out.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpd_amd
from dpd_amd import _lib as L
from dpd_amd.data import ofdm_rows
from dpd_amd.models import PADigitalTwin, TDNNGeneratorQAT
dev = torch.device("cuda:0")
B, S = 2048, 1029
iq = ofdm_rows(B, S, seed=1234, device=dev)
torch.manual_seed(0)
gen = TDNNGeneratorQAT().to(dev).eval()
pa = PADigitalTwin().to(dev)
ps = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
wq = torch.empty(L.GEN_NPARAM, device=dev); absmax = torch.zeros(L.GEN_NSCALE, device=dev)
st = L.stream()
L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), 1, L.fptr(wq), L.fptr(absmax), st)
out = torch.empty(B, S - 5, 2, device=dev)
for stage in range(4):
    L.call("dpd_gen_absmax_pass", stage, L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.fptr(out), 0, st)
L.call("dpd_gen_forward", L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, 1, L.fptr(absmax), L.fptr(out), 2, st)
cre, cim = pa.pa_model.coeffs_real.contiguous(), pa.pa_model.coeffs_imag.contiguous()
yb = torch.empty(B, S - 9, 2, device=dev)
out2 = torch.empty(B, S - 5, 2, device=dev)
L.call("dpd_gen_pa_forward", L.fptr(iq), B, S, L.fptr(wq), 0.2, 1, L.fptr(absmax), L.fptr(cre), L.fptr(cim), 4, 5, L.fptr(out2), L.fptr(yb), st)
torch.cuda.synchronize()
torch.save(dict(absmax=absmax.cpu(), out=out.cpu(), out2=out2.cpu(), yb=yb.cpu()), sys.argv[1])
print("saved", sys.argv[1], absmax.cpu().tolist())
