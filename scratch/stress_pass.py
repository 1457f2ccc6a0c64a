"""Repeatability of the QAT abs-max passes of the library named by DPD_HIP_LIBRARY: stress_pass.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpd_amd
from dpd_amd import _lib as L
from dpd_amd.data import ofdm_rows
from dpd_amd.models import TDNNGeneratorQAT
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
B, S = 8192, 1029
iq = ofdm_rows(B, S, seed=1234, device=dev)
torch.manual_seed(0)
gen = TDNNGeneratorQAT().to(dev).eval()
ps = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
wq = torch.empty(L.GEN_NPARAM, device=dev); absmax = torch.zeros(L.GEN_NSCALE, device=dev)
st = L.stream()
L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), 1, L.fptr(wq), L.fptr(absmax), st)
out = torch.empty(B, S - 5, 2, device=dev)
name = os.path.basename(os.environ.get("DPD_HIP_LIBRARY", "product"))
for stage in range(4):
    vals = []
    outs = []
    for r in range(reps):
        L.call("dpd_gen_absmax_pass", stage, L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.fptr(out), 0, st)
        vals.append(float(absmax[stage]))
        if stage == 3 and r < 8: outs.append(out.clone())
    print(name, "pass", stage, "distinct values:", sorted(set(vals))[:6], "count", len(set(vals)))
    if stage == 3:
        nb = [int((o != outs[0]).sum()) for o in outs[1:]]
        print(name, "pass 3 raw outputs differing from the first run:", nb)
