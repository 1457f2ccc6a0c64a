"""Time generator forward kernels (float forward; QAT passes 1..3) of the library named by DPD_HIP_LIBRARY."""
import sys, os, torch
sys.path.insert(0, '.')
import dpd_amd
from dpd_amd import _lib as L
from dpd_amd.data import ofdm_rows
from dpd_amd.models import TDNNGeneratorQAT
dev = torch.device('cuda:0'); torch.manual_seed(0)
gen = TDNNGeneratorQAT().to(dev)
B, S = 16384, 1029
iq = ofdm_rows(B, S, seed=1234, device=dev)
params = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
wq = torch.empty(1554, device=dev); absmax = torch.zeros(8, device=dev)
out = torch.empty(B, S - 5, 2, device=dev)
st = L.stream()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
name = os.path.basename(os.environ.get("DPD_HIP_LIBRARY", "default"))
L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in params), 0, L.fptr(wq), L.fptr(absmax), st)
t = timeit(lambda: L.call("dpd_gen_forward", L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(out), 2, st))
print("%-20s float forward %.4f ms  sum %.6f" % (name, t, out.double().sum().item()))
L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in params), 1, L.fptr(wq), L.fptr(absmax), st)
for stage in range(4):
    t = timeit(lambda: L.call("dpd_gen_absmax_pass", stage, L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.fptr(out), 2, st))
    print("%-20s qat pass %d     %.4f ms  absmax %s" % (name, stage, t, absmax[:4].tolist()))
