"""Repeatability stress at the bench shape: every generator kernel that issues bf16 MFMAs, N launches each, results compared
bit for bit with the first launch (training kernels in deterministic mode).  stress_all.py [N]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpd_amd  # noqa
from dpd_amd import _lib as L
from dpd_amd.data import ofdm_rows
from dpd_amd.models import PADigitalTwin, TDNNGeneratorQAT
from dpd_amd.training import ChainStep, Stage1Step
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
B, S = 16384, 1029
iq = ofdm_rows(B, S, seed=5, device=dev)
tgt = iq[:, 5:, :].contiguous()
tgt_pa = (tgt[:, 4:, :] * 31.0).contiguous()
pa = PADigitalTwin().to(dev)
cre, cim = pa.pa_model.coeffs_real.contiguous(), pa.pa_model.coeffs_imag.contiguous()
st = L.stream()
bad = 0
for qat in (False, True):
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(dev).train()
    gen.qat_enabled = qat
    ps = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
    wq = torch.empty(L.GEN_NPARAM, device=dev)
    absmax = torch.zeros(L.GEN_NSCALE, device=dev)
    L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), int(qat), L.fptr(wq), L.fptr(absmax), st)
    out, out2 = torch.empty(B, S - 5, 2, device=dev), torch.empty(B, S - 5, 2, device=dev)
    yb = torch.empty(B, S - 9, 2, device=dev)
    if qat:
        for stage in range(4):
            L.call("dpd_gen_absmax_pass", stage, L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.fptr(out), 0, st)
    first = None
    for it in range(n):
        L.call("dpd_gen_forward", L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, int(qat), L.fptr(absmax), L.fptr(out), 2, st)
        L.call("dpd_gen_pa_forward", L.fptr(iq), B, S, L.fptr(wq), 0.2, int(qat), L.fptr(absmax), L.fptr(cre), L.fptr(cim), 4, 5,
               L.fptr(out2), L.fptr(yb), st)
        if first is None:
            first = (out.clone(), yb.clone())
        d = int((out != first[0]).sum()) + int((out2 != first[0]).sum()) + int((yb != first[1]).sum())
        if d:
            bad += 1
            print("qat", qat, "forward launch", it, "differs in", d, "values", flush=True)
    L.call("dpd_set_option", b"gen_deterministic", 1)
    for name, make, target in (("stage1", lambda: Stage1Step(gen, optimizer=torch.optim.SGD(gen.parameters(), lr=0.0)), tgt),
                               ("chain", lambda: ChainStep(gen, pa, optimizer=torch.optim.SGD(gen.parameters(), lr=0.0)), tgt_pa)):
        step = make()
        ref = None
        for it in range(n):
            step(iq, target)
            g = step.grads[:L.GEN_NPARAM + 1].clone()
            if ref is None:
                ref = g
            if not torch.equal(g, ref):
                bad += 1
                print("qat", qat, name, "launch", it, "differs:", float((g - ref).abs().max()), flush=True)
    L.call("dpd_set_option", b"gen_deterministic", 0)
    print("qat", qat, "done", flush=True)
print("launch sets with differences:", bad)
sys.exit(1 if bad else 0)
