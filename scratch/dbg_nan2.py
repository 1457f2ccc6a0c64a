import sys, torch
sys.path.insert(0, '.')
import dpd_amd
from dpd_amd.data import ofdm_rows
from dpd_amd.models import TDNNGeneratorQAT
from dpd_amd.training import Stage1Step
dev = torch.device('cuda:0')
for impl in (2, 1):
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(dev); gen.qat_enabled = False; gen.train()
    iq = ofdm_rows(16384, 1029, seed=1234, device=dev)
    tgt = iq[:, 5:, :].contiguous()
    step = Stage1Step(gen, impl=impl)
    for i in range(8):
        loss = step(iq, tgt)
        torch.cuda.synchronize()
        print(impl, i, loss.item(), step.grads[:1554].abs().max().item(), torch.isnan(step.grads).sum().item(), gen.fc1.weight.abs().max().item())
