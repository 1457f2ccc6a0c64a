import sys, torch
sys.path.insert(0, '.')
import dpd_amd
from dpd_amd.data import ofdm_rows
from dpd_amd.models import TDNNGeneratorQAT
from dpd_amd.training import Stage1Step
dev = torch.device('cuda:0')
for impl in (1, 2):
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(dev); gen.qat_enabled = False; gen.train()
    iq = ofdm_rows(16384, 1029, seed=1234, device=dev)
    tgt = iq[:, 5:, :].contiguous()
    step = Stage1Step(gen, impl=impl)
    for i in range(11):
        pn = sum(torch.isnan(p).sum().item() for p in gen.parameters())
        loss = step(iq, tgt)
        torch.cuda.synchronize()
        if i >= 6:
            g = step.grads
            print(impl, i, 'loss', loss.item(), 'params nan before', pn, 'grad nan', torch.isnan(g[:1554]).sum().item(),
                  'first nan idx', (torch.isnan(g[:1554]).nonzero()[:5].flatten().tolist()), 'out nan', torch.isnan(gen(iq)).sum().item())
