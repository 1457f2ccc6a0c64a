import sys, torch
sys.path.insert(0, '.')
import dpd_amd
from dpd_amd.data import ofdm_rows
from dpd_amd.models import TDNNGeneratorQAT
from dpd_amd.training import Stage1Step
dev = torch.device('cuda:0')
for sync in (True, False):
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(dev); gen.qat_enabled = False; gen.train()
    iq = ofdm_rows(16384, 1029, seed=1234, device=dev)
    tgt = iq[:, 5:, :].contiguous()
    step = Stage1Step(gen)
    hist = []
    for i in range(30):
        loss = step(iq, tgt)
        if sync:
            torch.cuda.synchronize()
        hist.append(loss.clone())
    torch.cuda.synchronize()
    print('sync', sync, [round(h.item(), 6) for h in hist])
