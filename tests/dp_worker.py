"""One rank of the two-rank data-parallel rehearsal (started by test_gpu_dp_two_ranks.py; also runs alone with
WORLD_SIZE unset as the single-process comparison).  Every rank uses GPU 0 and the gloo backend: what is under test
is the N > 1 code path of the training steps (row sharding, MAX of the QAT scales between passes, global sums under the
clamped loss terms, one flat gradient all-reduce per optimiser step), not the transport.
usage: dp_worker.py <result.pt>"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ROWS, SEQ, N_CRITIC = 32, 261, 5      # 4096 outputs per rank: the MFMA kernel families on every rank


def main(out_path):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import dpd_amd  # noqa: F401
    from dpd_amd.data import ofdm_rows
    from dpd_amd.models import Discriminator, TDNNGeneratorQAT
    from dpd_amd.parallel import DataParallel
    from dpd_amd.training import Stage1Step, Stage2Step

    dp = DataParallel()
    assert dp.world == world
    # the whole batch is generated identically everywhere; a rank keeps its block of rows
    iq_all = ofdm_rows(ROWS, SEQ, seed=5, device=dev)
    tgt_all = (iq_all[:, 5:, :] * 0.9).contiguous()
    alphas_all = torch.rand(N_CRITIC, ROWS, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    lo, hi = dp.shard_rows(ROWS)
    iq, tgt, alphas = iq_all[lo:hi].contiguous(), tgt_all[lo:hi].contiguous(), alphas_all[:, lo:hi].contiguous()
    res = {}

    # ---- Stage 1, QAT on
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(dev).train()
    with torch.no_grad():
        for fc in (gen.fc1, gen.fc2, gen.fc3):
            fc.bias.uniform_(-0.1, 0.1)
    s1 = Stage1Step(gen, grad_sync=dp.mean_ if dp.enabled else None, absmax_sync=dp.max_ if dp.enabled else None)
    loss = s1(iq, tgt)
    res["s1_loss"] = loss.detach().cpu().clone()
    res["s1_absmax"] = s1.absmax[:7].cpu().clone()
    res["s1_grads"] = s1.grads[:1554].cpu().clone()
    res["s1_params"] = torch.cat([p.detach().reshape(-1) for p in s1.params]).cpu()
    res["s1_running"] = torch.stack([getattr(gen, n).running_scale for n in gen.ACT_QUANT + gen.WEIGHT_QUANT]).cpu()

    # ---- Stage 2 (fresh models), QAT on, fixed interpolation coefficients
    torch.manual_seed(0)
    gen2 = TDNNGeneratorQAT().to(dev).train()
    torch.manual_seed(1)
    disc = Discriminator().to(dev).train()
    s2 = Stage2Step(gen2, disc, dp=dp)
    info = s2(iq, tgt, tgt, alphas=alphas)
    for k in ("g_total", "g_adv", "recon", "evm", "acpr") + tuple(f"d_total_{i}" for i in range(N_CRITIC)):
        res["s2_" + k] = info[k].detach().cpu().clone()
    res["s2_critic_grads"] = s2.critic_update.total.cpu().clone()
    res["s2_gen_grads"] = torch.cat([p.grad.reshape(-1) for p in gen2.parameters()]).cpu()
    res["s2_gen_params"] = torch.cat([p.detach().reshape(-1) for p in gen2.parameters()]).cpu()
    res["s2_critic_params"] = torch.cat([p.detach().reshape(-1) for p in disc.parameters()]).cpu()
    res["s2_u"] = torch.cat([lin.weight_u for lin in disc._layers()]).cpu()
    torch.cuda.synchronize()
    if dp.rank == 0:
        torch.save(res, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
