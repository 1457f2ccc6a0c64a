"""Two data-parallel ranks (child processes, both on GPU 0, gloo) against one process on the whole batch: Stage-1
step with QAT on and the CWGAN-GP step, HIP kernels underneath.  Losses, QAT scales, gradients and updated parameters
of the two-rank run must equal the single-process run on the concatenated batch."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(world, out):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        if world > 1:
            env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=150)
            logs.append(o)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, logs):
        assert p.returncode == 0, o[-3000:]
    return torch.load(out, weights_only=True)


def test_two_ranks_equal_one_process_on_the_concatenated_batch(tmp_path):
    one = _run(1, str(tmp_path / "one.pt"))
    two = _run(2, str(tmp_path / "two.pt"))

    def close(key, rtol, atol=0.0):
        a, b = two[key].double(), one[key].double()
        assert a.shape == b.shape, key
        err = (a - b).abs()
        assert bool((err <= atol + rtol * b.abs()).all()), (key, float(err.max()), float(b.abs().max()))

    # QAT scales: a MAX over ranks of per-shard maxima is the maximum over the batch, bit for bit
    assert torch.equal(two["s1_absmax"], one["s1_absmax"])
    assert torch.equal(two["s1_running"], one["s1_running"])
    close("s1_loss", 1e-5)
    gmax = float(one["s1_grads"].abs().max())
    close("s1_grads", 2e-4, 2e-6 * gmax)
    # Adam's first step moves every entry by about lr * sign(gradient): entries whose gradient is at rounding level aside
    for key in ("s1_params", "s2_gen_params", "s2_critic_params"):
        diff = (two[key] - one[key]).abs()
        assert float((diff > 1e-6).float().mean()) < 0.02, (key, float((diff > 1e-6).float().mean()))
    for k in ("g_total", "g_adv", "recon", "evm", "acpr") + tuple(f"d_total_{i}" for i in range(5)):
        close("s2_" + k, 2e-5, 1e-6)
    close("s2_critic_grads", 3e-4, 3e-6 * float(one["s2_critic_grads"].abs().max()))
    close("s2_gen_grads", 3e-4, 3e-6 * float(one["s2_gen_grads"].abs().max()))
    close("s2_u", 1e-5, 1e-7)      # the power iteration sees weights only: the same on every rank


def test_rccl_backend_one_rank(tmp_path):
    """The transport the multi-GPU runs use (backend "nccl" = RCCL) in a child process with one rank: the process
    group comes up on the GPU and the in-place reductions of parallel.py go through it on device tensors."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["DPD_ROOT"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import dpd_amd
from dpd_amd import parallel
t = torch.arange(2945, dtype=torch.float32, device="cuda")
parallel._all_reduce(t, dist.ReduceOp.SUM)
m = torch.tensor([0.25], device="cuda")
parallel._all_reduce(m, dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert float(t.sum()) == 2944 * 2945 / 2 and float(m) == 0.25
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
'''
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DPD_ROOT=os.path.dirname(HERE))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=150)
    assert p.returncode == 0 and "rccl ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
