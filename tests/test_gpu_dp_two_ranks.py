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
