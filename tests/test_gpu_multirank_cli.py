"""The N > 1 code paths of the command-line programs, run as child processes on a one-GPU box: every rank on GPU 0 with
the gloo backend (DPD_BENCH_REHEARSAL=1).  What is under test is what the multi-GPU runs execute apart from the
transport: bench.py starting its own ranks (`--gpus 2` outside torch.distributed.run), the weak and the fixed-global-size
modes, the barriers and the max-over-ranks timing, and train.py on two ranks with a row count that does not divide."""
import csv
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["DPD_BENCH_REHEARSAL"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _bench(*flags):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras",
           "--no-cpu-baseline", *flags]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]           # rank 0 prints the one line
    return json.loads(lines[0])


def test_bench_starts_two_ranks_weak_scaling():
    res = _bench("--rows", "1024")
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["steps"] == 2
    assert res["config"]["parallelism"] == "dp2" and res["config"]["samples_per_gpu_per_step"] == 1024 * 1024
    assert np.isfinite(res["value"]) and res["value"] > 0
    # whole-job throughput: both ranks' samples over the slowest rank's time
    assert abs(res["value"] - 2 * 1024 * 1024 * 2 / (res["ms_per_step"] * 2e-3)) <= 1e-6 * res["value"]
    assert np.isfinite(res["g_total"]) and 0 < res["roofline"]["frac"] < 1


def test_bench_two_ranks_fixed_global_problem():
    res = _bench("--global-samples", str(1 << 22), "--global-batch", str(1 << 20))
    assert res["n_gpus"] == 2 and res["scaling"] == "strong"
    assert res["config"]["global_samples"] == 1 << 22 and res["config"]["global_batch"] == 1 << 20
    assert res["config"]["optimiser_steps_per_step"] == 4 and res["config"]["rows_per_gpu"] == 512
    assert np.isfinite(res["value"]) and res["value"] > 0


def _write_csv(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["I", "Q"])
        for a, b in rows:
            w.writerow([repr(float(a)), repr(float(b))])


def test_train_two_ranks_with_a_row_count_that_does_not_divide(tmp_path):
    """27 rows (9 windows x 3 temperature variants) over 2 ranks and batches of 4: every rank runs the same number of
    steps (rows beyond a multiple of world x batch are dropped), both epochs complete, rank 0 writes the checkpoints."""
    g = load_golden("g12_data")
    d = tmp_path / "data"
    d.mkdir()
    for split in ("train", "val"):
        _write_csv(d / f"{split}_input.csv", g["csv_input"])
        _write_csv(d / f"{split}_output.csv", g["csv_output"])
    out = tmp_path / "ck"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "train.py"), "--data-dir", str(d), "--epochs", "2", "--stage1-epochs", "1",
           "--seq-len", "261", "--batch-size", "4", "--temp", "all", "--qat", "--config", "", "--output-dir", str(out)]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    epochs = [ln for ln in p.stdout.splitlines() if ln.startswith("epoch ")]
    assert len(epochs) == 2 and "stage 2" in epochs[1], p.stdout[-2000:]
    assert (out / "latest.pth").exists()


def test_bench_line_contract_single_gpu():
    """The one JSON line of a single-GPU run (small rows, extras on, CPU baselines off): the keys the driver reads, the
    roofline object of the headline and of every extra workload, and a frac that is a fraction."""
    env = _env()
    env.pop("DPD_BENCH_REHEARSAL")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--rows", "256", "--stream-chunks", "2",
           "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "extra"):
        assert k in res, k
    assert res["n_gpus"] == 1 and res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["higher_is_better"] is True and res["vs_baseline"] is None and res["data"] == "synthetic"
    assert res["unit"] == "samples/s" and "workload" in res["config"]
    assert abs(res["value"] - 256 * 1024 / (res["ms_per_step"] * 1e-3)) <= 1e-6 * res["value"]
    roofs = {"stage2": res["roofline"], **{k: v["roofline"] for k, v in res["extra"].items()}}
    assert set(roofs) == {"stage2", "stage1", "stage1_qat", "q88", "q88_stream", "gen_pa", "gen_pa_float", "gen_pa_train", "spectral"}
    for name, r in roofs.items():
        assert r["bound"] in ("hbm", "mfma", "valu-issue"), name
        assert r["unit"] in ("GB/s", "TFLOP/s", "G wave-instructions/s"), name
        assert np.isfinite(r["achieved"]) and r["peak"] > 0 and "traffic" in r, name
        assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-6 + 1e-3 * r["frac"], (name, r["frac"])
    for v in res["extra"].values():
        assert np.isfinite(v["value"]) and v["value"] > 0 and "dtype" in v
