#!/usr/bin/env python3
"""Throughput of the batched DPD hot path on MI355X: IQ samples/s.

  python bench.py --gpus N --steps K --warmup W [--workload stage1|stage1_qat]

One step = one pass of the workload over one batch of synthetic OFDM IQ rows resident in HBM
(per GPU: B = 16384 rows of S = 1029 samples -> 2^24 generator outputs).  With N > 1 (launched by
torch.distributed.run, one rank per GPU) the rows are sharded: every rank holds its own 2^24
samples (weak scaling) and the flat gradient buffer is all-reduced over RCCL once per step.
Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOPS_PEAK_FP32 = 157.3e12     # MI355X fp32 vector = fp32 MFMA dense peak
GEN_FWD_MAC, GEN_BWD_MAC = 1504, 2048


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="stage1", choices=["stage1", "stage1_qat"])
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--seq", type=int, default=1029)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline_stage1(qat: bool, seq: int):
    """The CPU oracle (vectorised torch fp32, all host threads) on a bounded sample of the same workload."""
    from oracle import generator as og
    from oracle import steps as ost
    rows = 256
    g = torch.Generator().manual_seed(1)
    iq = (torch.randn(rows, seq, 2, generator=g) * 0.25).clamp(-0.7, 0.7)
    target = iq[:, 5:, :].contiguous()
    torch.manual_seed(0)
    from dpd_amd.models import TDNNGeneratorQAT
    params = {k: v for k, v in TDNNGeneratorQAT().state_dict().items() if k in og.PARAM_KEYS}
    gs = ost.GenState(params)
    ost.stage1_step(gs, iq, target, qat)           # warm-up
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 10.0:
        ost.stage1_step(gs, iq, target, qat)
        reps += 1
    dt = time.perf_counter() - t0
    n = rows * (seq - 5) * reps
    return {"value": n / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} Stage-1 steps on {rows}x{seq} rows ({rows * (seq - 5)} samples each), oracle/steps.py, torch CPU fp32"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import dpd_amd  # noqa: F401
    from dpd_amd import _lib as L
    from dpd_amd.data import ofdm_rows
    from dpd_amd.models import TDNNGeneratorQAT
    from dpd_amd.training import Stage1Step

    qat = args.workload == "stage1_qat"
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(dev)
    gen.qat_enabled = qat
    gen.train()
    iq = ofdm_rows(args.rows, args.seq, seed=1234 + rank, device=dev)
    target = iq[:, 5:, :].contiguous()
    n_local = args.rows * (args.seq - 5)

    grad_sync = absmax_sync = None
    if world > 1:
        def grad_sync(buf):
            dist.all_reduce(buf)
            buf.div_(world)

        def absmax_sync(buf):
            dist.all_reduce(buf, op=dist.ReduceOp.MAX)
    step = Stage1Step(gen, grad_sync=grad_sync, absmax_sync=absmax_sync)

    # dominant-kernel timing: HIP events on the launch stream around the fused forward+MSE+backward call
    kern_ms = []
    orig_call = L.call

    def timed_call(name, *a):
        if name != "dpd_gen_mse_step" or not timing[0]:
            return orig_call(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_call(name, *a)
        e1.record()
        events.append((e0, e1))
    timing, events = [False], []
    import dpd_amd.training as tr
    tr.L.call = timed_call

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(iq, target)
    barrier()
    timing[0] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(iq, target)
    barrier()
    dt = time.perf_counter() - t0
    timing[0] = False
    tr.L.call = orig_call
    kern_ms = [a.elapsed_time(b) for a, b in events]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        total = n_local * world * args.steps
        kavg = sum(kern_ms) / len(kern_ms) * 1e-3
        flops = 2.0 * (GEN_FWD_MAC + GEN_BWD_MAC) * n_local
        achieved = flops / kavg / 1e12
        res = {
            "metric": "IQ samples/s: Q8.8 DPD inference + CWGAN-GP train step, 1/2/4/8 MI355X",
            "value": total / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Stage-1 supervised MSE step (generator fwd+loss+bwd fused, Adam), qat={'on' if qat else 'off'}",
                       "rows_per_gpu": args.rows, "seq_len": args.seq, "samples_per_gpu": n_local,
                       "parallelism": f"dp{world}"},
            "loss": float(loss.item()),
            "roofline": {"bound": "mfma", "kernel": "k_gen_bwd_mfma (forward recompute + MSE + backward, fp32 MFMA)",
                         "achieved": achieved, "peak": FLOPS_PEAK_FP32 / 1e12, "unit": "TFLOP/s",
                         "frac": achieved / (FLOPS_PEAK_FP32 / 1e12), "traffic": None,
                         "flop_per_sample": 2 * (GEN_FWD_MAC + GEN_BWD_MAC), "kernel_ms": kavg * 1e3},
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline_stage1(qat, args.seq)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
