#!/usr/bin/env python3
"""Throughput of the batched DPD hot path on MI355X, in IQ samples/s.

  python bench.py --gpus N --steps K --warmup W [--workload stage1|stage1_qat|stage2|q88|gen_pa|spectral] [--extras]

A step is one pass of the workload over one batch of synthetic OFDM-like IQ rows that is already
resident in HBM (per GPU: 16384 rows of 1029 samples -> 2^24 generator outputs).  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank holds its own 2^24 samples (weak
scaling), rows are never exchanged, and the flat gradient buffers are all-reduced over RCCL once
per backward.  Rank 0 prints one JSON line.

Headline workload (`stage1`): the Stage-1 supervised MSE training step of the reference's two-stage
schedule (generator forward + MSE + backward fused in one kernel, then Adam).  `--extras` adds short
measurements of the other workloads under "extra" in the same JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_PEAK_TFLOPS = 157.3      # MI355X fp32 vector == fp32 MFMA dense peak
HBM_PEAK_GBS = 8000.0
GEN_FWD_MAC, GEN_BWD_MAC = 1504, 2048           # per sample: 30x32 + 32x16 + 16x2 ; backward without input gradient
CRITIC_FWD_MAC, CRITIC_BWD_MAC, CRITIC_DX_MAC = 2832, 5408, 2704
METRIC = "IQ samples/s: Q8.8 DPD inference + CWGAN-GP train step, 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="stage1",
                    choices=["stage1", "stage1_qat", "stage2", "q88", "q88_stream", "gen_pa", "spectral"])
    ap.add_argument("--stream-chunks", type=int, default=32, help="q88_stream: chunks of --rows rows per step (32 x 2^24 = 2^29 samples)")
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--seq", type=int, default=1029)
    ap.add_argument("--extras", action="store_true", help="also time the other workloads briefly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


class Ctx:
    def __init__(self, args):
        import torch.distributed as dist
        self.dist = dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        # DPD_BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- exercises the N > 1 code
        # path on a one-GPU box (numbers from such a run mean nothing)
        rehearsal = os.environ.get("DPD_BENCH_REHEARSAL") == "1"
        if rehearsal:
            self.local = 0
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        if self.world > 1:
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
        import dpd_amd  # noqa: F401
        from dpd_amd.data import ofdm_rows
        from dpd_amd.parallel import DataParallel
        self.dp = DataParallel()
        self.rows, self.seq = args.rows, args.seq
        self.iq = ofdm_rows(args.rows, args.seq, seed=1234 + self.rank, device=self.dev)
        self.target = self.iq[:, 5:, :].contiguous()
        self.n_local = args.rows * (args.seq - 5)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, fn, steps, warmup):
        """W untimed + K timed steps between barrier+synchronize; max over ranks; seconds."""
        for _ in range(warmup):
            fn()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.barrier()
        dt = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = t.item()
        return dt


class KernelTimer:
    """HIP events (torch.cuda.Event on the launch stream == torch's current stream, which is the stream
    handed to the C ABI) around every call of one C-ABI entry point during the timed region."""

    def __init__(self, name):
        from dpd_amd import _lib
        self.lib, self.name, self.events, self.on = _lib, name, [], False
        self.orig = _lib.call

        def call(n, *a):
            if n != self.name or not self.on:
                return self.orig(n, *a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.orig(n, *a)
            e1.record()
            self.events.append((e0, e1))
        _lib.call = call

    def close(self):
        self.lib.call = self.orig

    def avg_seconds(self):
        ms = [a.elapsed_time(b) for a, b in self.events]
        return sum(ms) / max(len(ms), 1) * 1e-3


# ------------------------------------------------------------------------------------------ workloads
def make_generator(ctx, qat):
    from dpd_amd.models import TDNNGeneratorQAT
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(ctx.dev)
    gen.qat_enabled = qat
    return gen.train()


def pmc_traffic(ctx, qat):
    """HBM bytes per launch of the fused Stage-1 kernel from the rocprofv3 counter passes recorded under
    profiles/ (FETCH_SIZE doubled, the gfx950 correction, + WRITE_SIZE).  The counters cannot be read
    from inside this process; the figure applies to the default shape only, otherwise null."""
    path = os.path.join(ROOT, "profiles", "r1_stage1_pmc.json")
    if qat or (ctx.rows, ctx.seq) != (16384, 1029) or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return int(json.load(f)["per_launch"]["hbm_bytes"])
    except (OSError, KeyError, ValueError):
        return None


def run_stage1(ctx, args, qat, steps, warmup):
    from dpd_amd.training import Stage1Step
    gen = make_generator(ctx, qat)
    dp = ctx.dp
    step = Stage1Step(gen, grad_sync=dp.mean_ if dp.enabled else None, absmax_sync=dp.max_ if dp.enabled else None)
    kt = KernelTimer("dpd_gen_mse_step")
    for _ in range(warmup):
        step(ctx.iq, ctx.target)
    kt.on = True
    dt = ctx.timed(lambda: step(ctx.iq, ctx.target), steps, 0)
    kt.close()
    k = kt.avg_seconds()
    flop = 2.0 * (GEN_FWD_MAC + GEN_BWD_MAC) * ctx.n_local
    roof = {"bound": "mfma", "kernel": "k_gen_bwd_mfma: tap assembly + forward + MSE + backward, fp32 MFMA (+ slab reduce)",
            "achieved": flop / k / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": flop / k / 1e12 / FP32_PEAK_TFLOPS, "traffic": pmc_traffic(ctx, qat),
            "flop_per_sample": 2 * (GEN_FWD_MAC + GEN_BWD_MAC), "kernel_ms": k * 1e3}
    loss = float(step.grads[1554].item())
    return dt, roof, {"loss": loss}


def run_stage2(ctx, args, steps, warmup):
    from dpd_amd.models import Discriminator
    from dpd_amd.training import Stage2Step
    gen = make_generator(ctx, True)
    torch.manual_seed(1)
    disc = Discriminator().to(ctx.dev).train()
    cond = ctx.target
    step = Stage2Step(gen, disc, dp=ctx.dp)
    info = {}

    def fn():
        info.update(step(ctx.iq, ctx.target, cond))
    dt = ctx.timed(fn, steps, warmup)
    # algorithmic work of one step (n_critic = 5): per critic update one generator forward, two critic
    # forward+backward passes (real, fake) and the gradient penalty (forward, input gradient, double
    # backward); per generator update forward, critic forward + input gradient, generator backward
    mac = 5 * (GEN_FWD_MAC + 2 * (CRITIC_FWD_MAC + CRITIC_BWD_MAC) + (CRITIC_FWD_MAC + CRITIC_DX_MAC + CRITIC_BWD_MAC)) \
        + (GEN_FWD_MAC + CRITIC_FWD_MAC + CRITIC_DX_MAC + GEN_BWD_MAC)
    flop = 2.0 * mac * ctx.n_local
    t = dt / steps
    # the critic's hidden-layer products run on the bf16 matrix cores (three-way split operands, fp32
    # accuracy): "achieved" is in fp32-equivalent flops and "frac" compares it with the fp32 peak
    roof = {"bound": "mfma", "kernel": "whole step (critic kernels dominate; fp32-equivalent flops, part on bf16 matrix cores)",
            "achieved": flop / t / 1e12,
            "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop / t / 1e12 / FP32_PEAK_TFLOPS, "traffic": None,
            "flop_per_sample": 2 * mac}
    return dt, roof, {"g_total": float(info["g_total"].item())}


def run_q88(ctx, args, steps, warmup):
    from dpd_amd.utils import Q88Inference, export_q88_weights
    gen = make_generator(ctx, True)
    eng = Q88Inference(export_q88_weights(gen))
    iq16 = (ctx.iq * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    out = [None]

    def fn():
        out[0] = eng(iq16)
    kt = KernelTimer("dpd_q88_infer")
    for _ in range(warmup):
        fn()
    kt.on = True
    dt = ctx.timed(fn, steps, 0)
    kt.close()
    k = kt.avg_seconds()
    mac = GEN_FWD_MAC * ctx.n_local
    # v_dot2c_i32_i16 issues at 16 lanes per SIMD-cycle (half the fp32 FMA rate; scratch/ubench/dot2_rate.hip
    # measures 36.4 T lane-instructions/s): 256 CU x 4 SIMD x 16 lanes x 2 MAC x 2.4 GHz = 78.6 TMAC/s
    roof = {"bound": "valu-int", "kernel": "k_q88 (v_dot2c_i32_i16)", "achieved": mac / k / 1e12, "peak": 78.6,
            "unit": "TMAC/s", "frac": mac / k / 1e12 / 78.6, "traffic": None, "hbm_gbs": 8.0 * ctx.n_local / k / 1e9,
            "acc64": eng.acc64, "kernel_ms": k * 1e3}
    return dt, roof, {"checksum": int(out[0].to(torch.int64).sum().item())}


def run_q88_stream(ctx, args, steps, warmup):
    """Q8.8 inference over a long int16 stream held in HBM, processed in chunks of `--rows` rows
    (2^24 outputs each at the default shape); `--stream-chunks` chunks per step.  The whole output of
    the first timed step is checksummed against a second, separate pass (bitwise reproducibility)."""
    from dpd_amd.utils import Q88Inference, export_q88_weights
    gen = make_generator(ctx, True)
    eng = Q88Inference(export_q88_weights(gen))
    nchunk = args.stream_chunks
    g = torch.Generator(device=ctx.dev).manual_seed(4321 + ctx.rank)
    base = (ctx.iq * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    # distinct chunks: the base rows under per-chunk row permutations (cheap to make, not compressible to one chunk)
    stream = torch.stack([base[torch.randperm(base.shape[0], generator=g, device=ctx.dev)] for _ in range(nchunk)])
    outs = torch.empty(nchunk, base.shape[0], base.shape[1] - 5, 2, dtype=torch.int16, device=ctx.dev)

    def fn():
        for c in range(nchunk):
            outs[c] = eng(stream[c])

    def checksum():
        v = outs.view(torch.int32).to(torch.int64)
        idx = torch.arange(v.numel(), device=ctx.dev, dtype=torch.int64).view_as(v)
        return int(((v * ((idx % 65521) + 1)).sum() % 2305843009213693951).item())
    kt = KernelTimer("dpd_q88_infer")
    for _ in range(warmup):
        fn()
    first = checksum()
    outs.zero_()
    kt.on = True
    dt = ctx.timed(fn, steps, 0)
    kt.close()
    k = kt.avg_seconds()
    again = checksum()
    mac = GEN_FWD_MAC * ctx.n_local
    roof = {"bound": "valu-int", "kernel": "k_q88 (v_dot2c_i32_i16), per chunk", "achieved": mac / k / 1e12, "peak": 78.6,
            "unit": "TMAC/s", "frac": mac / k / 1e12 / 78.6, "traffic": None, "hbm_gbs": 8.0 * ctx.n_local / k / 1e9,
            "acc64": eng.acc64, "kernel_ms": k * 1e3}
    return dt / nchunk, roof, {"stream_samples_per_gpu": ctx.n_local * nchunk, "checksum": again,
                               "reproducible": bool(first == again) if warmup > 0 else None}


def run_gen_pa(ctx, args, steps, warmup):
    from dpd_amd.models import PADigitalTwin, dpd_pa_chain
    gen = make_generator(ctx, True).eval()
    pa = PADigitalTwin().to(ctx.dev)
    y = [None]

    def fn():
        y[0] = dpd_pa_chain(gen, pa, ctx.iq, return_dpd=False)[1]
    dt = ctx.timed(fn, steps, warmup)
    flop = 2.0 * (GEN_FWD_MAC + 100) * ctx.n_local
    t = dt / steps
    roof = {"bound": "mfma", "kernel": "QAT generator: 4 abs-max passes, then output quantiser + PA twin in one kernel", "achieved": flop / t / 1e12,
            "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop / t / 1e12 / FP32_PEAK_TFLOPS, "traffic": None}
    return dt, roof, {"y_rms": float(y[0].pow(2).mean().sqrt().item())}


def run_spectral(ctx, args, steps, warmup):
    from dpd_amd.utils import SpectralLoss
    sl = SpectralLoss()
    pred = (ctx.target * 0.9).requires_grad_(True)
    val = [None]

    def fn():
        pred.grad = None
        val[0] = sl(pred, ctx.target)
        val[0].backward()
    dt = ctx.timed(fn, steps, warmup)
    t = dt / steps
    gbs = 40.0 * ctx.n_local / t / 1e9      # fwd: pred + target read (16 B); bwd: both again + dpred write (24 B)
    roof = {"bound": "hbm", "kernel": "k_spectral_stats + k_spectral_bwd (row FFT in LDS)", "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None}
    return dt, roof, {"loss": float(val[0].item())}


def cpu_baseline_stage1(qat: bool, seq: int):
    """The CPU oracle (vectorised torch fp32, all host threads given to the job) on a bounded sample."""
    from oracle import generator as og
    from oracle import steps as ost
    from dpd_amd.models import TDNNGeneratorQAT
    rows = 256
    g = torch.Generator().manual_seed(1)
    iq = (torch.randn(rows, seq, 2, generator=g) * 0.25).clamp(-0.7, 0.7)
    target = iq[:, 5:, :].contiguous()
    torch.manual_seed(0)
    params = {k: v for k, v in TDNNGeneratorQAT().state_dict().items() if k in og.PARAM_KEYS}
    gs = ost.GenState(params)
    ost.stage1_step(gs, iq, target, qat)
    t0, reps = time.perf_counter(), 0
    while time.perf_counter() - t0 < 10.0:
        ost.stage1_step(gs, iq, target, qat)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": rows * (seq - 5) * reps / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} Stage-1 steps (qat={'on' if qat else 'off'}) of {rows} rows x {seq} samples "
                      f"({rows * (seq - 5)} outputs each) with oracle/steps.py, torch CPU fp32, ~10 s"}


RUNNERS = {
    "stage1": lambda c, a, s, w: run_stage1(c, a, False, s, w),
    "stage1_qat": lambda c, a, s, w: run_stage1(c, a, True, s, w),
    "stage2": run_stage2, "q88": run_q88, "q88_stream": run_q88_stream, "gen_pa": run_gen_pa, "spectral": run_spectral,
}
LABELS = {
    "stage1": "Stage-1 supervised MSE step: fused generator forward + MSE + backward, Adam; qat=off",
    "stage1_qat": "Stage-1 supervised MSE step with QAT on: 4 abs-max passes + fused forward/MSE/backward, Adam",
    "stage2": "Stage-2 CWGAN-GP step: 5 critic updates (Wasserstein + gradient penalty) + generator update with SpectralLoss; qat=on",
    "q88": "Q8.8 fixed-point DPD inference, int16 in/out",
    "q88_stream": "Q8.8 fixed-point DPD inference over a resident int16 stream, chunk by chunk (value = rate per chunk pass)",
    "gen_pa": "validation chain: QAT generator forward + PA digital twin",
    "spectral": "SpectralLoss forward + backward",
}


def main():
    args = parse()
    ctx = Ctx(args)
    dt, roof, aux = RUNNERS[args.workload](ctx, args, args.steps, args.warmup)
    res = None
    if ctx.rank == 0:
        res = {
            "metric": METRIC, "value": ctx.n_local * ctx.world * args.steps / dt, "unit": "samples/s", "n_gpus": ctx.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int16" if args.workload.startswith("q88") else "f32",
            "data": "synthetic",
            "config": {"workload": LABELS[args.workload], "name": args.workload, "rows_per_gpu": args.rows,
                       "seq_len": args.seq, "samples_per_gpu": ctx.n_local, "parallelism": f"dp{ctx.world}"},
            "roofline": roof,
        }
        res.update(aux)
    if args.extras:
        extra = {}
        for name in RUNNERS:
            if name == args.workload or name == "q88_stream":
                continue
            k = 2 if name == "stage2" else 5
            d, r, a = RUNNERS[name](ctx, args, k, 1)
            if ctx.rank == 0:
                extra[name] = {"value": ctx.n_local * ctx.world * k / d, "unit": "samples/s", "ms_per_step": d / k * 1e3,
                               "steps": k, "workload": LABELS[name], "roofline": r, **a}
        if ctx.rank == 0:
            res["extra"] = extra
    if ctx.rank == 0:
        if not args.no_cpu_baseline and ctx.world == 1 and args.workload in ("stage1", "stage1_qat"):
            res["cpu_baseline"] = cpu_baseline_stage1(args.workload == "stage1_qat", args.seq)
        print(json.dumps(res), flush=True)
    if ctx.world > 1:
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
