#!/usr/bin/env python3
"""Throughput of the batched DPD hot path on MI355X, in IQ samples/s.

  python bench.py --gpus N --steps K --warmup W
        [--workload stage2|stage1|stage1_qat|q88|q88_stream|gen_pa|gen_pa_train|spectral]
        [--no-extras] [--no-cpu-baseline] [--global-samples G [--global-batch Bg]]

A step is one pass of the workload over one batch of synthetic OFDM-like IQ rows that is already
resident in HBM (per GPU: 16384 rows of 1029 samples -> 2^24 generator outputs).

Headline workload (`stage2`, the configuration the metric is quoted on): one CWGAN-GP training step -- 5 critic updates
(Wasserstein terms + gradient penalty with its double backward) and one generator update (adversarial +
L1 + SpectralLoss with the row FFT), QAT generator, Adam.  The other workloads (Stage-1 step with and
without QAT, Q8.8 integer inference, the generator -> PA chain forward and with its fused backward,
SpectralLoss) are timed briefly in the same run and reported under "extra", each with the roofline
of its dominant kernel (timed with HIP events on the launch stream) and a CPU baseline.

N > 1: one rank per GPU.  Launched by torch.distributed.run the ranks are already there; called as
plain `python bench.py --gpus N` this process starts them (before it touches a GPU) and passes the
ranks' JSON line through.  Weak scaling by default (every rank holds its own 2^24 samples, rows are
never exchanged, flat gradient buffers and QAT scales are all-reduced over RCCL).
`--global-samples G` switches to a fixed global problem (the data-parallel configuration: G = 2^28): G samples
are split over the ranks and one step is G / Bg optimiser steps on global batches of Bg samples
(default 2^24), each with its gradient all-reduce; "scaling" is then "strong".
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3      # MI355X fp32 vector == fp32 MFMA dense peak (256 CU x 4 SIMD x 64 flop/cycle x 2.4 GHz)
BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak (v_mfma_f32_16x16x32_bf16: 16,384 flop per 16 cycles and SIMD)
HBM_PEAK_GBS = 8000.0
CLOCK_HZ = 2.4e9
N_SIMD = 1024
GEN_FWD_MAC, GEN_BWD_MAC = 1504, 2048           # per sample: 30x32 + 32x16 + 16x2 ; backward without input gradient
CRITIC_FWD_MAC, CRITIC_BWD_MAC, CRITIC_DX_MAC = 2832, 5408, 2704
PA_FLOP = 200                                   # memory polynomial, 4 orders x 5 taps (complex Horner form)
PA_BWD_FLOP = 400
# MFMA instructions the critic's fused mean-score pass (forward + backward) issues per tile of 16 samples, with
# their issue cycles per SIMD (counted in the kernel's assembly): FC1 runs on v_mfma_f32_16x16x4_f32 (4 x 32 cycles) and
# the first-layer weight gradient on v_mfma_f32_4x4x1_16b_f32 (16 x 8); the hidden-layer products (60) and the weight
# gradients of the two inner layers (30) as three-way-split bf16 operands on v_mfma_f32_16x16x32_bf16 (16 cycles)
CRITIC_BWD_BF16_MFMA = 90
# k_q88_mfma, per chunk of 64 samples and wave: executed instructions, (SQ_INSTS_VALU - SQ_INSTS_MFMA) / chunks of a counter
# pass over the workload (profiles/r3_pmc_extras.json; the chunk loop of the compiled kernel holds 633 with its rarely taken branches)
Q88_VALU_PER_CHUNK = 573
Q88_MFMA_PER_CHUNK = 40
# QAT generator kernels in the digit form (fake-quantised values as two exact bf16 digits), per chunk of 64 samples and
# wave, executed instructions from the counter passes in profiles/r3_pmc_extras.json ((SQ_INSTS_VALU - SQ_INSTS_MFMA) / chunks;
# the loop bodies of the compiled kernels, scratch/isa_hist.py, hold 7-10 % more in rarely taken branches): vector-ALU instructions, bf16 MFMAs
# (v_mfma_f32_16x16x32_bf16, 16 cycles on the matrix pipe, 4 on the issue port) and fp32 MFMAs (v_mfma_f32_16x16x4_f32,
# 32 cycles during which the SIMD issues nothing else)
GEN_QAT_FWD_P4 = {"valu": 526, "bf16": 48, "fp32": 0}       # k_gen_fwd_mfma<true,3,0>: last abs-max pass
GEN_QAT_FWD_PA = {"valu": 571, "bf16": 48, "fp32": 0}       # k_gen_fwd_mfma<true,4,3>: fused chain forward
GEN_QAT_BWD = {"valu": 1056, "bf16": 144, "fp32": 0}           # k_gen_bwd_mfma<true,true,false>: Stage-1 step
GEN_QAT_BWD_PA = {"valu": 1231, "bf16": 140, "fp32": 0}        # k_gen_bwd_mfma<true,true,true>: chain step (56 outputs per chunk)
# float generator: FC1 and FC2 of the forward as three-way split bf16 products (18 MFMAs per tile); backward products
# (W2^T, dW1, dW2) as three-way split products as well (24 per tile)
GEN_FLOAT_FWD_PA = {"valu": 492, "bf16": 72, "fp32": 0}     # k_gen_fwd_mfma<false,4,3>: fused chain forward
GEN_FLOAT_BWD = {"valu": 980, "bf16": 168, "fp32": 0}         # k_gen_bwd_mfma<false,true,false>
CRITIC_BWD_MFMA_CYCLES = 4 * 32 + 16 * 8 + CRITIC_BWD_BF16_MFMA * 16
METRIC = "IQ samples/s: Q8.8 DPD inference + CWGAN-GP train step, 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="stage2",
                    choices=["stage2", "stage1", "stage1_qat", "q88", "q88_stream", "gen_pa", "gen_pa_float", "gen_pa_train", "spectral"])
    ap.add_argument("--stream-chunks", type=int, default=32, help="q88_stream: chunks of --rows rows per step (32 x 2^24 = 2^29 samples)")
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--seq", type=int, default=1029)
    ap.add_argument("--global-samples", type=int, default=0, help="fixed global problem size in samples (strong scaling), e.g. 268435456")
    ap.add_argument("--global-batch", type=int, default=1 << 24, help="with --global-samples: samples per optimiser step over all ranks")
    ap.add_argument("--extras", action="store_true", help="(default now; kept for older command lines)")
    ap.add_argument("--no-extras", action="store_true", help="only the headline workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child job.  Runs
    before anything in this process touches a GPU; the child's exit code is this process's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class Ctx:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        # DPD_BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- exercises the N > 1 code
        # path on a one-GPU box (numbers from such a run mean nothing)
        self.rehearsal = os.environ.get("DPD_BENCH_REHEARSAL") == "1"
        if self.rehearsal:
            self.local = 0
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        if self.world > 1:
            if self.rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
        import dpd_amd  # noqa: F401
        from dpd_amd.data import ofdm_rows
        from dpd_amd.parallel import DataParallel
        self.dp = DataParallel()
        self.seq = args.seq
        L = args.seq - 5
        self.micro = 1
        if args.global_samples:
            per_step = args.global_batch // self.world            # samples per rank and optimiser step
            self.rows = max(per_step // L, 1)
            self.micro = max(args.global_samples // max(args.global_batch, 1), 1)
            self.strong = True
        else:
            self.rows, self.strong = args.rows, False
        self.n_local = self.rows * L * self.micro               # samples this rank processes per bench step
        # micro-batches are distinct row sets, all resident in HBM before the timed region
        self.iqs = [ofdm_rows(self.rows, args.seq, seed=1234 + self.rank + 1000 * m, device=self.dev) for m in range(self.micro)]
        self.targets = [iq[:, 5:, :].contiguous() for iq in self.iqs]
        self.iq, self.target = self.iqs[0], self.targets[0]

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

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
            t = self.torch.tensor([dt], device=self.dev, dtype=self.torch.float64)
            if self.rehearsal:
                t = t.cpu()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt


class KernelTimer:
    """HIP events (torch.cuda.Event on the launch stream: torch's current stream is the stream handed to
    the C ABI) around every call of one C-ABI entry point during the timed region; `when` (optional)
    selects the calls by their arguments."""

    def __init__(self, name, when=None):
        import torch
        from dpd_amd import _lib
        self.torch, self.lib, self.name, self.events, self.on, self.when = torch, _lib, name, [], False, when
        self.orig = _lib.call

        def call(n, *a):
            if (n not in self.name if isinstance(self.name, tuple) else n != self.name) or not self.on or (self.when is not None and not self.when(a)):
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
        self.torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.events]
        return sum(ms) / max(len(ms), 1) * 1e-3


def pmc_bytes(key):
    """HBM bytes per launch of a kernel from the rocprofv3 counter passes recorded under profiles/
    (TCC_EA read and write requests converted to bytes as the microarchitecture guide prescribes).  The counters cannot
    be read from inside this process; the figures apply to the default shape only."""
    for name in ("r3_pmc_stage2.json", "r3_pmc_extras.json", "r2_pmc_stage2.json", "r2_pmc_extras.json", "r1_stage1_pmc.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                d = json.load(f)
            if key in d:
                return int(d[key]["hbm_bytes"])
            if key == "k_gen_bwd_mfma<false,true>" and "per_launch" in d:
                return int(d["per_launch"]["hbm_bytes"])
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None


def simd_time_fracs(counts, chunks, k):
    """Utilisation figures of a kernel from its per-chunk instruction counts: `matrix` = cycles its MFMAs occupy the matrix
    pipe (fp32 16x16x4: 32, bf16 16x16x32: 16) over the measured cycles per SIMD; `issue` = cycles its vector instructions
    hold the SIMD's issue port (VALU 4, bf16 MFMA 4, fp32 MFMA all of its 32) over the same."""
    cyc = k * CLOCK_HZ / (chunks / N_SIMD)
    matrix = (counts["fp32"] * 32 + counts["bf16"] * 16) / cyc
    issue = (counts["valu"] * 4 + counts["bf16"] * 4 + counts["fp32"] * 32) / cyc
    return matrix, issue, cyc


def default_shape(ctx):
    return (ctx.rows, ctx.seq, ctx.micro) == (16384, 1029, 1)


# ------------------------------------------------------------------------------------------ workloads
def make_generator(ctx, qat):
    from dpd_amd.models import TDNNGeneratorQAT
    ctx.torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(ctx.dev)
    gen.qat_enabled = qat
    return gen.train()


def run_stage1(ctx, args, qat, steps, warmup):
    from dpd_amd.training import Stage1Step
    gen = make_generator(ctx, qat)
    dp = ctx.dp
    step = Stage1Step(gen, grad_sync=dp.mean_ if dp.enabled else None, absmax_sync=dp.max_ if dp.enabled else None)

    def fn():
        for iq, tg in zip(ctx.iqs, ctx.targets):
            step(iq, tg)
    kt = KernelTimer("dpd_gen_mse_step")
    for _ in range(warmup):
        fn()
    kt.on = True
    dt = ctx.timed(fn, steps, 0)
    kt.close()
    k = kt.avg_seconds()
    n = ctx.rows * (ctx.seq - 5)
    flop = 2.0 * (GEN_FWD_MAC + GEN_BWD_MAC) * n
    counts = GEN_QAT_BWD if qat else GEN_FLOAT_BWD
    matrix, issue, cyc = simd_time_fracs(counts, n / 64.0, k)
    ginst = (counts["valu"] + counts["bf16"]) * (n / 64.0) / k / 1e9
    roof = {"bound": "valu-issue",
            "kernel": ("k_gen_bwd_mfma<true,true,false>: tap assembly + forward (exact digit products) + MSE + backward (three-way split "
                       "products), all on the bf16 MFMA at fp32 accuracy (+ slab reduce)" if qat else
                       "k_gen_bwd_mfma<false,true,false>: tap assembly + forward + MSE + backward, every product a three-way split product "
                       "on the bf16 MFMA at fp32 accuracy (+ slab reduce)"),
            # The kernel's products run on the bf16 matrix cores (36-42 MFMAs of 16 cycles per tile), which leaves the matrix pipe
            # mostly idle by design; what bounds it is the issue port of the SIMDs (splits, quantisers, masks, FC3, tanh).  frac =
            # cycles its vector instructions hold that port over the measured kernel cycles
            "achieved": ginst, "peak": N_SIMD * CLOCK_HZ / 4 / 1e9, "unit": "G wave-instructions/s", "frac": issue,
            "frac_definition": "vector issue time of one launch (%d VALU + %d bf16 MFMA wave-instructions per 64 samples, executed counts "
                               "from a counter pass, 4 cycles each on the SIMD's issue port, 1,024 SIMDs at 2.4 GHz) / measured kernel time"
                               % (counts["valu"], counts["bf16"]),
            "matrix_pipe_frac": matrix, "cycles_per_64_samples": cyc,
            # algorithmic fp32 flops (2 x (1,504 + 2,048) MAC per sample) over the kernel time against the fp32 peak: a speed-up
            # measure, not a utilisation (nothing here runs on the fp32 pipe)
            "fp32_equivalent": {"achieved": flop / k / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "ratio": flop / k / 1e12 / FP32_PEAK_TFLOPS},
            "traffic": pmc_bytes("k_gen_bwd_mfma<%s,true,false>" % ("true" if qat else "false")) if default_shape(ctx) else None,
            "flop_per_sample": 2 * (GEN_FWD_MAC + GEN_BWD_MAC), "kernel_ms": k * 1e3}
    if qat:
        # the four abs-max passes recompute the forward up to the layer whose scale they produce: work that the
        # roofline does not credit.  Executed MACs per sample: pass 1 960, pass 2 1472, pass 3 1504, then the fused step
        roof["note"] = "whole step also runs 4 abs-max passes (3,936 MAC/sample of recomputation, not credited)"
    loss = float(step.grads[1554].item())
    return dt, roof, {"loss": loss}


def run_stage2(ctx, args, steps, warmup):
    from dpd_amd.models import Discriminator
    from dpd_amd.training import Stage2Step
    torch = ctx.torch
    gen = make_generator(ctx, True)
    torch.manual_seed(1)
    disc = Discriminator().to(ctx.dev).train()
    step = Stage2Step(gen, disc, dp=ctx.dp)
    info = {}

    def fn():
        for iq, tg in zip(ctx.iqs, ctx.targets):
            info.update(step(iq, tg, tg))
    # dominant kernel: the critic's fused mean-score pass (forward + backward with weight gradients), launched
    # twice per critic update (real, fake): dpd_critic_backward calls that ask for grads and not for gx
    kt = KernelTimer("dpd_critic_backward", when=lambda a: a[7].value is not None and a[8].value is None)
    for _ in range(warmup):
        fn()
    kt.on = True
    dt = ctx.timed(fn, steps, 0)
    kt.close()
    k = kt.avg_seconds()
    n = ctx.rows * (ctx.seq - 5)
    # algorithmic work of one step as executed (n_critic = 5): ONE generator forward serves the step; per critic
    # update two critic forward+backward passes (real, fake) and the gradient penalty (forward, input gradient,
    # double backward); for the generator update critic forward + input gradient and the generator backward
    mac = GEN_FWD_MAC + 5 * (2 * (CRITIC_FWD_MAC + CRITIC_BWD_MAC) + (CRITIC_FWD_MAC + CRITIC_DX_MAC + CRITIC_BWD_MAC)) \
        + (CRITIC_FWD_MAC + CRITIC_DX_MAC + GEN_BWD_MAC)
    t = dt / steps / ctx.micro
    kflop = 2.0 * (CRITIC_FWD_MAC + CRITIC_BWD_MAC) * n
    tiles_per_simd = n / 16.0 / N_SIMD
    issue_s = tiles_per_simd * CRITIC_BWD_MFMA_CYCLES / CLOCK_HZ
    bf16_flop = (n / 16.0) * CRITIC_BWD_BF16_MFMA * 2 * 16 * 16 * 32
    util = issue_s / k
    roof = {"bound": "mfma",
            "kernel": "k_critic_bwd_mfma<true,false>: critic forward + backward with weight gradients in one pass "
                      "(10 of the step's launches); fp32 MFMA for FC1 and the first-layer weight gradient, hidden-layer "
                      "products and the inner weight gradients as 3-way-split bf16 operands on the bf16 MFMA (fp32 accuracy)",
            # utilisation of the matrix pipes: the time the kernel's MFMAs need at their own peak issue rates (4 fp32 16x16x4
            # at 32 cycles, 16 fp32 4x4x1 at 8, 90 bf16 16x16x32 at 16 per tile of 16 samples and SIMD, 2.4 GHz) over the
            # measured kernel time.  `achieved` states it on the scale of the pipe that carries 85 % of those cycles
            # (dense bf16, 2.5 PFLOP/s): frac x peak
            "achieved": util * BF16_PEAK_TFLOPS, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": util,
            "frac_definition": "matrix-pipe utilisation: MFMA issue time of one launch at the pipes' peak rates / measured kernel time "
                               "(fp32 and bf16 MFMA cycles added; 'achieved' = frac x the dense bf16 peak)",
            "mfma_issue_ms": issue_s * 1e3, "kernel_ms": k * 1e3,
            # the same launch as an fp32-equivalent rate: algorithmic fp32 flops (2 x (2,832 + 5,408) MAC per sample) over
            # the kernel time against the fp32 peak.  Not a utilisation: it can pass 1, because 90 of the 110 MFMAs of a
            # tile run on the bf16 pipe as split products of fp32 accuracy
            "fp32_equivalent": {"achieved": kflop / k / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "ratio": kflop / k / 1e12 / FP32_PEAK_TFLOPS},
            # the bf16 side by itself: flops the kernel's bf16 MFMAs execute (90 16x16x32 per tile of 16 samples: six
            # products per fp32-accurate product) over the kernel time, against the dense bf16 peak
            "bf16_executed": {"achieved": bf16_flop / k / 1e12, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": bf16_flop / k / 1e12 / BF16_PEAK_TFLOPS},
            "flop_per_sample": 2 * (CRITIC_FWD_MAC + CRITIC_BWD_MAC),
            "traffic": pmc_bytes("k_critic_bwd_mfma<true,false>") if default_shape(ctx) else None,
            "step_fp32_equivalent_tflops": 2.0 * mac * n / t / 1e12, "step_flop_per_sample": 2 * mac,
            "step_fp32_equivalent_ratio": 2.0 * mac * n / t / 1e12 / FP32_PEAK_TFLOPS}
    return dt, roof, {"g_total": float(info["g_total"].item())}


def q88_engine(ctx):
    from dpd_amd.utils import Q88Inference, export_q88_weights
    gen = make_generator(ctx, True)
    return Q88Inference(export_q88_weights(gen))


def q88_roof(ctx, k, eng):
    n = ctx.rows * (ctx.seq - 5)
    mac = GEN_FWD_MAC * n
    hbm = 8.0 * n / k / 1e9                  # 4 B in + 4 B out per sample
    if getattr(eng, "mode", 0) == 2:
        # Matrix-core form.  What bounds it is the issue port of the SIMDs: per chunk of 64 samples a wave issues
        # Q88_VALU_PER_CHUNK vector-ALU instructions (requantisation, LeakyReLU, digit split, feature planes; counted in the
        # kernel's ISA, scratch/isa_hist.py) and 40 MFMAs.  A wave64 VALU instruction holds the port of its SIMD for 4 cycles,
        # an int8 MFMA for 4 of its 16: the floor is (573 x 4 + 40 x 4) cycles per chunk and SIMD.  frac = that floor over the
        # measured kernel time; the matrix pipe and HBM are reported beside it and are far from their peaks.
        chunks_per_simd = n / 64.0 / N_SIMD
        issue_s = chunks_per_simd * (Q88_VALU_PER_CHUNK * 4 + Q88_MFMA_PER_CHUNK * 4) / CLOCK_HZ
        ginst = (Q88_VALU_PER_CHUNK + Q88_MFMA_PER_CHUNK) * (n / 64.0) / k / 1e9
        peak = N_SIMD * CLOCK_HZ / 4 / 1e9
        return {"bound": "valu-issue", "achieved": ginst, "peak": peak, "unit": "G wave-instructions/s", "frac": issue_s / k,
                "frac_definition": "vector issue time of one launch (573 VALU + 40 MFMA wave-instructions per 64 samples, 4 cycles each "
                                   "on the SIMD's issue port, 1,024 SIMDs at 2.4 GHz) / measured kernel time",
                "kernel": "k_q88_mfma (v_mfma_i32_16x16x64_i8, int16 = two 8-bit digits, three int32 accumulators per product)",
                "issue_floor_ms": issue_s * 1e3, "kernel_ms": k * 1e3, "valu_per_64_samples": Q88_VALU_PER_CHUNK,
                "mfma_per_64_samples": Q88_MFMA_PER_CHUNK, "mac_per_sample": GEN_FWD_MAC, "tmac_per_s": mac / k / 1e12,
                # the other ceilings of this kernel, for scale: executed and useful int8 MACs against the dense int8 MFMA peak
                # (2.5e15 MAC/s; 10,240 executed per sample, 6,016 useful), and bytes against HBM
                "mfma_i8_frac": 10240.0 * n / k / 2.5e15, "mfma_i8_useful_frac": 4.0 * mac / k / 2.5e15,
                "hbm_gbs": hbm, "hbm_frac": hbm / HBM_PEAK_GBS,
                "traffic": pmc_bytes("k_q88_mfma") if default_shape(ctx) else None}
    # integer-VALU form: v_dot2c_i32_i16 issues at 16 lanes per SIMD-cycle (half the fp32 FMA rate; scratch/ubench/dot2_rate.hip
    # measures 36.4 T lane-instructions/s): 256 CU x 4 SIMD x 16 lanes x 2 MAC x 2.4 GHz = 78.6 TMAC/s
    return {"bound": "valu-int", "achieved": mac / k / 1e12, "peak": 78.6, "unit": "TMAC/s", "frac": mac / k / 1e12 / 78.6,
            "kernel": "k_q88 (v_dot2c_i32_i16)", "acc64": eng.acc64, "hbm_gbs": hbm, "hbm_frac": hbm / HBM_PEAK_GBS,
            "kernel_ms": k * 1e3, "mac_per_sample": GEN_FWD_MAC,
            "traffic": pmc_bytes("k_q88<false>") if default_shape(ctx) else None}


def run_q88(ctx, args, steps, warmup):
    torch = ctx.torch
    eng = q88_engine(ctx)
    iq16 = [(iq * 32768.0).round().clamp(-32768, 32767).to(torch.int16) for iq in ctx.iqs]
    out = [None]

    def fn():
        for x in iq16:
            out[0] = eng(x)
    kt = KernelTimer(("dpd_q88_infer", "dpd_q88_matrix_infer"))
    for _ in range(warmup):
        fn()
    kt.on = True
    dt = ctx.timed(fn, steps, 0)
    kt.close()
    return dt, q88_roof(ctx, kt.avg_seconds(), eng), {"checksum": int(out[0].to(torch.int64).sum().item())}


def run_q88_stream(ctx, args, steps, warmup):
    """Q8.8 inference over a long int16 stream held in HBM, processed in chunks of `--rows` rows
    (2^24 outputs each at the default shape); `--stream-chunks` chunks per step.  The whole output of
    the first timed step is checksummed against a second, separate pass (bitwise reproducibility)."""
    torch = ctx.torch
    eng = q88_engine(ctx)
    nchunk = args.stream_chunks
    g = torch.Generator(device=ctx.dev).manual_seed(4321 + ctx.rank)
    base = (ctx.iq * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    # distinct chunks: the base rows under per-chunk row permutations (cheap to make, not compressible to one chunk)
    stream = torch.stack([base[torch.randperm(base.shape[0], generator=g, device=ctx.dev)] for _ in range(nchunk)])
    outs = torch.empty(nchunk, base.shape[0], base.shape[1] - 5, 2, dtype=torch.int16, device=ctx.dev)

    def fn():
        for c in range(nchunk):
            eng(stream[c], out=outs[c])

    def checksum():
        v = outs.view(torch.int32).to(torch.int64)
        idx = torch.arange(v.numel(), device=ctx.dev, dtype=torch.int64).view_as(v)
        return int(((v * ((idx % 65521) + 1)).sum() % 2305843009213693951).item())
    kt = KernelTimer(("dpd_q88_infer", "dpd_q88_matrix_infer"))
    for _ in range(warmup):
        fn()
    first = checksum()
    outs.zero_()
    kt.on = True
    dt = ctx.timed(fn, steps, 0)
    kt.close()
    again = checksum()
    ctx.n_local = ctx.rows * (ctx.seq - 5) * nchunk
    return dt, q88_roof(ctx, kt.avg_seconds(), eng), {"stream_samples_per_gpu": ctx.n_local, "checksum": again,
                                                      "reproducible": bool(first == again) if warmup > 0 else None}


def run_gen_pa(ctx, args, steps, warmup, qat=True):
    from dpd_amd.models import PADigitalTwin, dpd_pa_chain
    gen = make_generator(ctx, qat).eval()
    pa = PADigitalTwin().to(ctx.dev)
    y = [None]

    def fn():
        for iq in ctx.iqs:
            y[0] = dpd_pa_chain(gen, pa, iq, return_dpd=False)[1]
    # float generator: the whole chain is one kernel.  QAT: the dominant kernel is the last abs-max pass, a full
    # generator forward that also writes the (unquantised) output; quantiser + PA follow in a streaming kernel
    if qat:
        kt = KernelTimer("dpd_gen_absmax_pass", when=lambda a: a[0] == 3)
    else:
        kt = KernelTimer("dpd_gen_pa_forward")
    for _ in range(warmup):
        fn()
    kt.on = True
    dt = ctx.timed(fn, steps, 0)
    kt.close()
    k = kt.avg_seconds()
    n = ctx.rows * (ctx.seq - 5)
    kflop = (2.0 * GEN_FWD_MAC + (0 if qat else PA_FLOP)) * n
    flop = (2.0 * GEN_FWD_MAC + PA_FLOP) * n
    t = dt / steps / ctx.micro
    kname = ("k_gen_fwd_mfma<true,3,0>: tap assembly + QAT generator forward to the unquantised output (4th abs-max pass)" if qat else
             "k_gen_fwd_mfma<false,4,3>: tap assembly + generator (FC1, FC2 as three-way split products on the bf16 MFMA, fp32 accuracy) + "
             "PA twin in one kernel (whole rows per wave, outputs carried)")
    roof = {"bound": "mfma", "kernel": kname,
            "achieved": kflop / k / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": kflop / k / 1e12 / FP32_PEAK_TFLOPS,
            "traffic": pmc_bytes("k_gen_fwd_mfma<true,3,0>" if qat else "k_gen_fwd_mfma<false,4,3>") if default_shape(ctx) else None,
            "flop_per_sample": 2 * GEN_FWD_MAC + (0 if qat else PA_FLOP), "kernel_ms": k * 1e3,
            "chain_frac": flop / t / 1e12 / FP32_PEAK_TFLOPS}
    if not qat:
        # float: FC1 and FC2 as three-way split products on the bf16 matrix cores (18 MFMAs of 16 cycles per tile, fp32
        # accuracy); like the QAT form below the kernel is bound by the issue port of the SIMDs, and frac is that utilisation
        matrix, issue, cyc = simd_time_fracs(GEN_FLOAT_FWD_PA, n / 64.0, k)
        ginst = (GEN_FLOAT_FWD_PA["valu"] + GEN_FLOAT_FWD_PA["bf16"]) * (n / 64.0) / k / 1e9
        roof.update({"bound": "valu-issue", "achieved": ginst, "peak": N_SIMD * CLOCK_HZ / 4 / 1e9, "unit": "G wave-instructions/s",
                     "frac": issue,
                     "frac_definition": "vector issue time of one launch (492 VALU + 72 bf16 MFMA wave-instructions per 64 samples, 4 cycles "
                                        "each on the SIMD's issue port, 1,024 SIMDs at 2.4 GHz) / measured kernel time",
                     "matrix_pipe_frac": matrix, "cycles_per_64_samples": cyc,
                     "fp32_equivalent": {"achieved": kflop / k / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "ratio": kflop / k / 1e12 / FP32_PEAK_TFLOPS}})
    if qat:
        # The QAT forward kernels evaluate the layers as exact digit products on the bf16 matrix cores (12 MFMAs of 16 cycles per
        # tile), which leaves the matrix pipe mostly idle by design; what bounds them is the issue port of the SIMDs (quantisers,
        # digit split, LeakyReLU, FC3, tanh on the vector ALU).  frac = cycles the kernel's vector instructions hold that port over
        # the measured kernel cycles; the matrix-pipe share and the fp32-equivalent rate (credited flops over the kernel time
        # against the fp32 peak: a speed-up measure, not a utilisation) stand beside it.
        matrix, issue, cyc = simd_time_fracs(GEN_QAT_FWD_P4, n / 64.0, k)
        ginst = (GEN_QAT_FWD_P4["valu"] + GEN_QAT_FWD_P4["bf16"]) * (n / 64.0) / k / 1e9
        roof.update({"bound": "valu-issue", "achieved": ginst, "peak": N_SIMD * CLOCK_HZ / 4 / 1e9, "unit": "G wave-instructions/s",
                     "frac": issue,
                     "frac_definition": "vector issue time of one launch (526 VALU + 48 bf16 MFMA wave-instructions per 64 samples, 4 cycles "
                                        "each on the SIMD's issue port, 1,024 SIMDs at 2.4 GHz) / measured kernel time",
                     "matrix_pipe_frac": matrix, "cycles_per_64_samples": cyc,
                     "fp32_equivalent": {"achieved": kflop / k / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "ratio": kflop / k / 1e12 / FP32_PEAK_TFLOPS}})
        roof["note"] = ("chain_frac credits 3,208 flop/sample over the whole chain against the fp32 peak; the chain executes the layers "
                        "below each scale again in its abs-max passes (2 x (960 + 1,472 + 1,504) + 200 = 8,072 flop/sample)")
        # the one-kernel form of the QAT chain (tap assembly, three layers, the four quantisers, PA twin) with the scales of this
        # batch already known: what dpd_gen_pa_forward(qat=1) runs; timed on its own here, kernel only
        from dpd_amd import _lib as L
        torch = ctx.torch
        B, S = ctx.iq.shape[0], ctx.iq.shape[1]
        st = L.stream()
        wq = torch.empty(L.GEN_NPARAM, device=ctx.dev)
        absmax = torch.zeros(L.GEN_NSCALE, device=ctx.dev)
        ps = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
        L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), 1, L.fptr(wq), L.fptr(absmax), st)
        for stage in range(4):
            L.call("dpd_gen_absmax_pass", stage, L.fptr(ctx.iq), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.c_p(0), 0, st)
        cre, cim = pa.pa_model.coeffs_real.contiguous(), pa.pa_model.coeffs_imag.contiguous()
        yb = torch.empty(B, S - 9, 2, device=ctx.dev)

        def fused():
            L.call("dpd_gen_pa_forward", L.fptr(ctx.iq), B, S, L.fptr(wq), 0.2, 1, L.fptr(absmax), L.fptr(cre), L.fptr(cim), 4, 5,
                   L.c_p(0), L.fptr(yb), st)
        for _ in range(3):
            fused()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fused()
        e1.record()
        torch.cuda.synchronize()
        kf = e0.elapsed_time(e1) / 10 * 1e-3
        fmatrix, fissue, fcyc = simd_time_fracs(GEN_QAT_FWD_PA, n / 64.0, kf)
        roof["fused_qat_kernel"] = {"kernel": "k_gen_fwd_mfma<true,4,3>: tap assembly + QAT generator (quantisers on chip, scales given) + PA twin in one kernel",
                                    "kernel_ms": kf * 1e3, "bound": "valu-issue", "frac": fissue,
                                    "frac_definition": "as above with 571 VALU + 48 bf16 MFMA wave-instructions per 64 samples",
                                    "matrix_pipe_frac": fmatrix, "cycles_per_64_samples": fcyc,
                                    "fp32_equivalent": {"achieved": flop / kf / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                        "ratio": flop / kf / 1e12 / FP32_PEAK_TFLOPS},
                                    "flop_per_sample": 2 * GEN_FWD_MAC + PA_FLOP}
    return dt, roof, {"y_rms": float(y[0].pow(2).mean().sqrt().item())}


def run_gen_pa_train(ctx, args, steps, warmup):
    """Generator -> PA twin with the fused backward: loss = MSE(PA(G(iq)), target), gradients of the generator."""
    from dpd_amd.models import PADigitalTwin
    from dpd_amd.training import ChainStep
    gen = make_generator(ctx, True)
    pa = PADigitalTwin().to(ctx.dev)
    dp = ctx.dp
    step = ChainStep(gen, pa, grad_sync=dp.mean_ if dp.enabled else None, absmax_sync=dp.max_ if dp.enabled else None)
    tgs = [tg[:, 4:, :].contiguous() * 31.0 for tg in ctx.targets]

    def fn():
        for iq, tg in zip(ctx.iqs, tgs):
            step(iq, tg)
    kt = KernelTimer("dpd_gen_pa_mse_step")
    for _ in range(warmup):
        fn()
    kt.on = True
    dt = ctx.timed(fn, steps, 0)
    kt.close()
    k = kt.avg_seconds()
    n = ctx.rows * (ctx.seq - 5)
    flop = (2.0 * (GEN_FWD_MAC + GEN_BWD_MAC) + PA_FLOP + PA_BWD_FLOP) * n
    # a chunk of the chain kernel advances by 56 PA outputs (its two tile pairs overlap by the PA memory): rows of 1,020 PA
    # outputs take 19 chunks
    chunks = ctx.rows * float(-(-(ctx.seq - 9) // 56))
    matrix, issue, cyc = simd_time_fracs(GEN_QAT_BWD_PA, chunks, k)
    ginst = (GEN_QAT_BWD_PA["valu"] + GEN_QAT_BWD_PA["bf16"]) * chunks / k / 1e9
    roof = {"bound": "valu-issue",
            "kernel": "k_gen_bwd_mfma<true,true,true>: tap assembly + generator + quantiser + PA twin + MSE + PA adjoint + generator backward; "
                      "every matrix product on the bf16 MFMA at fp32 accuracy (digit products forward, three-way split products backward)",
            "achieved": ginst, "peak": N_SIMD * CLOCK_HZ / 4 / 1e9, "unit": "G wave-instructions/s", "frac": issue,
            "frac_definition": "vector issue time of one launch (1,231 VALU + 140 bf16 MFMA wave-instructions per chunk of 56 PA outputs, "
                               "4 cycles each on the SIMD's issue port) / measured kernel time",
            "matrix_pipe_frac": matrix, "cycles_per_chunk": cyc,
            "fp32_equivalent": {"achieved": flop / k / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "ratio": flop / k / 1e12 / FP32_PEAK_TFLOPS},
            "traffic": pmc_bytes("k_gen_bwd_mfma<true,true,true>") if default_shape(ctx) else None,
            "flop_per_sample": 2 * (GEN_FWD_MAC + GEN_BWD_MAC) + PA_FLOP + PA_BWD_FLOP, "kernel_ms": k * 1e3}
    return dt, roof, {"loss": float(step.grads[1554].item())}


def run_spectral(ctx, args, steps, warmup):
    from dpd_amd.utils import SpectralLoss
    sl = SpectralLoss()
    preds = [(tg * 0.9).requires_grad_(True) for tg in ctx.targets]
    val = [None]

    def fn():
        for pred, tg in zip(preds, ctx.targets):
            pred.grad = None
            val[0] = sl(pred, tg)
            val[0].backward()
    ks, kb = KernelTimer("dpd_spectral_stats"), KernelTimer("dpd_spectral_backward")
    for _ in range(warmup):
        fn()
    ks.on = kb.on = True
    dt = ctx.timed(fn, steps, 0)
    kb.close()
    ks.close()
    n = ctx.rows * (ctx.seq - 5)
    t = dt / steps / ctx.micro
    tk = ks.avg_seconds() + kb.avg_seconds()
    gbs = 40.0 * n / tk / 1e9      # fwd: pred + target read (16 B); bwd: both again + dpred write (24 B)
    roof = {"bound": "hbm", "kernel": "k_stats + k_bwd (row FFT in registers / LDS), the two streaming kernels", "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_bytes("spectral") if default_shape(ctx) else None,
            "bytes_per_sample": 40, "kernel_ms": tk * 1e3, "whole_loss_frac": 40.0 * n / t / 1e9 / HBM_PEAK_GBS}
    return dt, roof, {"loss": float(val[0].item())}


# ------------------------------------------------------------------------------------------ CPU baselines
def host_info():
    import torch
    info = {"threads": torch.get_num_threads(), "host_cpus": os.cpu_count()}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            k, _, v = line.partition(":")
            if k.strip() in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core"):
                info[k.strip()] = v.strip()
    except (OSError, subprocess.SubprocessError):
        pass
    return info


def _cpu_rows(rows, seq):
    import torch
    g = torch.Generator().manual_seed(1)
    iq = (torch.randn(rows, seq, 2, generator=g) * 0.25).clamp(-0.7, 0.7)
    return iq, iq[:, 5:, :].contiguous()


def _gen_state():
    import torch
    from oracle import generator as og
    from oracle import steps as ost
    from dpd_amd.models import TDNNGeneratorQAT
    torch.manual_seed(0)
    params = {k: v for k, v in TDNNGeneratorQAT().state_dict().items() if k in og.PARAM_KEYS}
    return ost.GenState(params)


def _time_loop(fn, budget):
    fn()
    t0, reps = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget or reps == 0:
        fn()
        reps += 1
    return reps, time.perf_counter() - t0


def cpu_stage1(qat, seq, budget=6.0):
    """The CPU oracle (vectorised torch fp32, all host threads given to the job) on a bounded sample."""
    import torch
    from oracle import steps as ost
    rows = 256
    iq, target = _cpu_rows(rows, seq)
    gs = _gen_state()
    reps, dt = _time_loop(lambda: ost.stage1_step(gs, iq, target, qat), budget)
    return {"value": rows * (seq - 5) * reps / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} Stage-1 steps (qat={'on' if qat else 'off'}) of {rows} rows x {seq} samples "
                      f"({rows * (seq - 5)} outputs each) with oracle/steps.py, torch CPU fp32, ~{budget:.0f} s"}


def cpu_stage2(seq, budget=15.0):
    import torch
    from oracle import critic as oc
    from oracle import steps as ost
    from dpd_amd.models import Discriminator
    rows = 64
    iq, target = _cpu_rows(rows, seq)
    gs = _gen_state()
    torch.manual_seed(1)
    ds = ost.CriticState(oc.layers_from_state_dict(Discriminator().state_dict()))
    g = torch.Generator().manual_seed(2)
    alphas = [torch.rand(rows, 1, 1, generator=g) for _ in range(5)]
    reps, dt = _time_loop(lambda: ost.stage2_step(gs, ds, iq, target, target, alphas), budget)
    return {"value": rows * (seq - 5) * reps / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} Stage-2 CWGAN-GP steps (5 critic updates + generator update, qat=on) of {rows} rows x {seq} samples "
                      f"({rows * (seq - 5)} outputs each) with oracle/steps.py, torch CPU fp32 autograd, ~{budget:.0f} s"}


def cpu_q88(seq, budget=5.0):
    import numpy as np
    from oracle import q88 as oq
    from oracle import fixed_point as ofp
    rng = np.random.default_rng(0)
    rows = 16
    iq = rng.integers(-20000, 20000, size=(rows, seq, 2)).astype(np.int16)
    w = [rng.integers(-3000, 3000, size=s).astype(np.int16) for s in ((32, 30), (32,), (16, 32), (16,), (2, 16), (2,))]
    lut = ofp.tanh_lut().numpy().astype(np.int16)
    reps, dt = _time_loop(lambda: oq.infer_rows(iq, *w, lut), budget)
    return {"value": rows * (seq - 5) * reps / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"{reps} passes of {rows} rows x {seq} int16 samples with oracle/q88.py (numpy int64, one thread), ~{budget:.0f} s"}


def cpu_gen_pa(seq, budget=5.0, qat=True):
    import torch
    from oracle import generator as og
    from oracle import pa as opa
    rows = 256
    iq, _ = _cpu_rows(rows, seq)
    gs = _gen_state()
    cre, cim = opa.init_coeffs()

    def fn():
        with torch.no_grad():
            return opa.volterra_forward(og.gen_forward(iq, gs.p, qat=qat), cre, cim)
    reps, dt = _time_loop(fn, budget)
    return {"value": rows * (seq - 5) * reps / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} passes of {'QAT' if qat else 'float'} generator + PA twin over {rows} rows x {seq} samples with the vectorised oracle, ~{budget:.0f} s"}


def cpu_gen_pa_train(seq, budget=5.0):
    """MSE on the PA output with autograd through PA twin and QAT generator (the oracle the fused chain backward is tested
    against), gradients of the generator."""
    import torch
    from oracle import generator as og
    from oracle import pa as opa
    rows = 128
    iq, target = _cpu_rows(rows, seq)
    tgt = target[:, 4:, :] * 31.0
    gs = _gen_state()
    cre, cim = opa.init_coeffs()

    def fn():
        for p in gs.p.values():
            p.grad = None
        y = opa.volterra_forward(og.gen_forward(iq, gs.p, qat=True), cre, cim)
        torch.nn.functional.mse_loss(y, tgt).backward()
    reps, dt = _time_loop(fn, budget)
    return {"value": rows * (seq - 5) * reps / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} forward + backward passes of QAT generator -> PA twin -> MSE over {rows} rows x {seq} samples with the "
                      f"oracle under torch autograd, ~{budget:.0f} s"}


def cpu_gen_pa_literal(budget=6.0):
    """The reference's own loop structure (one Python iteration per output sample in the tap assembly, the
    triple loop of the memory polynomial) on 2^14 samples."""
    import torch
    from oracle import generator as og
    from oracle import literal as ol
    from oracle import pa as opa
    rows, seq = 64, 261
    iq, _ = _cpu_rows(rows, seq)
    gs = _gen_state()
    cre, cim = opa.init_coeffs()

    def fn():
        with torch.no_grad():
            feat = ol.tap_assembly_loop(iq)
            out = og.gen_forward(feat.reshape(-1, 30), gs.p, qat=True, pre_assembled=True).reshape(rows, seq - 5, 2)
            return ol.volterra_forward_loop(out, cre, cim)
    reps, dt = _time_loop(fn, budget)
    return {"value": rows * (seq - 5) * reps / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} passes of the literal loop form (per-sample tap loop, triple PA loop; oracle/literal.py) over {rows} rows x {seq} "
                      f"samples = {rows * (seq - 5)} outputs, ~{budget:.0f} s"}


def cpu_spectral(seq, budget=3.0):
    import torch
    from oracle import spectral as osp
    rows = 1024
    _, target = _cpu_rows(rows, seq)
    pred = (target * 0.9).requires_grad_(True)

    def fn():
        pred.grad = None
        osp.spectral_loss(pred, target)[0].backward()
    reps, dt = _time_loop(fn, budget)
    return {"value": rows * (seq - 5) * reps / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} SpectralLoss forward + backward passes over {rows} rows x {seq - 5} samples with oracle/spectral.py, ~{budget:.0f} s"}


CPU_BASELINES = {
    "stage2": lambda a: cpu_stage2(a.seq), "stage1": lambda a: cpu_stage1(False, a.seq), "stage1_qat": lambda a: cpu_stage1(True, a.seq),
    "q88": lambda a: cpu_q88(a.seq), "q88_stream": lambda a: cpu_q88(a.seq), "gen_pa": lambda a: cpu_gen_pa(a.seq),
    "gen_pa_float": lambda a: cpu_gen_pa(a.seq, qat=False), "gen_pa_train": lambda a: cpu_gen_pa_train(a.seq),
    "spectral": lambda a: cpu_spectral(a.seq),
}

RUNNERS = {
    "stage2": run_stage2,
    "stage1": lambda c, a, s, w: run_stage1(c, a, False, s, w),
    "stage1_qat": lambda c, a, s, w: run_stage1(c, a, True, s, w),
    "q88": run_q88, "q88_stream": run_q88_stream, "gen_pa": run_gen_pa,
    "gen_pa_float": lambda c, a, s, w: run_gen_pa(c, a, s, w, qat=False),
    "gen_pa_train": run_gen_pa_train, "spectral": run_spectral,
}
LABELS = {
    "stage1": "Stage-1 supervised MSE step: fused generator forward + MSE + backward, Adam; qat=off",
    "stage1_qat": "Stage-1 supervised MSE step with QAT on: 4 abs-max passes + fused forward/MSE/backward, Adam",
    "stage2": "Stage-2 CWGAN-GP step: 5 critic updates (Wasserstein + gradient penalty) + generator update with SpectralLoss; qat=on",
    "q88": "Q8.8 fixed-point DPD inference, int16 in/out",
    "q88_stream": "Q8.8 fixed-point DPD inference over a resident int16 stream, chunk by chunk",
    "gen_pa": "validation chain: QAT generator forward + PA digital twin (4 abs-max passes, then one fused kernel)",
    "gen_pa_float": "validation chain: float generator forward + PA digital twin in one kernel",
    "gen_pa_train": "generator -> PA twin with fused backward: MSE on the PA output, generator gradients, Adam; qat=on",
    "spectral": "SpectralLoss forward + backward",
}
EXTRAS = ("stage1", "stage1_qat", "q88", "q88_stream", "gen_pa", "gen_pa_float", "gen_pa_train", "spectral")
_QAT_DT = ("f32 (QAT generator forward products: exact 2-digit bf16 products of the 16-bit quantised integers; backward products: "
           "bf16 x 3-way split; fp32 accumulate)")
_F32_DT = "f32 (generator products: bf16 x 3-way split, fp32 accumulate)"
DTYPE = {"q88": "int16", "q88_stream": "int16",
         "stage2": "f32 (critic hidden-layer products and inner weight gradients: bf16 x 3-way split, fp32 accumulate; QAT generator "
                   "forward products: exact 2-digit bf16 products of the 16-bit quantised integers, its backward products bf16 x 3-way "
                   "split; fp32 accumulate)",
         "stage1": _F32_DT, "gen_pa_float": _F32_DT, "stage1_qat": _QAT_DT, "gen_pa": _QAT_DT, "gen_pa_train": _QAT_DT}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if os.environ.get("DPD_BENCH_WATCHDOG"):      # debugging aid: dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DPD_BENCH_WATCHDOG"]), exit=True)
    ctx = Ctx(args)
    if args.gpus != ctx.world and ctx.rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {ctx.world} rank(s); reporting n_gpus={ctx.world}", file=sys.stderr)
    dt, roof, aux = RUNNERS[args.workload](ctx, args, args.steps, args.warmup)
    res = None
    if ctx.rank == 0:
        res = {
            "metric": METRIC, "value": ctx.n_local * ctx.world * args.steps / dt, "unit": "samples/s", "n_gpus": ctx.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if ctx.strong else "weak", "vs_baseline": None, "dtype": DTYPE.get(args.workload, "f32"),
            "data": "synthetic",
            "config": {"workload": LABELS[args.workload], "name": args.workload, "rows_per_gpu": ctx.rows,
                       "seq_len": args.seq, "samples_per_gpu_per_step": ctx.n_local, "optimiser_steps_per_step": ctx.micro,
                       "parallelism": f"dp{ctx.world}"},
            "roofline": roof,
        }
        if ctx.strong:
            res["config"].update(global_samples=ctx.n_local * ctx.world, global_batch=ctx.rows * (args.seq - 5) * ctx.world)
        res.update(aux)
    if not args.no_extras and not ctx.strong:
        extra = {}
        for name in EXTRAS:
            if name == args.workload:
                continue
            n_local = ctx.n_local
            d, r, a = RUNNERS[name](ctx, args, 5, 2)          # (the stream workload sets its own samples per step)
            if ctx.rank == 0:
                extra[name] = {"value": ctx.n_local * ctx.world * 5 / d, "unit": "samples/s", "ms_per_step": d / 5 * 1e3,
                               "steps": 5, "workload": LABELS[name], "dtype": DTYPE.get(name, "f32"), "roofline": r, **a}
            ctx.n_local = n_local
        if ctx.rank == 0:
            res["extra"] = extra
    if ctx.rank == 0:
        if not args.no_cpu_baseline and ctx.world == 1:
            res["host"] = host_info()
            res["cpu_baseline"] = CPU_BASELINES[args.workload](args)
            if "extra" in res:
                for name in ("stage1", "stage1_qat", "q88", "gen_pa", "gen_pa_float", "gen_pa_train", "spectral"):
                    if name in res["extra"]:
                        res["extra"][name]["cpu_baseline"] = CPU_BASELINES[name](args)
                if "gen_pa" in res["extra"]:
                    res["extra"]["gen_pa"]["cpu_baseline_literal"] = cpu_gen_pa_literal()
        print(json.dumps(res), flush=True)
    if ctx.world > 1:
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
