"""Runnable two-stage training harness in sequence mode, on the HIP path.

Counterpart of the reference's train.py (CLI :430-451, optimisers :202-254, train_step :257-351,
validate :354-393, checkpoint dict :396-427) and of the notebook's Stage-1 (supervised) / Stage-2
(CWGAN-GP) schedule, rebuilt for the formulation the reference's modules were designed for: rows
[B, L+5, 2] in, [B, L, 2] out (the reference's own per-sample train_step cannot run for B > 1).

Indirect learning, as in the reference: the generator's input is the PA *output* (distorted), its
target the PA *input* (clean).  Data either comes from `<data-dir>/{train,val}_{input,output}.csv`
(columns I,Q; normalised by 0.7 / max|u| as train.py:80-83) cut into rows on the device, or is synthetic:
OFDM-like clean rows pushed through the PA twin of the chosen temperature state (both on the device).

One process per GPU under torch.distributed.run: rows are sharded, every loss term is that of the
global batch (parallel.py), gradients are all-reduced once per backward.  Metrics stay on the device
during training; validation reads its metrics once per validation batch (compute_metrics returns Python floats,
as the reference's does).
"""
import argparse
import csv
import os
from pathlib import Path
from typing import Dict, Tuple

import torch

DEFAULT_CONFIG = {
    "system": {"sample_rate": 200e6},
    "model": {"generator": {"memory_depth": 5, "hidden_dims": [32, 16], "leaky_slope": 0.2},
              "discriminator": {"input_dim": 4, "hidden_dims": [64, 32, 16], "leaky_slope": 0.2, "use_spectral_norm": True}},
    "training": {"batch_size": 256, "epochs": 4, "stage1_epochs": 2, "n_critic": 5, "gp_weight": 10.0,
                 "optimizer": {"lr_generator": 1e-4, "lr_discriminator": 1e-4, "betas": [0.0, 0.9], "weight_decay": 1e-5},
                 "loss": {"adversarial": 1.0, "reconstruction_l1": 50.0, "spectral_evm": 20.0, "spectral_acpr": 10.0},
                 "scheduler": {"type": "cosine", "min_lr": 1e-6}, "checkpoint_interval": 25},
}


def _coerce(node):
    """YAML 1.1 reads '1e-4' and '200e6' as strings: turn numeric strings into numbers, recursively."""
    if isinstance(node, dict):
        return {k: _coerce(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_coerce(v) for v in node]
    if isinstance(node, str):
        try:
            f = float(node)
            return int(f) if f.is_integer() and "e" not in node.lower() and "." not in node else f
        except ValueError:
            return node
    return node


def _merge(base, over):
    out = dict(base)
    for k, v in (over or {}).items():
        out[k] = _merge(base[k], v) if isinstance(v, dict) and isinstance(base.get(k), dict) else v
    return out


def load_config(path: str) -> dict:
    """The reference's YAML layout (config/config.yaml) over built-in defaults; a missing file gives the defaults."""
    if path and os.path.exists(path):
        import yaml
        with open(path) as f:
            return _merge(DEFAULT_CONFIG, _coerce(yaml.safe_load(f)))
    return _merge(DEFAULT_CONFIG, {})


# ------------------------------------------------------------------------------------- data
def _read_iq_csv(path: Path) -> torch.Tensor:
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    return torch.tensor([[float(r["I"]), float(r["Q"])] for r in rows], dtype=torch.float32)


def window_rows(x: torch.Tensor, seq_len: int) -> torch.Tensor:
    """Rows [B, seq_len, 2] of a stream [n, 2]: windows of seq_len = L + 5 samples that advance by L // 2 (50 % overlap),
    (n - 5 - L) // (L // 2) of them -- start indices and count of DPDDataset._prepare_sequences
    (utils/dataset.py:262-279) -- as one strided view made contiguous, without a Python loop."""
    L = seq_len - 5
    hop = max(L // 2, 1)
    count = (x.shape[0] - 5 - L) // hop
    if count < 1:
        raise ValueError(f"{x.shape[0]} samples are too few for one row of {seq_len} (need more than {seq_len + hop - 1})")
    return x.unfold(0, seq_len, hop)[:count].permute(0, 2, 1).contiguous()


def measured_rows(data_dir: Path, split: str, seq_len: int, device, temp: str = "normal") -> Tuple[torch.Tensor, torch.Tensor]:
    """(input rows [B, seq_len, 2] = PA output, target rows [B, seq_len-5, 2] = PA input), normalised by
    0.7 / max|u| (train.py:80-83), windows as `window_rows`.  `temp`: 'normal' leaves the PA output as measured, 'cold' /
    'hot' apply the thermal drift of -20 / 70 degrees C to it, 'all' stacks the three variants (train.py:497-513)."""
    from .data import TEMPERATURE_C, apply_thermal_drift
    u = _read_iq_csv(data_dir / f"{split}_input.csv").to(device)
    y = _read_iq_csv(data_dir / f"{split}_output.csv").to(device)
    n = min(u.shape[0], y.shape[0])
    scale = 0.7 / torch.sqrt((u[:n] ** 2).sum(-1)).max()
    u, y = u[:n] * scale, y[:n] * scale
    ur = window_rows(u, seq_len)[:, 5:, :].contiguous()
    states = ["cold", "normal", "hot"] if temp == "all" else [temp]
    yr = torch.cat([window_rows(y if st == "normal" else apply_thermal_drift(y, TEMPERATURE_C[st]), seq_len) for st in states])
    return yr, ur.repeat(len(states), 1, 1)


def shard_equal(x: torch.Tensor, t: torch.Tensor, batch_size: int, world: int, rank: int):
    """This rank's rows of a row set that every rank sees whole, and the batch size to use: the same number of
    rows and the same batch size on every rank (rows beyond a multiple of world * batch are dropped), so that all
    ranks run the same number of steps -- every step issues collectives, and a rank with one step more would wait
    in them for ever -- and the mean over ranks of equal-sized local means is the global mean."""
    n = x.shape[0]
    if n < world:
        raise ValueError(f"{n} rows cannot be split over {world} ranks")
    batch = min(batch_size, n // world)
    per_rank = (n // (world * batch)) * batch
    lo = rank * per_rank
    return x[lo:lo + per_rank], t[lo:lo + per_rank], batch


def synthetic_rows(rows: int, seq_len: int, temp: str, seed: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Clean OFDM-like rows through the PA twin of `temp` ('all': rows cycle through cold/normal/hot),
    scaled back by the twin's small-signal gain so that input and target have comparable level."""
    from .data import ofdm_rows
    from .models import PADigitalTwin
    pa = PADigitalTwin().to(device)
    M = pa.memory_depth
    clean = ofdm_rows(rows, seq_len + M - 1, seed=seed, device=device)
    states = ["cold", "normal", "hot"] if temp == "all" else [temp]
    out = torch.empty(rows, seq_len, 2, device=device)
    gain = 10.0 ** (pa.pa_model.gain_db / 20.0)
    with torch.no_grad():
        for i, st in enumerate(states):
            sel = slice(i, rows, len(states))
            out[sel] = pa(clean[sel].contiguous(), add_noise=False, temp_state=st) / gain
    return out.contiguous(), clean[:, M - 1 + 5:, :].contiguous()


# ------------------------------------------------------------------------------------- harness
def create_models(config: dict, qat: bool, device):
    from .models import TDNNGeneratorQAT, create_discriminator
    g, q = config["model"]["generator"], config.get("quantization", {})
    # the QAT class covers both modes: qat_enabled switches the quantisers (Stage 1 of the notebook runs without)
    gen = TDNNGeneratorQAT(memory_depth=g.get("memory_depth", 5), hidden_dims=g.get("hidden_dims", [32, 16]),
                           leaky_slope=g.get("leaky_slope", 0.2), weight_bits=q.get("weight", {}).get("bits", 16),
                           activation_bits=q.get("activation", {}).get("bits", 16))
    gen.qat_enabled = bool(qat)
    return gen.to(device), create_discriminator(config).to(device)


def create_optimizers(gen, disc, config: dict):
    o = config["training"]["optimizer"]
    kw = dict(betas=tuple(o.get("betas", [0.0, 0.9])), weight_decay=float(o.get("weight_decay", 1e-5)), fused=True)
    return (torch.optim.Adam(gen.parameters(), lr=float(o.get("lr_generator", 1e-4)), **kw),
            torch.optim.Adam(disc.parameters(), lr=float(o.get("lr_discriminator", 1e-4)), **kw))


def create_schedulers(g_opt, d_opt, config: dict, epochs: int):
    s = config["training"].get("scheduler", {})
    if s.get("type", "cosine") == "cosine":
        mk = lambda o: torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=epochs, eta_min=float(s.get("min_lr", 1e-6)))
    elif s.get("type") == "step":
        mk = lambda o: torch.optim.lr_scheduler.StepLR(o, step_size=int(s.get("step_size", 100)), gamma=float(s.get("gamma", 0.5)))
    else:
        return None, None
    return mk(g_opt), mk(d_opt)


@torch.no_grad()
def validate(gen, spectral_loss, inp: torch.Tensor, tgt: torch.Tensor, batch: int) -> Dict[str, float]:
    """EVM / NMSE / L1 of the generator output against the clean signal over the validation rows (batch means
    accumulated on the device: one host synchronisation per call)."""
    was_training = gen.training
    gen.eval()
    acc = torch.zeros(5, device=inp.device)
    n = 0
    for i in range(0, inp.shape[0], batch):
        acc += spectral_loss.metrics_tensor(gen(inp[i:i + batch]), tgt[i:i + batch])
        n += 1
    gen.train(was_training)
    acc = (acc / max(n, 1)).tolist()
    return {"val_evm_db": acc[0], "val_nmse_db": acc[1], "val_l1": acc[4]}


def save_checkpoint(gen, disc, g_opt, d_opt, epoch: int, best_evm: float, config: dict, ckpt_dir: Path, is_best: bool):
    """Same dictionary and file names as the reference (train.py:396-427)."""
    ck = {"epoch": epoch, "generator_state_dict": gen.state_dict(), "discriminator_state_dict": disc.state_dict(),
          "g_optimizer_state_dict": g_opt.state_dict(), "d_optimizer_state_dict": d_opt.state_dict(),
          "best_evm": best_evm, "config": config}
    torch.save(ck, ckpt_dir / "latest.pth")
    if (epoch + 1) % int(config["training"].get("checkpoint_interval", 25)) == 0:
        torch.save(ck, ckpt_dir / f"epoch_{epoch + 1}.pth")
    if is_best:
        torch.save(ck, ckpt_dir / "best.pth")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Train the GAN-DPD generator on MI355X (sequence mode)")
    p.add_argument("--config", type=str, default="config/config.yaml")
    p.add_argument("--temp", type=str, default="all", choices=["cold", "normal", "hot", "all"])
    p.add_argument("--epochs", type=int, default=None)
    p.add_argument("--batch-size", type=int, default=None, help="rows per step and rank")
    p.add_argument("--qat", action="store_true")
    p.add_argument("--resume", type=str, default=None)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--output-dir", type=str, default="checkpoints")
    p.add_argument("--output", type=str, default=None, help="also write the final checkpoint to this path")
    # additions of this harness
    p.add_argument("--data-dir", type=str, default=None, help="directory with {train,val}_{input,output}.csv; synthetic data if absent")
    p.add_argument("--seq-len", type=int, default=1029, help="row length L + 5 (L a power of two for the ACPR term)")
    p.add_argument("--rows", type=int, default=4096, help="synthetic training rows per rank")
    p.add_argument("--stage1-epochs", type=int, default=None, help="supervised epochs before the CWGAN-GP epochs")
    p.add_argument("--augment", type=float, default=0.0, help="per-epoch augmentation strength of the input rows (0 = off)")
    return p


def main(argv=None) -> Dict[str, float]:
    args = build_parser().parse_args(argv)
    from . import _lib as L  # noqa: F401  (fails here, loudly, when the library is missing)
    from .models import WassersteinLoss  # noqa: F401
    from .data import augment_rows
    from .parallel import DataParallel
    from .training import Stage1Step, Stage2Step
    from .utils.spectral_loss import create_spectral_loss

    config = load_config(args.config)
    tr = config["training"]
    if args.epochs:
        tr["epochs"] = args.epochs
    if args.batch_size:
        tr["batch_size"] = args.batch_size
    if args.stage1_epochs is not None:
        tr["stage1_epochs"] = args.stage1_epochs
    if not torch.cuda.is_available() or not args.device.startswith("cuda"):
        raise RuntimeError("this harness runs on a GPU only (there is no CPU path)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DPD_BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend -- runs the N > 1 code path on a one-GPU box
    rehearsal = os.environ.get("DPD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 and not torch.distributed.is_initialized():
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)
    dp = DataParallel()

    if args.data_dir:
        d = Path(args.data_dir)
        x_all, t_all = measured_rows(d, "train", args.seq_len, device, args.temp)
        x_tr, t_tr, batch = shard_equal(x_all, t_all, int(tr["batch_size"]), dp.world, dp.rank)
        x_va, t_va = measured_rows(d, "val" if (d / "val_output.csv").exists() else "train", args.seq_len, device)
    else:
        x_tr, t_tr = synthetic_rows(args.rows, args.seq_len, args.temp, 1234 + rank, device)
        x_va, t_va = synthetic_rows(max(args.rows // 8, 8), args.seq_len, args.temp, 99, device)
        batch = min(int(tr["batch_size"]), x_tr.shape[0])

    torch.manual_seed(0)
    gen, disc = create_models(config, args.qat, device)
    g_opt, d_opt = create_optimizers(gen, disc, config)
    epochs, s1_epochs = int(tr["epochs"]), int(tr.get("stage1_epochs", 0))
    g_sch, d_sch = create_schedulers(g_opt, d_opt, config, epochs)
    sl = create_spectral_loss(config)
    loss_w = tr.get("loss", {})
    stage1 = Stage1Step(gen, g_opt, grad_sync=dp.mean_ if dp.enabled else None, absmax_sync=dp.max_ if dp.enabled else None)
    stage2 = Stage2Step(gen, disc, g_opt, d_opt, n_critic=int(tr.get("n_critic", 5)), gp_weight=float(tr.get("gp_weight", 10.0)),
                        w_adv=float(loss_w.get("adversarial", 1.0)), w_l1=float(loss_w.get("reconstruction_l1", 50.0)),
                        spectral_loss=sl, dp=dp)

    ckpt_dir = Path(args.output_dir)
    if rank == 0:
        ckpt_dir.mkdir(parents=True, exist_ok=True)
    start, best = 0, float("inf")
    if args.resume:
        ck = torch.load(args.resume, map_location=device, weights_only=True)    # tensors, numbers, lists and dicts only
        gen.load_state_dict(ck["generator_state_dict"])
        disc.load_state_dict(ck["discriminator_state_dict"])
        g_opt.load_state_dict(ck["g_optimizer_state_dict"])
        d_opt.load_state_dict(ck["d_optimizer_state_dict"])
        start, best = int(ck["epoch"]) + 1, float(ck.get("best_evm", best))

    metrics: Dict[str, float] = {}
    gen.train()
    disc.train()
    for epoch in range(start, epochs):
        perm = torch.randperm(x_tr.shape[0], device=device, generator=torch.Generator(device=device).manual_seed(epoch))
        logs = []
        for i in range(0, x_tr.shape[0] - batch + 1, batch):
            idx = perm[i:i + batch]
            xb, tb = x_tr[idx], t_tr[idx]
            if args.augment > 0:
                xb = augment_rows(xb, args.augment, seed=epoch * 100003 + i + rank)
            if epoch < s1_epochs:
                logs.append(stage1(xb, tb).detach().reshape(1))
            else:
                info = stage2(xb, tb, xb[:, 5:, :].contiguous())
                logs.append(info["g_total"].detach().reshape(1))
        # the critic's schedule advances only in epochs in which the critic was updated (as the two-stage loop of
        # training_colab.ipynb code cell 14 does: scheduler_D.step() is skipped while pre-training)
        if g_sch is not None:
            g_sch.step()
        if d_sch is not None and epoch >= s1_epochs:
            d_sch.step()
        metrics = validate(gen, sl, x_va, t_va, batch)
        metrics["train_loss"] = float(torch.cat(logs).mean().item()) if logs else float("nan")
        metrics["stage"] = 1 if epoch < s1_epochs else 2
        is_best = metrics["val_evm_db"] < best
        best = min(best, metrics["val_evm_db"])
        if rank == 0:
            print(f"epoch {epoch + 1}/{epochs} stage {metrics['stage']} loss {metrics['train_loss']:.5f} "
                  f"val EVM {metrics['val_evm_db']:.2f} dB NMSE {metrics['val_nmse_db']:.2f} dB L1 {metrics['val_l1']:.5f}", flush=True)
            save_checkpoint(gen, disc, g_opt, d_opt, epoch, best, config, ckpt_dir, is_best)
    if args.output and rank == 0:
        Path(args.output).parent.mkdir(parents=True, exist_ok=True)
        torch.save({"epoch": epochs - 1, "generator_state_dict": gen.state_dict(), "discriminator_state_dict": disc.state_dict(),
                    "g_optimizer_state_dict": g_opt.state_dict(), "d_optimizer_state_dict": d_opt.state_dict(),
                    "best_evm": best, "config": config}, args.output)
    return metrics


if __name__ == "__main__":
    main()
