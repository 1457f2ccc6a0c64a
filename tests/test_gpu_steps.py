"""Stage-1 and Stage-2 training steps on the GPU against the golden parameter updates captured from
the reference modules (one Adam step each; Stage 2 with n_critic = 5 and fixed alphas)."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, T, close, load_golden, sd

pytestmark = pytest.mark.gpu


def _gen(g, prefix, qat):
    from dpd_amd.models import TDNNGeneratorQAT
    gen = TDNNGeneratorQAT()
    gen.load_state_dict(sd(g, prefix))
    gen.qat_enabled = qat
    return gen.to(DEV).train()


@pytest.mark.parametrize("tag,qat", [("f", False), ("q", True)])
@pytest.mark.parametrize("impl", [1, 2])
def test_stage1_step_golden(tag, qat, impl):
    from dpd_amd.training import Stage1Step
    g = load_golden("g10_steps")
    gen = _gen(g, f"s1{tag}_g0__", qat)
    step = Stage1Step(gen, impl=impl)
    loss = step(T(g["iq"]), T(g["target"]))
    close(loss[0], g[f"s1{tag}_loss"], rtol=1e-5)
    ref = sd(g, f"s1{tag}_g1__")
    for k, v in gen.state_dict().items():
        close(v, ref[k], rtol=1e-5, atol=3e-7)


@pytest.mark.parametrize("critic_impl", [1, 2], ids=["valu", "mfma"])
def test_stage2_step_golden(critic_impl):
    from dpd_amd import _lib as L
    L.call("dpd_set_option", b"critic_impl", critic_impl)
    try:
        _stage2_step_golden()
    finally:
        L.call("dpd_set_option", b"critic_impl", 0)


def _stage2_step_golden():
    from dpd_amd.models import Discriminator
    from dpd_amd.training import Stage2Step
    g = load_golden("g10_steps")
    gen = _gen(g, "s2_g0__", True)
    disc = Discriminator()
    disc.load_state_dict(sd(g, "s2_d0__"))
    disc = disc.to(DEV).train()
    iq, target = T(g["iq"]), T(g["target"])
    cond = iq[:, 5:, :].contiguous()
    step = Stage2Step(gen, disc)
    info = step(iq, target, cond, alphas=T(g["s2_alphas"]))
    for it in range(5):
        close(info[f"d_total_{it}"], g["s2_d_totals"][it], rtol=1e-5, atol=1e-6)
    for k, ref in (("g_total", "s2_g_total"), ("g_adv", "s2_g_adv"), ("recon", "s2_recon"), ("spectral", "s2_spectral"),
                   ("evm", "s2_evm"), ("acpr", "s2_acpr")):
        close(info[k], g[ref], rtol=2e-5, atol=1e-6)
    g1, d2 = sd(g, "s2_g1__"), sd(g, "s2_d2__")
    for k, v in gen.state_dict().items():
        close(v, g1[k], rtol=1e-5, atol=5e-7)
    for k, v in disc.state_dict().items():
        close(v, d2[k], rtol=1e-4 if "weight_" in k and "orig" not in k else 1e-5, atol=1e-6)


def test_stage1_many_steps_stay_finite():
    """Dozens of steps at a size where every wave walks many chunks: the loss must stay finite and fall
    (guards the padding-slot operand reads of the MFMA kernels, which must never see stale LDS)."""
    from dpd_amd.data import ofdm_rows
    from dpd_amd.models import TDNNGeneratorQAT
    from dpd_amd.training import Stage1Step
    # leave non-finite fp32 bit patterns in LDS first: fp64 partial sums in the slab reduction do that
    for qat in (False, True):
        torch.manual_seed(0)
        gen = TDNNGeneratorQAT().to(DEV).train()
        gen.qat_enabled = qat
        iq = ofdm_rows(4096, 1029, seed=7, device=DEV)
        tgt = iq[:, 5:, :].contiguous()
        step = Stage1Step(gen)
        losses = torch.stack([step(iq, tgt).clone() for _ in range(24)]).flatten().cpu()
        assert torch.isfinite(losses).all()
        assert losses[-1] < losses[0]


def test_training_harness_two_stage_and_resume(tmp_path):
    """Two-stage schedule on synthetic rows: checkpoint dictionary and state-dict keys of the reference
    (train.py:396-427), finite metrics, resume continues from the saved epoch."""
    from dpd_amd.train import main
    out = tmp_path / "ck"
    args = ["--epochs", "2", "--stage1-epochs", "1", "--rows", "64", "--seq-len", "261", "--batch-size", "32",
            "--config", "", "--output-dir", str(out), "--temp", "all", "--qat"]
    m = main(args)
    assert all(np.isfinite(m[k]) for k in ("val_evm_db", "val_nmse_db", "val_l1", "train_loss")) and m["stage"] == 2
    ck = torch.load(out / "latest.pth", map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "generator_state_dict", "discriminator_state_dict", "g_optimizer_state_dict",
                       "d_optimizer_state_dict", "best_evm", "config"}
    assert ck["epoch"] == 1 and (out / "best.pth").exists()
    g_keys = set(ck["generator_state_dict"])
    assert {"fc1.weight", "fc3.bias", "input_quant.running_scale", "fc3_weight_quant.running_zero_point"} <= g_keys
    d_keys = set(ck["discriminator_state_dict"])
    assert {"features.0.linear.weight_orig", "features.4.linear.weight_u", "output.linear.weight_v", "output.linear.bias"} <= d_keys
    m2 = main(["--epochs", "3", "--stage1-epochs", "1", "--rows", "64", "--seq-len", "261", "--batch-size", "32",
               "--config", "", "--output-dir", str(out), "--resume", str(out / "latest.pth"), "--qat"])
    assert np.isfinite(m2["val_evm_db"])
    assert torch.load(out / "latest.pth", map_location="cpu", weights_only=True)["epoch"] == 2


def test_augment_rows_distributions():
    """Device-side augmentation: deterministic per seed, per-row draws, effects within the documented ranges."""
    from dpd_amd.data import augment_rows, ofdm_rows
    x = ofdm_rows(512, 261, seed=5, device=DEV)
    a = augment_rows(x, strength=0.3, seed=1)
    b = augment_rows(x, strength=0.3, seed=1)
    c = augment_rows(x, strength=0.3, seed=2)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.shape == x.shape and bool(torch.isfinite(a).all())
    # per-row gain: (1 +- 0.03) * thermal gain in {0.95, 1, 1.05} * compression (>= 1 - 0.1 * 0.49 * 1.02)
    ratio = (a.pow(2).sum((1, 2)) / x.pow(2).sum((1, 2))).sqrt()
    assert float(ratio.min()) > 0.95 * 0.97 * 0.94 and float(ratio.max()) < 1.05 * 1.03 * 1.001
    assert float(ratio.std()) > 0.01                       # rows got different draws
    # strength 0: only the thermal drift remains; a third of the rows are unchanged
    z = augment_rows(x, strength=0.0, seed=3)
    same = ((z - x).abs().amax((1, 2)) < 1e-6).float().mean()
    assert 0.2 < float(same) < 0.47


def test_generator_gradients_bitwise_reproducible_in_deterministic_mode():
    """dpd_set_option("gen_deterministic", 1): two launches of the fused Stage-1 kernel give identical bits;
    the default mode agrees with it to fp32 summation order."""
    from dpd_amd import _lib as L
    from dpd_amd.data import ofdm_rows
    from dpd_amd.models import TDNNGeneratorQAT
    from dpd_amd.training import Stage1Step
    torch.manual_seed(3)
    gen = TDNNGeneratorQAT().to(DEV)
    gen.qat_enabled = False
    iq = ofdm_rows(512, 1029, seed=2, device=DEV)
    tgt = iq[:, 5:, :].contiguous()

    def grads():
        step = Stage1Step(gen, optimizer=torch.optim.SGD(gen.parameters(), lr=0.0))
        step(iq, tgt)
        return step.grads[:1554].clone()
    L.call("dpd_set_option", b"gen_deterministic", 1)
    try:
        a, b = grads(), grads()
    finally:
        L.call("dpd_set_option", b"gen_deterministic", 0)
    c = grads()
    assert torch.equal(a, b)
    close(c, a.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(a.abs().max()))


@pytest.mark.gpu
def test_critic_gradients_bitwise_reproducible_in_deterministic_mode():
    """dpd_set_option("critic_deterministic", 1): tiles go to waves in a fixed order, two launches of the critic
    backward and of the gradient penalty give identical bits; the default (tiles handed out first come, first
    served) agrees to fp32 summation order."""
    from dpd_amd import _lib as L
    torch.manual_seed(5)
    B, Lr = 96, 1024
    N = B * Lr
    x = (torch.randn(N, 2, device=DEV) * 0.3).contiguous()
    y = (torch.randn(N, 2, device=DEV) * 0.3).contiguous()
    c = (torch.randn(N, 2, device=DEV) * 0.3).contiguous()
    alpha = torch.rand(B, device=DEV)
    wb = torch.randn(2945, device=DEV) * 0.2
    ws, wsp, wsn = L.workspace(DEV, N)
    st = L.stream()

    def run():
        g = torch.empty(2945, device=DEV)
        gg = torch.empty(2945, device=DEV)
        ssum = torch.zeros(1, device=DEV)
        gp = torch.zeros(1, device=DEV)
        rn = torch.empty(B, device=DEV)
        L.call("dpd_critic_backward", L.fptr(x), L.fptr(c), N, L.fptr(wb), 0.2, L.c_p(0), 1.0 / N, L.fptr(g), L.c_p(0),
               L.fptr(ssum), wsp, wsn, st)
        L.call("dpd_critic_gp", L.fptr(x), L.fptr(y), L.fptr(c), L.fptr(alpha), B, Lr, L.fptr(wb), 0.2, L.fptr(gp),
               L.fptr(rn), L.fptr(gg), wsp, wsn, st)
        torch.cuda.synchronize()
        return torch.cat([g, gg, ssum, gp])
    L.call("dpd_set_option", b"critic_deterministic", 1)
    try:
        a, b = run(), run()
    finally:
        L.call("dpd_set_option", b"critic_deterministic", 0)
    d = run()
    assert torch.equal(a, b)
    close(d, a.cpu().numpy(), rtol=1e-5, atol=2e-6 * float(a.abs().max()))
