"""Chain iq -> generator -> PA twin on the HIP path: the measured-data slice of the reference's validation
chain (fixture g9), the fused chain backward against autograd through the oracle, and the QAT form of the
fused forward entry point."""
import numpy as np
import pytest
import torch

import dpd_amd  # noqa: F401
from dpd_amd import _lib as L
from gpu_util import DEV, T, close, close_q, load_golden, ofdm_rows, sd

pytestmark = pytest.mark.gpu


def _generator(qat, seed=11):
    from oracle import generator as og
    from dpd_amd.models import TDNNGeneratorQAT
    torch.manual_seed(seed)
    gen = TDNNGeneratorQAT()
    with torch.no_grad():
        for fc in (gen.fc1, gen.fc2, gen.fc3):
            fc.bias.uniform_(-0.1, 0.1)
    params = {k: v.clone() for k, v in gen.state_dict().items() if k in og.PARAM_KEYS}
    gen = gen.to(DEV).eval()
    gen.qat_enabled = qat
    return gen, params


# ------------------------------------------------------------------ measured data (the CSV plumbing configuration)
def test_csv_slice_through_the_chain():
    """First 4,096 rows of the reference's data/test_input.csv, normalised as train.py:80-83, through
    TDNNGeneratorQAT (eval) -> PADigitalTwin(add_noise=False) -> compute_metrics against the delayed,
    gain-scaled input: generator output, PA output and the six metrics against what the reference's own
    modules gave for the same seed (fixture g9)."""
    from dpd_amd.models import PADigitalTwin, TDNNGeneratorQAT, dpd_pa_chain
    from dpd_amd.utils import SpectralLoss
    g = load_golden("g9_csv_chain")
    rows = g["csv_rows"]
    u = rows[:, 0] + 1j * rows[:, 1]
    u = (u * (0.7 / np.max(np.abs(u)))).astype(np.complex64)
    iq = T(np.stack([u.real, u.imag], -1).astype(np.float32)).unsqueeze(0)
    assert np.array_equal(iq.cpu().numpy(), g["iq"])
    torch.manual_seed(909)
    gen = TDNNGeneratorQAT().to(DEV).eval()
    pa = PADigitalTwin().to(DEV)
    dpd, y = dpd_pa_chain(gen, pa, iq)
    close_q(dpd, g["dpd"])
    scale = float(np.abs(g["y"]).max())
    # measured on MI355X: 1.4e-7 of the peak (no generator output lands on the other side of a rounding boundary here)
    close(y, g["y"], rtol=1e-5, atol=2e-6 * scale)
    # the module-by-module route gives the same
    with torch.no_grad():
        y_mod = pa(gen(iq), add_noise=False)
    close(y_mod, y.detach().cpu().numpy(), rtol=1e-5, atol=1e-5 * scale)
    ref_sig = (iq[:, 9:, :] * (10 ** (30 / 20))).contiguous()
    met = SpectralLoss().compute_metrics(y.detach(), ref_sig)
    for k, v in met.items():
        want = float(g["metric__" + k])
        if k.endswith("_db"):
            assert abs(v - want) < 0.1, (k, v, want)          # +-0.1 dB on EVM / NMSE / ACPR
        else:
            assert abs(v - want) < 1e-4 * max(1.0, abs(want)), (k, v, want)


# ------------------------------------------------------------------ fused chain backward
def _oracle_chain_grads(iq, params, qat, cre, cim, gy=None, target=None):
    from oracle import generator as og, pa as opa
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    y = opa.volterra_forward(og.gen_forward(iq, p, qat=qat), cre, cim)
    loss = torch.nn.functional.mse_loss(y, target) if target is not None else (y * gy).sum()
    loss.backward()
    return y.detach(), loss.detach(), torch.cat([p[k].grad.reshape(-1) for k in og.PARAM_KEYS])


CHAIN_SHAPES = [(8, 1029), (5, 901), (3, 2000), (70, 64), (74, 61), (2, 37), (1, 10)]


@pytest.mark.parametrize("qat", [False, True])
@pytest.mark.parametrize("shape", CHAIN_SHAPES)
def test_chain_backward_vs_oracle_autograd(qat, shape):
    """Gradient of a random linear functional of the PA output w.r.t. the six generator parameters, through
    dpd_pa_chain's autograd node (fused kernel for >= 4096 samples: pairs of tiles overlapping by 4 generator
    outputs, ragged last chunks and rows shorter than a chunk; separate kernels below), against torch autograd
    through the oracle's generator and memory polynomial."""
    from dpd_amd.models import PADigitalTwin, dpd_pa_chain
    B, S = shape
    gen, params = _generator(qat)
    gen.train()
    pa = PADigitalTwin().to(DEV)
    pa.set_temperature_state("cold")
    cre, cim = pa.pa_model.coeffs_real.cpu(), pa.pa_model.coeffs_imag.cpu()
    iq = ofdm_rows(B, S, seed=B * 31 + S)
    gy = torch.randn(B, S - 9, 2, generator=torch.Generator().manual_seed(5)) / (B * (S - 9))
    y_ref, _, g_ref = _oracle_chain_grads(iq, params, qat, cre, cim, gy=gy)
    _, y = dpd_pa_chain(gen, pa, iq.to(DEV), return_dpd=False)
    (y * gy.to(DEV)).sum().backward()
    got = torch.cat([p.grad.reshape(-1) for p in (gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias,
                                                  gen.fc3.weight, gen.fc3.bias)]).cpu()
    scale = float(y_ref.abs().max())
    # measured on MI355X over these shapes, as a fraction of the peak: float 6.5e-7, QAT 3.1e-5 (a generator output on
    # the other side of a rounding boundary is one 16-bit step, which the PA gain multiplies); gradients 2.3e-7 and
    # 2.8e-6 of the largest entry.  The allowances are about three times that.
    close(y, y_ref.numpy(), rtol=1e-5, atol=(2e-6 if not qat else 1e-4) * scale)
    close(got, g_ref.numpy(), rtol=1e-5, atol=(1e-6 if not qat else 1e-5) * float(g_ref.abs().max()))


@pytest.mark.parametrize("qat", [False, True])
@pytest.mark.parametrize("shape", [(8, 1029), (70, 64), (2, 37)])
def test_chain_mse_step_vs_oracle(qat, shape):
    """dpd_gen_pa_mse_step through the C ABI: loss and gradients of MSE(PA(G(iq)), target)."""
    from dpd_amd.models import PADigitalTwin
    B, S = shape
    gen, params = _generator(qat, seed=12)
    pa = PADigitalTwin().to(DEV)
    cre_d, cim_d = pa.pa_model.coeffs_real.contiguous(), pa.pa_model.coeffs_imag.contiguous()
    iq = ofdm_rows(B, S, seed=B + S)
    target = (iq[:, 9:, :] * 31.0 + 0.3 * torch.randn(B, S - 9, 2, generator=torch.Generator().manual_seed(6))).contiguous()
    _, loss_ref, g_ref = _oracle_chain_grads(iq, params, qat, cre_d.cpu(), cim_d.cpu(), target=target)
    x, tg = iq.to(DEV), target.to(DEV)
    st = L.stream()
    wq = torch.empty(L.GEN_NPARAM, device=DEV)
    absmax = torch.zeros(L.GEN_NSCALE, device=DEV)
    ps = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
    L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), int(qat), L.fptr(wq), L.fptr(absmax), st)
    if qat:
        out = torch.empty(B, S - 5, 2, device=DEV)
        for stage in range(4):
            L.call("dpd_gen_absmax_pass", stage, L.fptr(x), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.fptr(out), 0, st)
    res = torch.full((L.GEN_NPARAM + 1,), float("nan"), device=DEV)
    _, wsp, wsn = L.workspace(DEV, B * (S - 5))
    L.call("dpd_gen_pa_mse_step", L.fptr(x), B, S, L.fptr(tg), L.fptr(wq), 0.2, int(qat), L.fptr(absmax), L.fptr(cre_d),
           L.fptr(cim_d), 4, 5, L.fptr(res), L.fptr(res[L.GEN_NPARAM:]), wsp, wsn, st)
    res = res.cpu()
    # measured on MI355X: loss 1.1e-7 relative, gradients 1.9e-7 of the largest entry (float and QAT alike)
    close(res[L.GEN_NPARAM], loss_ref.numpy(), rtol=1e-6)
    close(res[:L.GEN_NPARAM], g_ref.numpy(), rtol=1e-5, atol=(1e-6 if not qat else 3e-6) * float(g_ref.abs().max()))


def test_chain_backward_other_pa_shapes_and_step():
    """PA twins the fused kernel is not built for (3 taps; 5 odd orders) take the separate kernels; ChainStep
    moves the parameters as Adam on the oracle's gradients does."""
    from oracle import pa as opa
    from dpd_amd.models import PADigitalTwin, dpd_pa_chain
    from dpd_amd.training import ChainStep, ADAM
    gen, params = _generator(False, seed=13)
    gen.train()
    iq = ofdm_rows(6, 1029, seed=2)
    for kw in (dict(memory_depth=3), dict(nonlinear_order=9), dict(nonlinear_order=5)):
        pa = PADigitalTwin(**kw).to(DEV)
        cre, cim = pa.pa_model.coeffs_real.cpu(), pa.pa_model.coeffs_imag.cpu()
        M = cre.shape[1]
        gy = torch.randn(6, 1024 - M + 1, 2, generator=torch.Generator().manual_seed(1)) / 6144
        _, _, g_ref = _oracle_chain_grads(iq, params, False, cre, cim, gy=gy)
        for p in gen.parameters():
            p.grad = None
        _, y = dpd_pa_chain(gen, pa, iq.to(DEV), return_dpd=False)
        (y * gy.to(DEV)).sum().backward()
        got = torch.cat([p.grad.reshape(-1) for p in (gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias,
                                                      gen.fc3.weight, gen.fc3.bias)]).cpu()
        close(got, g_ref.numpy(), rtol=1e-5, atol=1e-6 * float(g_ref.abs().max()))     # measured: 2.8e-7 of the largest entry
    # one optimiser step
    gen, params = _generator(True, seed=14)
    gen.train()
    pa = PADigitalTwin().to(DEV)
    target = (iq[:, 9:, :] * 31.0).contiguous()
    _, loss_ref, g_ref = _oracle_chain_grads(iq, params, True, *opa.init_coeffs(), target=target)
    from oracle import generator as og
    ref_p = [params[k].clone().requires_grad_(True) for k in og.PARAM_KEYS]
    opt = torch.optim.Adam(ref_p, **{k: v for k, v in ADAM.items() if k != "fused"})
    off = 0
    for p in ref_p:
        p.grad = g_ref[off:off + p.numel()].view_as(p).clone()
        off += p.numel()
    opt.step()
    step = ChainStep(gen, pa)
    loss = step(iq.to(DEV), target.to(DEV))
    close(loss, loss_ref.numpy().reshape(1), rtol=1e-5)
    for p, r in zip((gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias), ref_p):
        # Adam's first step moves every entry by lr * sign(g): entries whose tiny gradient differs in sign aside
        diff = (p.detach().cpu() - r.detach()).abs()
        assert (diff > 1e-6).float().mean() < 0.02, float((diff > 1e-6).float().mean())


# ------------------------------------------------------------------ QAT form of the fused forward entry point
@pytest.mark.parametrize("shape", [(8, 1029), (2, 37)])
def test_gen_pa_forward_qat_entry_point(shape):
    """dpd_gen_pa_forward with qat = 1 (scales from the four abs-max passes; one kernel that quantises the
    generator output on chip for >= 4096 samples, pass 3 + quantiser + PA kernel below) against
    dpd_pa_chain and the oracle."""
    from oracle import generator as og, pa as opa
    from dpd_amd.models import PADigitalTwin, dpd_pa_chain
    B, S = shape
    gen, params = _generator(True)
    pa = PADigitalTwin().to(DEV)
    cre, cim = pa.pa_model.coeffs_real.contiguous(), pa.pa_model.coeffs_imag.contiguous()
    x = ofdm_rows(B, S, seed=B * 5 + S).to(DEV)
    st = L.stream()
    wq = torch.empty(L.GEN_NPARAM, device=DEV)
    absmax = torch.zeros(L.GEN_NSCALE, device=DEV)
    ps = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
    L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), 1, L.fptr(wq), L.fptr(absmax), st)
    tmp = torch.empty(B, S - 5, 2, device=DEV)
    for stage in range(4):
        L.call("dpd_gen_absmax_pass", stage, L.fptr(x), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.fptr(tmp), 0, st)
    dpd = torch.full((B, S - 5, 2), float("nan"), device=DEV)
    y = torch.full((B, S - 9, 2), float("nan"), device=DEV)
    L.call("dpd_gen_pa_forward", L.fptr(x), B, S, L.fptr(wq), 0.2, 1, L.fptr(absmax), L.fptr(cre), L.fptr(cim), 4, 5,
           L.fptr(dpd), L.fptr(y), st)
    d_c, y_c = dpd_pa_chain(gen, pa, x)
    close_q(dpd, d_c.cpu().numpy())
    scale = float(y_c.abs().max())
    close(y, y_c.detach().cpu().numpy(), rtol=1e-5, atol=2e-6 * scale)       # measured: 3.4e-7 of the peak
    # the four scales against the oracle's: the abs-max of each quantiser input, to one unit in the last place
    d_ref = og.gen_forward(x.cpu(), params, qat=True)
    y_ref = opa.volterra_forward(d_ref, cre.cpu(), cim.cpu())
    close(y, y_ref.numpy(), rtol=1e-5, atol=1e-4 * scale)                       # measured: 3.0e-5 of the peak


def test_qat_forward_repeats_bit_for_bit_under_load():
    """The QAT forward kernels (bf16 matrix cores, VALU work issuing beside the matrix instructions of other waves) have
    no atomics and no order-dependent sums: at a size that fills every SIMD, repeated launches must give identical bits,
    and the generator output of the fused chain must equal the four-pass forward (guards instruction hazards that only
    show as a few wrong tiles per million under contention)."""
    from dpd_amd import _lib as L
    from dpd_amd.data import ofdm_rows
    from dpd_amd.models import PADigitalTwin, TDNNGeneratorQAT
    B, S = 4096, 1029
    iq = ofdm_rows(B, S, seed=77, device=DEV)
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(DEV).eval()
    pa = PADigitalTwin().to(DEV)
    ps = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
    wq = torch.empty(L.GEN_NPARAM, device=DEV)
    absmax = torch.zeros(L.GEN_NSCALE, device=DEV)
    st = L.stream()
    L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), 1, L.fptr(wq), L.fptr(absmax), st)
    out = torch.empty(B, S - 5, 2, device=DEV)
    for stage in range(4):
        L.call("dpd_gen_absmax_pass", stage, L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.fptr(out), 0, st)
    L.call("dpd_gen_forward", L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, 1, L.fptr(absmax), L.fptr(out), 2, st)
    ref = out.clone()
    cre, cim = pa.pa_model.coeffs_real.contiguous(), pa.pa_model.coeffs_imag.contiguous()
    yb = torch.empty(B, S - 9, 2, device=DEV)
    out2 = torch.empty(B, S - 5, 2, device=DEV)
    first = None
    for _ in range(12):
        L.call("dpd_gen_forward", L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, 1, L.fptr(absmax), L.fptr(out), 2, st)
        L.call("dpd_gen_pa_forward", L.fptr(iq), B, S, L.fptr(wq), 0.2, 1, L.fptr(absmax), L.fptr(cre), L.fptr(cim), 4, 5,
               L.fptr(out2), L.fptr(yb), st)
        assert torch.equal(out, ref) and torch.equal(out2, ref)
        if first is None:
            first = yb.clone()
        assert torch.equal(yb, first)


def test_qat_backward_repeats_bit_for_bit_under_load():
    """The fused QAT training kernels run their forward half on the bf16 matrix cores beside the VALU stages of other
    waves.  In deterministic mode (chunks in a fixed order, fixed summation order) repeated launches of the Stage-1 step
    kernel and of the chain step kernel must give identical gradients and losses, at a size that fills every SIMD."""
    from dpd_amd import _lib as L
    from dpd_amd.data import ofdm_rows
    from dpd_amd.models import PADigitalTwin, TDNNGeneratorQAT
    from dpd_amd.training import ChainStep, Stage1Step
    B, S = 4096, 1029
    iq = ofdm_rows(B, S, seed=78, device=DEV)
    tgt = iq[:, 5:, :].contiguous()
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(DEV).train()
    gen.qat_enabled = True
    pa = PADigitalTwin().to(DEV)
    L.call("dpd_set_option", b"gen_deterministic", 1)
    try:
        for make, target in ((lambda: Stage1Step(gen, optimizer=torch.optim.SGD(gen.parameters(), lr=0.0)), tgt),
                             (lambda: ChainStep(gen, pa, optimizer=torch.optim.SGD(gen.parameters(), lr=0.0)),
                              (tgt[:, 4:, :] * 31.0).contiguous())):
            step = make()
            first = None
            for _ in range(12):
                step(iq, target)
                g = step.grads[:L.GEN_NPARAM + 1].clone()
                assert bool(torch.isfinite(g).all())
                if first is None:
                    first = g
                assert torch.equal(g, first)
    finally:
        L.call("dpd_set_option", b"gen_deterministic", 0)


def test_float_kernels_repeat_bit_for_bit_under_load():
    """The float generator kernels run FC1 and FC2 as three-way split products on the bf16 matrix cores, VALU work issuing
    beside the matrix instructions: forward, fused chain and (in deterministic mode) the two fused training kernels must
    give identical bits on repeated launches at a size that fills every SIMD."""
    from dpd_amd import _lib as L
    from dpd_amd.data import ofdm_rows
    from dpd_amd.models import PADigitalTwin, TDNNGeneratorQAT
    from dpd_amd.training import ChainStep, Stage1Step
    B, S = 4096, 1029
    iq = ofdm_rows(B, S, seed=79, device=DEV)
    tgt = iq[:, 5:, :].contiguous()
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(DEV).train()
    gen.qat_enabled = False
    pa = PADigitalTwin().to(DEV)
    ps = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
    wq = torch.empty(L.GEN_NPARAM, device=DEV)
    absmax = torch.zeros(L.GEN_NSCALE, device=DEV)
    st = L.stream()
    L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), 0, L.fptr(wq), L.fptr(absmax), st)
    cre, cim = pa.pa_model.coeffs_real.contiguous(), pa.pa_model.coeffs_imag.contiguous()
    out, out2 = torch.empty(B, S - 5, 2, device=DEV), torch.empty(B, S - 5, 2, device=DEV)
    yb = torch.empty(B, S - 9, 2, device=DEV)
    first = None
    for _ in range(12):
        L.call("dpd_gen_forward", L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(out), 2, st)
        L.call("dpd_gen_pa_forward", L.fptr(iq), B, S, L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(cre), L.fptr(cim), 4, 5,
               L.fptr(out2), L.fptr(yb), st)
        if first is None:
            first = (out.clone(), yb.clone())
        assert torch.equal(out, first[0]) and torch.equal(out2, first[0]) and torch.equal(yb, first[1])
    L.call("dpd_set_option", b"gen_deterministic", 1)
    try:
        for make, target in ((lambda: Stage1Step(gen, optimizer=torch.optim.SGD(gen.parameters(), lr=0.0)), tgt),
                             (lambda: ChainStep(gen, pa, optimizer=torch.optim.SGD(gen.parameters(), lr=0.0)),
                              (tgt[:, 4:, :] * 31.0).contiguous())):
            step = make()
            ref = None
            for _ in range(12):
                step(iq, target)
                g = step.grads[:L.GEN_NPARAM + 1].clone()
                assert bool(torch.isfinite(g).all())
                if ref is None:
                    ref = g
                assert torch.equal(g, ref)
    finally:
        L.call("dpd_set_option", b"gen_deterministic", 0)
