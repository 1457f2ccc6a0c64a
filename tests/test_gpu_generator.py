"""GPU parity of the generator path (tap assembly, float and QAT forward/backward) against the
golden fixtures and the CPU oracle.  Tolerances: tap indexing bit-exact; fp32 outputs rel <= 1e-5
(atol 1e-6 for values near zero); fake-quantised outputs within one 16-bit step."""
import os

import numpy as np
import pytest
import torch

from gpu_util import DEV, T, close, close_q, load_golden, ofdm_rows, sd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dpd():
    import dpd_amd
    return dpd_amd


def load_gen(dpd, params, qat_cls=True):
    from dpd_amd.models import TDNNGenerator, TDNNGeneratorQAT
    g = (TDNNGeneratorQAT if qat_cls else TDNNGenerator)()
    g.load_state_dict({k: v for k, v in params.items()}, strict=False)
    return g.to(DEV)


def test_taps_bit_exact(dpd):
    from dpd_amd.models import MemoryTapAssembly
    g = load_golden("g1_taps")
    mta = MemoryTapAssembly(5)
    assert np.array_equal(mta(T(g["probe"])).cpu().numpy(), g["feat_probe"])
    f = mta(T(g["iq"])).cpu().numpy()
    mask = np.ones(30, bool); mask[14:20] = False
    assert np.array_equal(f[..., mask], g["feat"][..., mask])
    np.testing.assert_allclose(f[..., 14:20], g["feat"][..., 14:20], rtol=3e-7, atol=0)
    with pytest.raises(ValueError):
        mta(torch.zeros(1, 5, 2, device=DEV))


@pytest.mark.parametrize("impl", [1, 2])
def test_float_generator_golden(dpd, impl):
    from oracle import generator as og
    g = load_golden("g2_float_gen")
    gen = load_gen(dpd, sd(g, "sd__"), qat_cls=False)
    gen.kernel_impl = impl
    out = gen(T(g["iq"]))
    close(out, g["out"], rtol=1e-5, atol=1e-6)
    loss = torch.nn.functional.mse_loss(out, T(g["target"]))
    loss.backward()
    for k in og.PARAM_KEYS:
        mod, nm = k.split(".")
        close(getattr(getattr(gen, mod), nm).grad, g["grad__" + k.replace(".", "__")], rtol=2e-5, atol=1e-7)
    if impl == 1:
        close(gen(T(g["feats"]), pre_assembled=True), g["out_pre"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("impl", [1, 2])
def test_qat_generator_golden(dpd, impl):
    from oracle import generator as og
    g = load_golden("g3_qat_gen")
    params = {k: v for k, v in sd(g, "sd0__").items()}
    gen = load_gen(dpd, params)
    gen.kernel_impl = impl
    gen.train()
    out1 = gen(T(g["iq"]))
    close_q(out1, g["out1"])
    loss = torch.nn.functional.mse_loss(out1, T(g["target"]))
    loss.backward()
    close(loss, g["loss"], rtol=1e-5)
    for k in og.PARAM_KEYS:
        mod, nm = k.split(".")
        close(getattr(getattr(gen, mod), nm).grad, g["grad__" + k.replace(".", "__")], rtol=3e-5, atol=1e-7)
    sd1 = gen.state_dict()
    for k in gen.ACT_QUANT + gen.WEIGHT_QUANT:
        close(sd1[k + ".running_scale"], g[f"sd1__{k}__running_scale"], rtol=1e-6, atol=0)
    close_q(gen(T(g["iq2"])), g["out2"])
    for k in gen.ACT_QUANT + gen.WEIGHT_QUANT:
        close(gen.state_dict()[k + ".running_scale"], g[f"sd2__{k}__running_scale"], rtol=1e-6, atol=0)
    assert set(gen.state_dict().keys()) == {k[len("sd2__"):].replace("__", ".") for k in g.files if k.startswith("sd2__")}
    gen.eval()
    close_q(gen(T(g["iq"])), g["out_eval"])
    for k in gen.ACT_QUANT:
        close(gen.state_dict()[k + ".running_scale"], g[f"sd3__{k}__running_scale"], rtol=1e-6, atol=0)
    close_q(gen(T(g["iq"][:1])), g["out_row0"])
    gen.disable_qat()
    close(gen(T(g["iq"])), g["out_noq"], rtol=1e-5, atol=1e-6)
    gen.enable_qat()
    if impl == 1:
        close_q(gen(T(g["feats"]), pre_assembled=True), g["out_pre"])
    qw = gen.get_quantized_weights()
    for k, v in qw.items():
        assert np.array_equal(v.cpu().numpy(), g["qw__" + k]), k


def test_dynamic_fake_quantize(dpd):
    from dpd_amd.models import fake_quantize
    g = load_golden("g3_qat_gen")
    x = T(g["fq_x"]).requires_grad_(True)
    y = fake_quantize(x, 16)
    assert np.array_equal(y.detach().cpu().numpy(), g["fq_tensor"])
    y.sum().backward()
    assert torch.equal(x.grad, torch.ones_like(x))
    assert np.array_equal(fake_quantize(T(g["fq_x"]), 16, per_channel=True, channel_dim=0).cpu().numpy(), g["fq_channel"])


def test_init_matches_reference_rng(dpd):
    """Same seed, same construction order -> same initial parameters as the reference classes."""
    from dpd_amd.models import TDNNGeneratorQAT
    g = load_golden("g9_csv_chain")
    torch.manual_seed(909)
    gen = TDNNGeneratorQAT()
    ref = sd(g, "sd__")
    for k, v in gen.state_dict().items():
        assert torch.equal(v, ref[k]), k


@pytest.mark.parametrize("qat", [False, True])
@pytest.mark.parametrize("shape", [(3, 21), (5, 133), (2, 1029), (1, 70)])
def test_mfma_vs_valu_vs_oracle_ragged(dpd, qat, shape):
    """Ragged row lengths (tiles and chunks partly empty): both kernel families against the oracle."""
    from oracle import generator as og
    from dpd_amd.models import TDNNGeneratorQAT
    B, S = shape
    torch.manual_seed(7)
    gen = TDNNGeneratorQAT()
    with torch.no_grad():
        for fc in (gen.fc1, gen.fc2, gen.fc3):
            fc.bias.uniform_(-0.1, 0.1)
    params = {k: v.clone() for k, v in gen.state_dict().items() if k in og.PARAM_KEYS}
    gen = gen.to(DEV)
    gen.qat_enabled = qat
    iq = ofdm_rows(B, S, seed=B * 1000 + S)
    tgt = torch.randn(B, S - 5, 2, generator=torch.Generator().manual_seed(3)) * 0.3
    ref_out, ref_loss, ref_grads = og.gen_mse_grads(iq, tgt, params, qat=qat)
    for impl in (1, 2):
        gen.kernel_impl = impl
        gen.zero_grad()
        out = gen(iq.to(DEV))
        (close_q if qat else close)(out, ref_out)
        loss = torch.nn.functional.mse_loss(out, tgt.to(DEV))
        loss.backward()
        for k in og.PARAM_KEYS:
            mod, nm = k.split(".")
            close(getattr(getattr(gen, mod), nm).grad, ref_grads[k], rtol=3e-4, atol=2e-7)


@pytest.mark.parametrize("qat", [False, True])
def test_fused_mse_step_matches_autograd_path(dpd, qat):
    """dpd_gen_mse_step (forward + MSE + backward in one kernel) against the oracle."""
    from oracle import generator as og
    from dpd_amd import _lib as L
    from dpd_amd.models import TDNNGeneratorQAT
    torch.manual_seed(11)
    gen = TDNNGeneratorQAT()
    params = {k: v.clone() for k, v in gen.state_dict().items() if k in og.PARAM_KEYS}
    B, S = 16, 1029
    iq = ofdm_rows(B, S, seed=99)
    tgt = iq[:, 5:, :].contiguous() * 0.9
    _, ref_loss, ref_grads = og.gen_mse_grads(iq, tgt, params, qat=qat)
    ref_flat = torch.cat([ref_grads[k].reshape(-1) for k in og.PARAM_KEYS])
    for impl in (1, 2):
        dev = torch.device(DEV)
        p = [params[k].to(dev) for k in og.PARAM_KEYS]
        wq = torch.empty(L.GEN_NPARAM, device=dev); absmax = torch.zeros(8, device=dev)
        grads = torch.empty(L.GEN_NPARAM, device=dev); loss = torch.zeros(1, device=dev)
        x, t = iq.to(dev), tgt.to(dev)
        st = L.stream()
        L.call("dpd_gen_prepare_weights", *(L.fptr(w) for w in p), int(qat), L.fptr(wq), L.fptr(absmax), st)
        if qat:
            out = torch.empty(B, S - 5, 2, device=dev)
            for stage in range(4):
                L.call("dpd_gen_absmax_pass", stage, L.fptr(x), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.fptr(out), impl, st)
        _, wsp, wsn = L.workspace(dev)
        L.call("dpd_gen_mse_step", L.fptr(x), B, S, L.fptr(t), L.fptr(wq), 0.2, int(qat), L.fptr(absmax), L.fptr(grads),
               L.fptr(loss), wsp, wsn, impl, st)
        close(loss[0], ref_loss, rtol=1e-5)
        close(grads, ref_flat, rtol=3e-5, atol=2e-8)


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("shape", [(4, 69), (8, 1029), (70, 64)])
def test_qat_scales_match_oracle(dpd, impl, shape):
    """The seven abs-max values behind the quantiser scales (features, act1, act2, tanh output, W1, W2, W3) against
    the oracle's: the feature and weight maxima bit for bit, the activation maxima to four units in the last place
    of fp32 (a 30- or 32-term dot product summed in another order moves the largest pre-activation by that much;
    measured: 0 to 3)."""
    from oracle import generator as og
    from dpd_amd.models import TDNNGeneratorQAT
    B, S = shape
    torch.manual_seed(21)
    gen = TDNNGeneratorQAT()
    with torch.no_grad():
        for fc in (gen.fc1, gen.fc2, gen.fc3):
            fc.bias.uniform_(-0.1, 0.1)
    params = {k: v.clone() for k, v in gen.state_dict().items() if k in og.PARAM_KEYS}
    gen = gen.to(DEV).train()
    gen.kernel_impl = impl
    iq = ofdm_rows(B, S, seed=S)
    ref = {}
    og.gen_forward(iq, params, qat=True, absmax_out=ref)
    gen(iq.to(DEV))
    got = gen._last_absmax.cpu().numpy()
    keys = og.ACT_QUANT_KEYS + og.WEIGHT_QUANT_KEYS
    for i, k in enumerate(keys):
        want = float(ref[k])
        ulps = abs(float(got[i]) - want) / np.spacing(np.float32(want))
        limit = 0 if k in ("input_quant",) + tuple(og.WEIGHT_QUANT_KEYS) else 4
        assert ulps <= limit, (k, float(got[i]), want, ulps)


# ------------------------------------------------------------------ gradient w.r.t. the IQ rows
def test_tap_assembly_gradient_vs_oracle(dpd):
    """MemoryTapAssembly under autograd: gradient of a random linear functional of the 30 columns w.r.t. the IQ rows
    (ragged row length, first and last points have fewer than six readers), against autograd through the oracle."""
    from oracle import taps as ot
    from dpd_amd.models import MemoryTapAssembly
    for B, S in ((3, 6), (2, 40), (5, 261)):
        iq = torch.randn(B, S, 2, generator=torch.Generator().manual_seed(100 + S)) * 0.3
        gf = torch.randn(B, S - 5, 30, generator=torch.Generator().manual_seed(S))
        xr = iq.clone().requires_grad_(True)
        (ot.tap_assembly(xr) * gf).sum().backward()
        x = iq.to(DEV).requires_grad_(True)
        (MemoryTapAssembly(5)(x) * gf.to(DEV)).sum().backward()
        close(x.grad, xr.grad, rtol=1e-5, atol=2e-6 * float(xr.grad.abs().max()))
    # a point at the origin: the envelope columns have no finite derivative there (torch gives nan); the kernel drops them
    iq = ofdm_rows(1, 20, seed=1)
    iq[0, 7] = 0.0
    x = iq.to(DEV).requires_grad_(True)
    MemoryTapAssembly(5)(x)[..., :2].sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad[0, 7, 0]) == 1.0


@pytest.mark.parametrize("qat", [False, True])
@pytest.mark.parametrize("shape", [(3, 21), (2, 1029), (40, 133)])
def test_generator_gradient_wrt_iq_rows(dpd, qat, shape):
    """The generator as a differentiable function of its IQ rows (the reference's is plain torch; the fake quantisers pass
    the gradient through, utils/quantization.py:164-173 and models/tdnn_generator.py:46-103): dx and the parameter
    gradients of the same backward, against autograd through the oracle."""
    from oracle import generator as og
    from dpd_amd.models import TDNNGeneratorQAT
    B, S = shape
    torch.manual_seed(21)
    gen = TDNNGeneratorQAT()
    with torch.no_grad():
        for fc in (gen.fc1, gen.fc2, gen.fc3):
            fc.bias.uniform_(-0.1, 0.1)
    params = {k: v.clone().requires_grad_(True) for k, v in gen.state_dict().items() if k in og.PARAM_KEYS}
    gen = gen.to(DEV).train()
    gen.qat_enabled = qat
    iq = ofdm_rows(B, S, seed=B + 7 * S) + 1e-3      # no point exactly at the origin (torch's gradient there is nan)
    gy = torch.randn(B, S - 5, 2, generator=torch.Generator().manual_seed(3)) / (B * (S - 5))
    xr = iq.clone().requires_grad_(True)
    (og.gen_forward(xr, params, qat=qat) * gy).sum().backward()
    x = iq.to(DEV).requires_grad_(True)
    (gen(x) * gy.to(DEV)).sum().backward()
    close(x.grad, xr.grad, rtol=1e-5, atol=2e-6 * float(xr.grad.abs().max()))
    got = torch.cat([p.grad.reshape(-1) for p in (gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias)])
    ref = torch.cat([params[k].grad.reshape(-1) for k in og.PARAM_KEYS])
    close(got, ref, rtol=1e-5, atol=(2e-6 if not qat else 1e-5) * float(ref.abs().max()))


@pytest.mark.parametrize("qat", [False, True])
def test_chain_gradient_wrt_iq_rows(dpd, qat):
    """dpd_pa_chain with IQ rows that ask for a gradient: dL/d iq through PA twin and generator against the oracle."""
    from oracle import generator as og, pa as opa
    from dpd_amd.models import PADigitalTwin, TDNNGeneratorQAT, dpd_pa_chain
    B, S = 6, 1029
    torch.manual_seed(22)
    gen = TDNNGeneratorQAT()
    params = {k: v.clone().requires_grad_(True) for k, v in gen.state_dict().items() if k in og.PARAM_KEYS}
    gen = gen.to(DEV).train()
    gen.qat_enabled = qat
    pa = PADigitalTwin().to(DEV)
    cre, cim = pa.pa_model.coeffs_real.cpu(), pa.pa_model.coeffs_imag.cpu()
    iq = ofdm_rows(B, S, seed=5) + 1e-3
    gy = torch.randn(B, S - 9, 2, generator=torch.Generator().manual_seed(4)) / (B * (S - 9))
    xr = iq.clone().requires_grad_(True)
    (opa.volterra_forward(og.gen_forward(xr, params, qat=qat), cre, cim) * gy).sum().backward()
    x = iq.to(DEV).requires_grad_(True)
    _, y = dpd_pa_chain(gen, pa, x, return_dpd=False)
    (y * gy.to(DEV)).sum().backward()
    close(x.grad, xr.grad, rtol=1e-5, atol=(2e-6 if not qat else 5e-6) * float(xr.grad.abs().max()))
    got = torch.cat([p.grad.reshape(-1) for p in (gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias)])
    ref = torch.cat([params[k].grad.reshape(-1) for k in og.PARAM_KEYS])
    close(got, ref, rtol=1e-5, atol=(2e-6 if not qat else 5e-6) * float(ref.abs().max()))


def test_running_scales_in_one_launch_match_the_module_arithmetic():
    """dpd_qat_observe (all seven running scales, `times` repeats, one launch) against FakeQuantizeModule.observe
    applied element by element on the CPU (the reference's arithmetic, models/tdnn_generator.py:121-130)."""
    from dpd_amd.models import TDNNGeneratorQAT
    torch.manual_seed(4)
    gen = TDNNGeneratorQAT().to(DEV).train()
    absmax = (torch.rand(8, device=DEV) * 3 + 1e-3)
    absmax[2] = 0.0                                  # below the clamp
    start = [float(getattr(gen, n).running_scale) for n in gen.ACT_QUANT + gen.WEIGHT_QUANT]
    gen.observe_all(absmax)
    gen.observe_all(absmax, times=4)
    got = torch.stack([getattr(gen, n).running_scale for n in gen.ACT_QUANT + gen.WEIGHT_QUANT]).cpu()
    a = absmax[:7].cpu()
    rs = torch.tensor(start, dtype=torch.float32)
    for _ in range(5):
        scale = torch.clamp(a, min=1e-8) / 32767
        rs = (1 - 0.1) * rs + 0.1 * scale
    close(got, rs.numpy(), rtol=3e-7, atol=0)
    # the buffers are views of one tensor now and survive a state-dict round trip
    sd = {k: v.clone() for k, v in gen.state_dict().items()}
    gen2 = TDNNGeneratorQAT().to(DEV)
    gen2.load_state_dict(sd)
    assert torch.equal(gen2.input_quant.running_scale, gen.input_quant.running_scale)
    gen2.train()
    gen2.observe_all(absmax)
    gen.observe_all(absmax)
    assert torch.equal(gen2.fc3_weight_quant.running_scale, gen.fc3_weight_quant.running_scale)


def test_mfma_generator_accuracy_against_fp64(dpd):
    """The MFMA family of the float generator runs every matrix product as six bf16 products of three-way split
    operands (fp32 accumulate).  Output and parameter gradients must stay at fp32 rounding level against an fp64
    evaluation, next to the VALU family's plain fp32 FMAs (measured, of the largest magnitude, VALU / MFMA: output
    3.4e-7 / 3.1e-7; gradients 5e-8 / 3.1e-7 -- the MFMA family keeps a wave's gradient sums over thousands of samples
    in fp32 accumulators, the VALU family hands shorter partial sums to the fp64 reduction; the bounds leave a factor of
    about 4)."""
    from oracle import generator as og
    from dpd_amd.data import ofdm_rows
    from dpd_amd.models import TDNNGenerator
    torch.manual_seed(11)
    iq = ofdm_rows(64, 1029, seed=21, device=DEV)
    target = (iq[:, 5:, :] * 0.8).contiguous()
    gen = TDNNGenerator().to(DEV)
    p64 = {k: v.detach().double().cpu() for k, v in gen.state_dict().items() if k in og.PARAM_KEYS}
    out64, _, g64 = og.gen_mse_grads(iq.double().cpu(), target.double().cpu(), p64)
    errs = {}
    for impl in (1, 2):
        gen.kernel_impl = impl
        gen.zero_grad()
        out = gen(iq)
        torch.nn.functional.mse_loss(out, target).backward()
        e_out = float((out.detach().double().cpu() - out64).abs().max() / out64.abs().max())
        e_g = max(float((dict(gen.named_parameters())[k].grad.double().cpu() - g64[k]).abs().max() / g64[k].abs().max())
                  for k in og.PARAM_KEYS)
        errs[impl] = (e_out, e_g)
    if os.environ.get("DPD_CLOSE_REPORT"):
        print("generator error against fp64 (output, gradients): VALU %.2e %.2e, MFMA %.2e %.2e" % (errs[1] + errs[2]))
    assert errs[2][0] < 1.2e-6 and errs[2][1] < 1.2e-6
    assert errs[2][0] < 2 * errs[1][0] + 1e-7
