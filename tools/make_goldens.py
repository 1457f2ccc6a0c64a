#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference implementation.

Runs ONLY in the build container, where the reference checkout is mounted at
/root/reference.  It imports the reference's Python modules (CPU, torch fp32),
feeds them seeded inputs and writes small ``.npz`` fixtures to ``tests/golden``.
The fixtures are data only: inputs, explicit weights and the outputs the
reference produced.  Nothing under tests/, bench.py or the package reads the
reference at run time.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py
"""
import importlib.util
import io
import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from models.tdnn_generator import (  # noqa: E402
    MemoryTapAssembly, TDNNGenerator, TDNNGeneratorQAT, fake_quantize as ref_dyn_fake_quantize,
)
from models.pa_digital_twin import PADigitalTwin, VolterraPA  # noqa: E402
from models.discriminator import (  # noqa: E402
    Discriminator, WassersteinLoss, compute_gradient_penalty,
)


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# utils/__init__.py pulls in h5py (absent here); these two files have no intra-package imports.
ref_spec = _load("ref_spectral_loss", "utils/spectral_loss.py")
ref_quant = _load("ref_quantization", "utils/quantization.py")

META = dict(torch_version=np.array(torch.__version__), generator=np.array("tools/make_goldens.py"))


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().copy()   # copy: buffers are later updated in place
    return np.array(t)


def sd_np(module, prefix=""):
    return {prefix + k.replace(".", "__"): npy(v) for k, v in module.state_dict().items()}


def save(name, **arrays):
    arrays.update(META)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB, {len(arrays)} arrays")


def ofdm_rows(B, S, seed, peak=0.7):
    """OFDM-like rows: QPSK on 1/4 of the bins around DC, 5-sample cyclic prefix."""
    g = torch.Generator().manual_seed(seed)
    L = S - 5
    nsc = L // 4
    bits = torch.randint(0, 4, (B, nsc), generator=g)
    sym = torch.exp(1j * (np.pi / 4 + np.pi / 2 * bits.to(torch.float32)))
    spec = torch.zeros(B, L, dtype=torch.complex64)
    idx = (torch.arange(nsc) - nsc // 2) % L
    spec[:, idx] = sym.to(torch.complex64)
    x = torch.fft.ifft(spec, dim=-1)
    x = torch.cat([x[:, -5:], x], dim=1)
    x = x / x.abs().amax(dim=1, keepdim=True) * peak
    return torch.stack([x.real, x.imag], dim=-1).to(torch.float32).contiguous()


# ----------------------------------------------------------------------------- g1
def g1_taps():
    torch.manual_seed(101)
    mta = MemoryTapAssembly(5)
    iq = torch.randn(2, 37, 2)
    feat = mta(iq)
    # index probe: distinct, exactly representable values; 3-4-5 triangles keep the envelope exact
    k = torch.arange(1, 15, dtype=torch.float32)
    probe = torch.stack([3 * k, 4 * k], dim=-1).unsqueeze(0) / 64.0
    feat_probe = mta(probe)
    ar = (torch.arange(1, 29, dtype=torch.float32) / 10).reshape(1, 14, 2)
    feat_ar = mta(ar)
    save("g1_taps", iq=npy(iq), feat=npy(feat), probe=npy(probe), feat_probe=npy(feat_probe),
         arange_iq=npy(ar), feat_arange=npy(feat_ar))


# ----------------------------------------------------------------------------- g2
def g2_float_gen():
    torch.manual_seed(202)
    g = TDNNGenerator()
    with torch.no_grad():  # non-zero biases so the bias path is exercised
        for lin in (g.fc1, g.fc2, g.fc3):
            lin.bias.uniform_(-0.1, 0.1)
    iq = torch.randn(4, 69, 2) * 0.4
    target = torch.randn(4, 64, 2) * 0.3
    out = g(iq)
    loss = torch.nn.functional.mse_loss(out, target)
    grads = torch.autograd.grad(loss, list(g.parameters()))
    feats = torch.randn(64, 30) * 0.5
    out_pre = g(feats, pre_assembled=True)
    arrays = sd_np(g, "sd__")
    for (n, _), gr in zip(g.named_parameters(), grads):
        arrays["grad__" + n.replace(".", "__")] = npy(gr)
    save("g2_float_gen", iq=npy(iq), target=npy(target), out=npy(out), loss=npy(loss),
         feats=npy(feats), out_pre=npy(out_pre), **arrays)


# ----------------------------------------------------------------------------- g3
def g3_qat_gen():
    torch.manual_seed(303)
    g = TDNNGeneratorQAT()
    with torch.no_grad():
        for lin in (g.fc1, g.fc2, g.fc3):
            lin.bias.uniform_(-0.05, 0.05)
    arrays = sd_np(g, "sd0__")
    iq = ofdm_rows(4, 69, seed=31)
    target = torch.randn(4, 64, 2) * 0.3
    g.train()
    out1 = g(iq)
    loss = torch.nn.functional.mse_loss(out1, target)
    params = [p for _, p in g.named_parameters()]
    grads = torch.autograd.grad(loss, params)
    for (n, _), gr in zip(g.named_parameters(), grads):
        arrays["grad__" + n.replace(".", "__")] = npy(gr)
    arrays.update(sd_np(g, "sd1__"))
    iq2 = ofdm_rows(4, 69, seed=32) * 1.3
    out2 = g(iq2)
    arrays.update(sd_np(g, "sd2__"))
    g.eval()
    out_eval = g(iq)
    arrays.update(sd_np(g, "sd3__"))  # buffers unchanged in eval
    # batch dependence: first row alone
    out_row0 = g(iq[:1])
    # QAT disabled == float path
    g.disable_qat()
    out_noq = g(iq)
    g.enable_qat()
    feats = torch.randn(48, 30) * 0.5
    out_pre = g(feats, pre_assembled=True)
    # stand-alone dynamic fake quant (per tensor / per row)
    x = torch.randn(5, 7) * 3
    fq_t = ref_dyn_fake_quantize(x, 16)
    fq_c = ref_dyn_fake_quantize(x, 16, per_channel=True, channel_dim=0)
    qw = g.get_quantized_weights()
    for k, v in qw.items():
        arrays["qw__" + k] = npy(v)
    save("g3_qat_gen", iq=npy(iq), iq2=npy(iq2), target=npy(target), out1=npy(out1), loss=npy(loss),
         out2=npy(out2), out_eval=npy(out_eval), out_row0=npy(out_row0), out_noq=npy(out_noq),
         feats=npy(feats), out_pre=npy(out_pre), fq_x=npy(x), fq_tensor=npy(fq_t), fq_channel=npy(fq_c),
         **arrays)


# ----------------------------------------------------------------------------- g4
def g4_fixed_point():
    torch.manual_seed(404)
    cfg = ref_quant.QuantizationConfig()
    # ties at +-0.5 LSB, saturation, signed zero
    lsb15, lsb8 = 2.0 ** -15, 2.0 ** -8
    edge15 = torch.tensor([0.5, -0.25, 0.999, -1.0, 0.0, 1.0, 1.5, -1.5, 0.5 * lsb15, 1.5 * lsb15, 2.5 * lsb15,
                           -0.5 * lsb15, -1.5 * lsb15, 32766.5 * lsb15, 32767.5 * lsb15, -32768.5 * lsb15,
                           0.3333333, -0.7777777], dtype=torch.float32)
    edge8 = torch.tensor([0.5 * lsb8, 1.5 * lsb8, 2.5 * lsb8, -0.5 * lsb8, -2.5 * lsb8, 127.996, 127.998, 128.0,
                          -128.0, -128.002, -200.0, 200.0, 3.14159, -2.71828, 0.0], dtype=torch.float32)
    rnd = torch.randn(257) * 2
    q15 = ref_quant.quantize_to_fixed_point(edge15, cfg.weight_scale, 16)
    q8 = ref_quant.quantize_to_fixed_point(edge8, cfg.activation_scale, 16)
    q8r = ref_quant.quantize_to_fixed_point(rnd, cfg.activation_scale, 16)
    qu8 = ref_quant.quantize_to_fixed_point(edge8, 16.0, 8, signed=False)
    fq15 = ref_quant.fake_quantize(edge15, cfg.weight_scale, 16)
    fq8r = ref_quant.fake_quantize(rnd, cfg.activation_scale, 16)
    dq8 = ref_quant.dequantize_from_fixed_point(q8, cfg.activation_scale)

    ql = ref_quant.QuantizedLinear(30, 32, cfg)
    with torch.no_grad():
        ql.bias.uniform_(-0.3, 0.3)
    x = (torch.randn(16, 30) * 0.7).requires_grad_(True)
    y = ql(x)
    gy = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, [x, ql.weight, ql.bias], gy)
    qlw = ql.get_quantized_weights()

    lut = ref_quant.generate_tanh_lut()
    lut16 = ref_quant.generate_tanh_lut(num_entries=16)
    lrp = ref_quant.generate_leaky_relu_params(0.2)

    wdict = {"fc1_weight": torch.arange(-6, 6, dtype=torch.int16).reshape(3, 4) * 1000,
             "fc1_bias": torch.tensor([-300, 0, 12345], dtype=torch.int16)}
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "w.bin")
        ref_quant.export_weights_binary(wdict, p)
        blob_le = np.frombuffer(open(p, "rb").read(), dtype=np.uint8).copy()
        ref_quant.export_weights_binary(wdict, p, endianness="big")
        blob_be = np.frombuffer(open(p, "rb").read(), dtype=np.uint8).copy()

    save("g4_fixed_point", edge15=npy(edge15), edge8=npy(edge8), rnd=npy(rnd), q15=npy(q15), q8=npy(q8),
         q8r=npy(q8r), qu8=npy(qu8), fq15=npy(fq15), fq8r=npy(fq8r), dq8=npy(dq8),
         ql_weight=npy(ql.weight), ql_bias=npy(ql.bias), ql_x=npy(x), ql_y=npy(y), ql_gy=npy(gy),
         ql_gx=npy(gx), ql_gw=npy(gw), ql_gb=npy(gb), ql_w_int=npy(qlw["weight"]), ql_b_int=npy(qlw["bias"]),
         tanh_lut=npy(lut), tanh_lut16=npy(lut16),
         lrelu_shift=np.array(lrp["shift_amount"]), lrelu_slope=np.array(lrp["actual_slope"]),
         lrelu_error=np.array(lrp["error"]),
         bin_fc1_weight=npy(wdict["fc1_weight"]), bin_fc1_bias=npy(wdict["fc1_bias"]),
         bin_blob_le=blob_le, bin_blob_be=blob_be)


# ----------------------------------------------------------------------------- g5
def g5_pa():
    torch.manual_seed(505)
    pa = PADigitalTwin()
    x = torch.randn(2, 64, 2) * 0.3
    arrays = {"base_re": npy(pa.base_coeffs_real), "base_im": npy(pa.base_coeffs_imag)}
    for st in ("cold", "normal", "hot"):
        xg = x.clone().requires_grad_(True)
        y = pa(xg, add_noise=False, temp_state=st)
        (dx,) = torch.autograd.grad((y ** 2).sum(), xg)
        arrays[f"y_{st}"] = npy(y)
        arrays[f"dx_{st}"] = npy(dx)
        arrays[f"coef_re_{st}"] = npy(pa.pa_model.coeffs_real)
        arrays[f"coef_im_{st}"] = npy(pa.pa_model.coeffs_imag)
    pa.set_temperature_state("normal")
    xz = x.clone()
    xz[0, 3] = 0.0  # |x| = 0 sample: gradient of abs at 0 is 0
    xg = xz.requires_grad_(True)
    y0 = pa(xg, add_noise=False)
    gy = torch.randn_like(y0)
    (dx0,) = torch.autograd.grad(y0, xg, gy)
    xc = torch.complex(x[..., 0], x[..., 1])
    yc = pa.pa_model(xc)  # complex input form
    inp, amp, ph = pa.get_transfer_function(num_points=8)
    save("g5_pa", x=npy(x), xz=npy(xz), y_zero=npy(y0), gy_zero=npy(gy), dx_zero=npy(dx0), y_complex_in=npy(yc),
         tf_in=npy(inp), tf_amp=npy(amp), tf_phase=npy(ph), **arrays)


# ----------------------------------------------------------------------------- g6
def g6_critic():
    torch.manual_seed(606)
    d = Discriminator()
    with torch.no_grad():
        for n, p in d.named_parameters():
            if n.endswith("bias"):
                p.uniform_(-0.1, 0.1)
    arrays = sd_np(d, "sd0__")
    x2 = torch.randn(8, 2)
    c2 = torch.randn(8, 2)
    x3 = torch.randn(4, 256, 2) * 0.5
    c3 = torch.randn(4, 256, 2) * 0.5
    d.eval()
    arrays["eval_out2"] = npy(d(x2, c2))
    arrays["eval_out3"] = npy(d(x3, c3))
    arrays.update(sd_np(d, "sd_eval__"))
    d.train()
    arrays["train_out2"] = npy(d(x2, c2))
    arrays.update(sd_np(d, "sd1__"))
    xg = x3.clone().requires_grad_(True)
    out3 = d(xg, c3)
    arrays.update(sd_np(d, "sd2__"))
    gs = torch.randn_like(out3)
    params = [p for _, p in d.named_parameters()]
    grads = torch.autograd.grad(out3, [xg] + params, gs)
    arrays["train_out3"] = npy(out3)
    arrays["train_gs3"] = npy(gs)
    arrays["train_dx3"] = npy(grads[0])
    for (n, _), gr in zip(d.named_parameters(), grads[1:]):
        arrays["grad3__" + n.replace(".", "__")] = npy(gr)
    # no spectral norm variant
    torch.manual_seed(607)
    dn = Discriminator(use_spectral_norm=False)
    arrays.update(sd_np(dn, "nosn__"))
    arrays["nosn_out3"] = npy(dn(x3, c3))
    save("g6_critic", x2=npy(x2), c2=npy(c2), x3=npy(x3), c3=npy(c3), **arrays)


# ----------------------------------------------------------------------------- g7
def g7_wgan_gp():
    torch.manual_seed(707)
    d = Discriminator()
    d.train()
    arrays = sd_np(d, "sd0__")
    B, L = 4, 64
    real = torch.randn(B, L, 2) * 0.4
    fake = torch.randn(B, L, 2) * 0.4
    cond = torch.randn(B, L, 2) * 0.4
    wl = WassersteinLoss(gp_weight=10.0)
    cpu = torch.device("cpu")

    # gradient penalty alone (eval mode: no power iteration) with replayed alpha
    d.eval()
    torch.manual_seed(7071)
    alpha = torch.rand(B, 1)
    torch.manual_seed(7071)
    gp = compute_gradient_penalty(d, real, fake, cond, cpu)
    params = [p for _, p in d.named_parameters()]
    gp_grads = torch.autograd.grad(gp, params, allow_unused=True)
    arrays["gp_eval"] = npy(gp)
    arrays["gp_alpha"] = npy(alpha)
    for (n, p), gr in zip(d.named_parameters(), gp_grads):
        arrays["gpgrad__" + n.replace(".", "__")] = npy(gr if gr is not None else torch.zeros_like(p))
    # 2-D input form [B,2]
    torch.manual_seed(7072)
    alpha2 = torch.rand(8, 1)
    r2, f2, c2 = torch.randn(8, 2), torch.randn(8, 2), torch.randn(8, 2)
    torch.manual_seed(7072)
    gp2 = compute_gradient_penalty(d, r2, f2, c2, cpu)
    arrays.update(gp2_alpha=npy(alpha2), gp2_real=npy(r2), gp2_fake=npy(f2), gp2_cond=npy(c2), gp2=npy(gp2))

    # full critic loss in train mode (3 power iterations), replayed alpha
    d.train()
    torch.manual_seed(7073)
    alpha_t = torch.rand(B, 1)
    torch.manual_seed(7073)
    total, info = wl.discriminator_loss(d, real, fake, cond, cpu)
    grads = torch.autograd.grad(total, params, allow_unused=True)
    arrays["dl_alpha"] = npy(alpha_t)
    arrays["dl_total"] = npy(total)
    for k, v in info.items():
        arrays["dl_info__" + k] = np.array(v, dtype=np.float64)
    for (n, p), gr in zip(d.named_parameters(), grads):
        arrays["dlgrad__" + n.replace(".", "__")] = npy(gr if gr is not None else torch.zeros_like(p))
    arrays.update(sd_np(d, "sd1__"))

    # generator loss (train mode, 1 power iteration), grad wrt fake
    fg = fake.clone().requires_grad_(True)
    gl, ginfo = wl.generator_loss(d, fg, cond)
    (dfake,) = torch.autograd.grad(gl, fg)
    arrays["gl"] = npy(gl)
    arrays["gl_dfake"] = npy(dfake)
    arrays.update(sd_np(d, "sd2__"))
    save("g7_wgan_gp", real=npy(real), fake=npy(fake), cond=npy(cond), **arrays)


# ----------------------------------------------------------------------------- g8
def g8_spectral():
    torch.manual_seed(808)
    arrays = {}
    sl = ref_spec.SpectralLoss()
    for L in (256, 1024):
        target = ofdm_rows(4, L + 5, seed=80 + L)[:, 5:, :].contiguous()
        pred = (target + 0.05 * target ** 3 + 0.01 * torch.randn_like(target)).requires_grad_(True)
        total, comps = sl(pred, target, return_components=True)
        (dpred,) = torch.autograd.grad(total, pred)
        met = sl.compute_metrics(pred.detach(), target)
        lo, up = ref_spec.compute_acpr(pred.detach())
        lo_lin, up_lin = ref_spec.compute_acpr(pred.detach(), return_db=False)
        arrays.update({
            f"L{L}_pred": npy(pred), f"L{L}_target": npy(target), f"L{L}_total": npy(total),
            f"L{L}_l1": npy(comps["l1"]), f"L{L}_evm": npy(comps["evm"]), f"L{L}_acpr": npy(comps["acpr"]),
            f"L{L}_dpred": npy(dpred),
            f"L{L}_evm_db": npy(ref_spec.compute_evm(pred.detach(), target)),
            f"L{L}_evm_lin": npy(ref_spec.compute_evm(pred.detach(), target, return_db=False)),
            f"L{L}_nmse_db": npy(ref_spec.compute_nmse(pred.detach(), target)),
            f"L{L}_nmse_lin": npy(ref_spec.compute_nmse(pred.detach(), target, return_db=False)),
            f"L{L}_acpr_lo_db": npy(lo), f"L{L}_acpr_up_db": npy(up),
            f"L{L}_acpr_lo_lin": npy(lo_lin), f"L{L}_acpr_up_lin": npy(up_lin),
        })
        for k, v in met.items():
            arrays[f"L{L}_metric__{k}"] = np.array(v, dtype=np.float64)
    # a case where the clamp / max branches are active: tiny error (evm term clamps to 0)
    t = arrays["L256_target"]
    tt = torch.from_numpy(t)
    pp = (tt * (1 + 1e-4)).requires_grad_(True)
    total, comps = sl(pp, tt, return_components=True)
    (dpp,) = torch.autograd.grad(total, pp)
    arrays.update(clamp_pred=npy(pp), clamp_total=npy(total), clamp_evm=npy(comps["evm"]),
                  clamp_acpr=npy(comps["acpr"]), clamp_dpred=npy(dpp))
    # standalone EVM / ACPR losses
    p256 = torch.from_numpy(arrays["L256_pred"]).requires_grad_(True)
    el = ref_spec.EVMLoss()(p256, tt)
    al = ref_spec.ACPRLoss()(p256)
    arrays.update(evmloss=npy(el), acprloss=npy(al),
                  evmloss_d=npy(torch.autograd.grad(el, p256)[0]),
                  acprloss_d=npy(torch.autograd.grad(al, p256)[0]))
    # mask bin lists for several lengths (incl. a non power of two)
    for L in (256, 1024, 4096, 100):
        f = torch.fft.fftfreq(L, d=1 / 200e6)
        main = torch.abs(f) <= 50e6
        lower = (f >= -150e6) & (f <= -50e6)
        upper = (f >= 50e6) & (f <= 150e6)
        arrays[f"mask{L}"] = np.stack([npy(main), npy(lower), npy(upper)]).astype(np.uint8)
    save("g8_spectral", **arrays)


# ----------------------------------------------------------------------------- g9
def g9_csv_chain():
    rows = np.loadtxt(os.path.join(REF, "data", "test_input.csv"), delimiter=",", skiprows=1, max_rows=4096)
    u = rows[:, 0] + 1j * rows[:, 1]
    u = (u * (0.7 / np.max(np.abs(u)))).astype(np.complex64)  # train.py:80-83 normalisation
    iq = torch.from_numpy(np.stack([u.real, u.imag], axis=-1).astype(np.float32)).unsqueeze(0)
    torch.manual_seed(909)
    g = TDNNGeneratorQAT()
    g.eval()
    pa = PADigitalTwin()
    with torch.no_grad():
        dpd = g(iq)                      # [1, 4091, 2]
        y = pa(dpd, add_noise=False)     # [1, 4087, 2]
        ref_sig = iq[:, 9:, :] * (10 ** (30 / 20))
        met = ref_spec.SpectralLoss().compute_metrics(y, ref_sig)
    arrays = sd_np(g, "sd__")
    for k, v in met.items():
        arrays["metric__" + k] = np.array(v, dtype=np.float64)
    save("g9_csv_chain", csv_rows=rows.astype(np.float64), iq=npy(iq), dpd=npy(dpd), y=npy(y), **arrays)


# ----------------------------------------------------------------------------- g10
def g10_steps():
    """One Stage-1 (MSE) step and one Stage-2 (CWGAN-GP) step in sequence mode [B, L+5, 2] -> [B, L, 2]."""
    B, L = 8, 256
    iq = ofdm_rows(B, L + 5, seed=1001)
    target = (iq[:, 5:, :] * 0.9 + 0.02 * torch.randn(B, L, 2, generator=torch.Generator().manual_seed(5))).contiguous()
    cond = iq[:, 5:, :].contiguous()
    cpu = torch.device("cpu")
    arrays = dict(iq=npy(iq), target=npy(target))

    def adam(params):
        return torch.optim.Adam(params, lr=1e-4, betas=(0.0, 0.9), weight_decay=1e-5)  # train.py:206-218

    for tag, qat in (("f", False), ("q", True)):
        torch.manual_seed(1010)
        g = TDNNGeneratorQAT()
        g.qat_enabled = qat
        g.train()
        arrays.update(sd_np(g, f"s1{tag}_g0__"))
        opt = adam(g.parameters())
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(g(iq), target)
        loss.backward()
        opt.step()
        arrays[f"s1{tag}_loss"] = npy(loss)
        arrays.update(sd_np(g, f"s1{tag}_g1__"))

    # Stage 2 (QAT on), n_critic = 5, fixed alphas
    torch.manual_seed(1020)
    g = TDNNGeneratorQAT()
    d = Discriminator()
    g.train(); d.train()
    arrays.update(sd_np(g, "s2_g0__")); arrays.update(sd_np(d, "s2_d0__"))
    g_opt, d_opt = adam(g.parameters()), adam(d.parameters())
    wl = WassersteinLoss(10.0)
    sl = ref_spec.SpectralLoss()
    alphas = []
    d_totals = []
    for it in range(5):
        d_opt.zero_grad()
        with torch.no_grad():
            fake = g(iq)
        torch.manual_seed(2000 + it)
        alphas.append(npy(torch.rand(B, 1)))
        torch.manual_seed(2000 + it)
        d_loss, d_info = wl.discriminator_loss(d, target, fake, cond, cpu)
        d_loss.backward()
        d_opt.step()
        d_totals.append(float(d_loss))
    arrays["s2_alphas"] = np.stack(alphas)
    arrays["s2_d_totals"] = np.array(d_totals)
    arrays.update(sd_np(d, "s2_d1__"))
    g_opt.zero_grad()
    fake = g(iq)
    g_adv, _ = wl.generator_loss(d, fake, cond)
    recon = torch.nn.functional.l1_loss(fake, target)
    spectral, comps = sl(fake, target, return_components=True)
    g_total = 1.0 * g_adv + 50.0 * recon + spectral  # train.py:336-340
    g_total.backward()
    g_opt.step()
    arrays.update(s2_g_total=npy(g_total), s2_g_adv=npy(g_adv), s2_recon=npy(recon), s2_spectral=npy(spectral),
                  s2_evm=npy(comps["evm"]), s2_acpr=npy(comps["acpr"]))
    arrays.update(sd_np(g, "s2_g1__")); arrays.update(sd_np(d, "s2_d2__"))
    save("g10_steps", **arrays)


# ----------------------------------------------------------------------------- g11
def g11_q88():
    """Fixed-point inference composed from the reference's own primitives in float64.

    Every product and sum below is exact in float64 (|acc| < 2^40), so the result equals
    pure-integer arithmetic.  Order: int16 Q1.15 IQ -> MemoryTapAssembly (float64) ->
    quantize_to_fixed_point(.,256,16) -> per layer [W int16 Q1.15, bias int16 Q8.8,
    acc Q9.23, requantise to Q8.8] -> LeakyReLU(0.2) -> requantise -> FC3 -> tanh LUT.
    The LUT index rule (nearest entry of linspace(-4,4,256), saturating) is this build's choice;
    the reference has no end-to-end integer function.
    """
    torch.manual_seed(1111)
    qtf = ref_quant.quantize_to_fixed_point
    g = TDNNGeneratorQAT()
    with torch.no_grad():
        for lin in (g.fc1, g.fc2, g.fc3):
            lin.weight.mul_(1.6)
            lin.bias.uniform_(-0.2, 0.2)
    W = [qtf(l.weight.detach(), 32768.0, 16).to(torch.float64) for l in (g.fc1, g.fc2, g.fc3)]
    Bq = [qtf(l.bias.detach(), 256.0, 16).to(torch.float64) for l in (g.fc1, g.fc2, g.fc3)]
    lut = ref_quant.generate_tanh_lut().to(torch.float64)

    iqf = ofdm_rows(1, 4101, seed=1112, peak=0.95)
    iq_i = qtf(iqf, 32768.0, 16)  # int16 Q1.15 values held in fp32 (exact)
    iq_i[0, 100] = torch.tensor([32767.0, -32768.0])   # full-scale corner
    iq_i[0, 200] = torch.tensor([0.0, 0.0])
    x = iq_i.to(torch.float64) / 32768.0
    feat = MemoryTapAssembly(5)(x)                      # float64 features
    f_q = qtf(feat.reshape(-1, 30), 256.0, 16)          # Q8.8 integers (float64)

    def layer(a_q, Wq, bq):
        acc = a_q @ Wq.t() + bq * 32768.0               # Q9.23, exact
        return qtf(acc / 32768.0 / 256.0, 256.0, 16)    # -> Q8.8 (round-half-even, saturate)

    def lrelu_q(z_q):
        v = torch.where(z_q > 0, z_q, z_q * 0.2)        # LeakyReLU(0.2) on the Q8.8 value
        return qtf(v / 256.0, 256.0, 16)

    z1 = layer(f_q, W[0], Bq[0]); h1 = lrelu_q(z1)
    z2 = layer(h1, W[1], Bq[1]); h2 = lrelu_q(z2)
    z3 = layer(h2, W[2], Bq[2])
    idx = torch.clamp(torch.round((z3 / 256.0 + 4.0) * (255.0 / 8.0)), 0, 255).to(torch.int64)
    out = lut[idx]
    i16 = lambda t: npy(t).astype(np.int16)
    save("g11_q88", iq_i16=i16(iq_i), w1=i16(W[0]), w2=i16(W[1]), w3=i16(W[2]), b1=i16(Bq[0]), b2=i16(Bq[1]),
         b3=i16(Bq[2]), lut=i16(lut), feat_q=i16(f_q), z1=i16(z1), h1=i16(h1), z2=i16(z2), h2=i16(h2), z3=i16(z3),
         lut_idx=npy(idx).astype(np.int16), out_i16=i16(out))


# ----------------------------------------------------------------------------- g12
def _ref_functions(rel, names, extra_globals):
    """Function (or method) definitions taken out of a reference file whose module-level imports cannot be
    satisfied here (train.py wants tensorboard, utils/dataset.py wants h5py): the named `def`s are compiled
    from the file's own syntax tree and run as they stand; nothing else of the file is executed."""
    import ast
    with open(os.path.join(REF, rel)) as f:
        tree = ast.parse(f.read())
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = dict(extra_globals)
            exec(compile(mod, os.path.join(REF, rel), "exec"), ns)
            found[node.name] = ns[node.name]
    assert set(found) == set(names), (names, list(found))
    return found


def g12_data():
    """Data path: CSV loading + normalisation (train.py:55-89), thermal drift (train.py:92-126) and the 50 %-overlap
    windows of DPDDataset._prepare_sequences (utils/dataset.py:230-287) on a 1,500-row slice of the test split."""
    import contextlib
    import types
    from pathlib import Path
    import pandas as pd
    n = 1500
    tmp = Path(tempfile.mkdtemp())
    for kind in ("input", "output"):
        pd.read_csv(os.path.join(REF, "data", f"test_{kind}.csv"), nrows=n).to_csv(tmp / f"test_{kind}.csv", index=False)
    from torch.utils.data import TensorDataset
    fns = _ref_functions("train.py", ["load_measured_data", "apply_thermal_drift"], dict(np=np, pd=pd, Path=Path, torch=torch, TensorDataset=TensorDataset))
    with contextlib.redirect_stdout(io.StringIO()):
        u, y = fns["load_measured_data"](tmp, "test")
    arrays = dict(csv_input=pd.read_csv(tmp / "test_input.csv").values.astype(np.float64),
                  csv_output=pd.read_csv(tmp / "test_output.csv").values.astype(np.float64),
                  u=np.stack([u.real, u.imag], -1).astype(np.float32), y=np.stack([y.real, y.imag], -1).astype(np.float32))
    for t in (-20.0, 25.0, 70.0):
        d = fns["apply_thermal_drift"](y, t)
        arrays[f"drift_{int(t)}"] = np.stack([d.real, d.imag], -1).astype(np.float32)
    meth = _ref_functions("utils/dataset.py", ["_prepare_sequences", "_apply_thermal_drift"], dict(np=np))
    for L in (256, 100):
        ds = types.SimpleNamespace(memory_depth=5, sequence_length=L, temp_state="normal", normalize=False,
                                   pa_input_raw=u.copy(), pa_output_raw=y.copy(),
                                   thermal_params={"normal": {"gain_drift": 0.0, "phase_drift_deg": 0, "am_am_scale": 1.0}})
        ds._apply_thermal_drift = types.MethodType(meth["_apply_thermal_drift"], ds)
        with contextlib.redirect_stdout(io.StringIO()):
            meth["_prepare_sequences"](ds)
        arrays[f"win{L}_input"] = np.stack([a for a, _ in ds.sequences]).astype(np.float32)
        arrays[f"win{L}_target"] = np.stack([b for _, b in ds.sequences]).astype(np.float32)
    save("g12_data", **arrays)


# ----------------------------------------------------------------------------- g13
def g13_q88_hex():
    """Q8.8 hex export of the generator weights: the notebook's own `to_q8_8` and `export_to_hex` (training_colab.ipynb,
    weight-export cell), taken out of the cell's syntax tree and run on seeded weights that include saturating values
    and exact ties."""
    import ast
    import json
    with open(os.path.join(REF, "training_colab.ipynb")) as f:
        nb = json.load(f)
    src = next("".join(c["source"]) for c in nb["cells"] if c["cell_type"] == "code" and "def to_q8_8" in "".join(c["source"]))
    # the cell is notebook text with shell escapes elsewhere: parse only the two function definitions
    fns = {}
    for name in ("to_q8_8", "export_to_hex"):
        start = src.index(f"def {name}(")
        lines = src[start:].split("\n")
        body = [lines[0]]
        for ln in lines[1:]:
            if ln.strip() and not ln.startswith((" ", "\t")):
                break
            body.append(ln)
        ns = dict(np=np)
        exec(compile(ast.parse("\n".join(body)), "training_colab.ipynb", "exec"), ns)
        fns[name] = ns[name]
    torch.manual_seed(1313)
    g = TDNNGeneratorQAT()
    with torch.no_grad():
        g.fc1.weight[0, :4] = torch.tensor([130.0, -129.5, 127.998, -128.0])         # saturation
        g.fc1.weight[1, :4] = torch.tensor([0.5, 1.5, 2.5, -0.5]) / 256               # ties
        g.fc2.bias.copy_(torch.linspace(-0.3, 0.3, 16))
        g.fc3.bias.copy_(torch.tensor([0.001953125, -0.005859375]))                     # 0.5 / 256, -1.5 / 256
    arrays = sd_np(g, "sd__")
    tmp = tempfile.mkdtemp()
    for i, fc in ((1, g.fc1), (2, g.fc2), (3, g.fc3)):
        for kind, t in (("weights", fc.weight), ("bias", fc.bias)):
            q = fns["to_q8_8"](t.detach().numpy())
            path = os.path.join(tmp, f"fc{i}_{kind}.hex")
            # numpy 2 refuses `int16 & 0xFFFF` (the notebook ran under numpy 1, which widened): hand the same values over as int64
            fns["export_to_hex"](q.astype(np.int64), path)
            arrays[f"fc{i}_{kind}_q"] = q
            arrays[f"fc{i}_{kind}_hex"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    save("g13_q88_hex", **arrays)


# ----------------------------------------------------------------------------- g14
def g14_augment():
    """Augmentation and PA noise with recorded draws: the notebook's own `augment_signal` (training_colab.ipynb,
    data-loading cell), taken out of the cell's text, and PADigitalTwin.add_noise, each run under a seed; the random
    numbers they consumed are recorded by replaying the same sequence of generator calls under the same seed."""
    import ast
    import json
    with open(os.path.join(REF, "training_colab.ipynb")) as f:
        nb = json.load(f)
    src = next("".join(c["source"]) for c in nb["cells"] if c["cell_type"] == "code" and "def augment_signal" in "".join(c["source"]))
    start = src.index("def augment_signal(")
    lines = src[start:].split("\n")
    body = [lines[0]]
    for ln in lines[1:]:
        if ln.strip() and not ln.startswith((" ", "\t")):
            break
        body.append(ln)
    ns = dict(torch=torch, np=np)
    exec(compile(ast.parse("\n".join(body)), "training_colab.ipynb", "exec"), ns)
    augment_signal = ns["augment_signal"]
    sig = ofdm_rows(1, 2048 + 5, seed=77)[0].contiguous()            # [2053, 2]
    arrays = dict(signal=npy(sig))
    cases = []
    # seeds picked so that the thermal draw of the 'thermal' / 'all' cases covers cold, unchanged and hot
    for kind in ("noise", "phase", "gain"):
        cases.append((kind, 0.1, 5))
    seed = 0
    want = {"cold": None, "normal": None, "hot": None}
    while any(v is None for v in want.values()):
        torch.manual_seed(seed)
        t = torch.rand(1).item()
        st = "cold" if t < 0.33 else ("normal" if t < 0.67 else "hot")
        if want[st] is None:
            want[st] = seed
        seed += 1
    for st, sd in want.items():
        cases.append(("thermal", 0.1, sd))
    want = {"cold": None, "normal": None, "hot": None}
    seed = 100
    while any(v is None for v in want.values()):
        torch.manual_seed(seed)
        torch.randn(1); torch.randn_like(sig); torch.rand(1); torch.rand(1)
        t = torch.rand(1).item()
        st = "cold" if t < 0.33 else ("normal" if t < 0.67 else "hot")
        if want[st] is None:
            want[st] = seed
        seed += 1
    for st, sd in want.items():
        cases.append(("all", 0.3 if st == "hot" else 0.1, sd))
    for i, (kind, strength, sd) in enumerate(cases):
        torch.manual_seed(sd)
        out = augment_signal(sig, augment_type=kind, strength=strength)
        torch.manual_seed(sd)                                           # replay: the draws in the order the function takes them
        d = dict(snr=0.0, phase=0.5, gain=0.5, thermal=0.5, noise=torch.zeros_like(sig))
        if kind in ("noise", "all"):
            d["snr"] = torch.randn(1).item()
            d["noise"] = torch.randn_like(sig)
        if kind in ("phase", "all"):
            d["phase"] = torch.rand(1).item()
        if kind in ("gain", "all"):
            d["gain"] = torch.rand(1).item()
        if kind in ("thermal", "all"):
            d["thermal"] = torch.rand(1).item()
        arrays[f"c{i}_kind"] = np.array(kind)
        arrays[f"c{i}_strength"] = np.array(strength)
        arrays[f"c{i}_scalars"] = np.array([d["snr"], d["phase"], d["gain"], d["thermal"]], dtype=np.float64)
        arrays[f"c{i}_noise"] = npy(d["noise"])
        arrays[f"c{i}_out"] = npy(out)
    arrays["n_cases"] = np.array(len(cases))
    # PADigitalTwin.add_noise: awgn = randn_like(x), then phase noise = randn(B, S, 1)
    pa = PADigitalTwin()
    x = ofdm_rows(3, 261, seed=78) * 20.0
    torch.manual_seed(4242)
    y = pa.add_noise(x)
    torch.manual_seed(4242)
    awgn = torch.randn_like(x)
    pn = torch.randn(x.shape[0], x.shape[1], 1)
    arrays.update(pa_x=npy(x), pa_awgn=npy(awgn), pa_phase=npy(pn.squeeze(-1)), pa_out=npy(y),
                  pa_noise_floor_db=np.array(pa.noise_floor_db), pa_phase_noise_deg=np.array(pa.phase_noise_deg))
    save("g14_augment", **arrays)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    for fn in (g1_taps, g2_float_gen, g3_qat_gen, g4_fixed_point, g5_pa, g6_critic, g7_wgan_gp, g8_spectral,
               g9_csv_chain, g10_steps, g11_q88, g12_data, g13_q88_hex, g14_augment):
        if len(sys.argv) > 1 and fn.__name__ not in sys.argv[1:]:
            continue
        fn()
