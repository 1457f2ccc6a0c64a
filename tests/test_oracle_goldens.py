"""The CPU oracle against fixtures captured from the reference modules (tools/make_goldens.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import critic as oc, fixed_point as ofp, generator as og, pa as opa, q88 as oq, spectral as osp, steps as ost, taps as ot

T = torch.from_numpy


def sd(g, prefix):
    return {k[len(prefix):].replace("__", "."): T(g[k]) for k in g.files if k.startswith(prefix)}


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def close_q(a, b, frac=0.99):
    """Fake-quantised outputs: equal up to one 16-bit step of the tensor's own scale (a last-bit
    difference in a GEMM can move a value across a rounding boundary), and mostly bit-identical."""
    a, b = np.asarray(a), np.asarray(b)
    lsb = np.abs(b).max() / 32767
    assert np.abs(a - b).max() <= 1.001 * lsb
    assert (a == b).mean() >= frac


# ------------------------------------------------------------------ tap assembly
def test_taps_index_map_bit_exact():
    g = load_golden("g1_taps")
    # exactly representable probe: every feature must be bit-identical
    assert np.array_equal(ot.tap_assembly(T(g["probe"])).numpy(), g["feat_probe"])
    f = ot.tap_assembly(T(g["iq"])).numpy()
    e4 = slice(14, 20)
    mask = np.ones(30, bool); mask[e4] = False
    assert np.array_equal(f[..., mask], g["feat"][..., mask])
    close(f[..., e4], g["feat"][..., e4], rtol=3e-7, atol=0)       # pow(x, 4): last-ulp freedom
    fa = ot.tap_assembly(T(g["arange_iq"])).numpy()
    close(fa, g["feat_arange"], rtol=3e-7, atol=0)


def test_taps_short_sequence_raises():
    with pytest.raises(ValueError):
        ot.tap_assembly(torch.zeros(1, 5, 2))


# ------------------------------------------------------------------ generator
def test_float_generator():
    g = load_golden("g2_float_gen")
    p = sd(g, "sd__")
    close(og.gen_forward(T(g["iq"]), p), g["out"])
    close(og.gen_forward(T(g["feats"]), p, pre_assembled=True), g["out_pre"])
    out, loss, grads = og.gen_mse_grads(T(g["iq"]), T(g["target"]), p)
    close(loss, g["loss"])
    for k in og.PARAM_KEYS:
        close(grads[k], g["grad__" + k.replace(".", "__")], rtol=1e-4, atol=1e-7)


def test_qat_generator_and_buffers():
    g = load_golden("g3_qat_gen")
    p = {k: v for k, v in sd(g, "sd0__").items() if k in og.PARAM_KEYS}
    running = ost.new_running()
    out1 = og.gen_forward(T(g["iq"]), p, qat=True, running=running)
    close_q(out1.numpy(), g["out1"])
    for k, v in running.items():
        close(v, g[f"sd1__{k}__running_scale"], rtol=1e-6, atol=0)
    out2 = og.gen_forward(T(g["iq2"]), p, qat=True, running=running)
    close_q(out2.numpy(), g["out2"])
    for k, v in running.items():
        close(v, g[f"sd2__{k}__running_scale"], rtol=1e-6, atol=0)
    close_q(og.gen_forward(T(g["iq"]), p, qat=True).numpy(), g["out_eval"])
    close_q(og.gen_forward(T(g["iq"][:1]), p, qat=True).numpy(), g["out_row0"])
    close(og.gen_forward(T(g["iq"]), p, qat=False).numpy(), g["out_noq"], rtol=1e-6, atol=1e-8)
    close_q(og.gen_forward(T(g["feats"]), p, qat=True, pre_assembled=True).numpy(), g["out_pre"])
    _, loss, grads = og.gen_mse_grads(T(g["iq"]), T(g["target"]), p, qat=True)
    close(loss, g["loss"])
    for k in og.PARAM_KEYS:
        close(grads[k], g["grad__" + k.replace(".", "__")], rtol=1e-5, atol=1e-8)
    assert np.array_equal(og.fake_quant_dyn(T(g["fq_x"])).numpy(), g["fq_tensor"])
    assert np.array_equal(og.fake_quant_dyn(T(g["fq_x"]), per_row=True).numpy(), g["fq_channel"])
    qw = og.export_int16_weights(p)
    for k, v in qw.items():
        assert np.array_equal(v.numpy(), g["qw__" + k])


# ------------------------------------------------------------------ fixed point
def test_fixed_point_primitives():
    g = load_golden("g4_fixed_point")
    assert np.array_equal(ofp.quantize_fixed(T(g["edge15"]), 32768.0, 16).numpy(), g["q15"])
    assert np.array_equal(ofp.quantize_fixed(T(g["edge8"]), 256.0, 16).numpy(), g["q8"])
    assert np.array_equal(ofp.quantize_fixed(T(g["rnd"]), 256.0, 16).numpy(), g["q8r"])
    assert np.array_equal(ofp.quantize_fixed(T(g["edge8"]), 16.0, 8, signed=False).numpy(), g["qu8"])
    assert np.array_equal(ofp.fake_quantize_fixed(T(g["edge15"]), 32768.0, 16).numpy(), g["fq15"])
    assert np.array_equal(ofp.fake_quantize_fixed(T(g["rnd"]), 256.0, 16).numpy(), g["fq8r"])
    assert np.array_equal(ofp.dequantize_fixed(T(g["q8"]), 256.0).numpy(), g["dq8"])
    x = T(g["ql_x"]).requires_grad_(True)
    w = T(g["ql_weight"]).requires_grad_(True)
    b = T(g["ql_bias"]).requires_grad_(True)
    y = ofp.quantized_linear(x, w, b)
    assert np.array_equal(y.detach().numpy(), g["ql_y"])
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], T(g["ql_gy"]))
    close(gx, g["ql_gx"]); close(gw, g["ql_gw"]); close(gb, g["ql_gb"])
    assert np.array_equal(ofp.quantize_fixed(w.detach(), 32768.0, 16).to(torch.int16).numpy(), g["ql_w_int"])
    assert np.array_equal(ofp.quantize_fixed(b.detach(), 256.0, 16).to(torch.int16).numpy(), g["ql_b_int"])
    assert np.array_equal(ofp.tanh_lut().numpy(), g["tanh_lut"])
    assert np.array_equal(ofp.tanh_lut(16).numpy(), g["tanh_lut16"])
    assert ofp.tanh_lut()[0] == -32746 and ofp.tanh_lut()[-1] == 32746
    lp = ofp.leaky_relu_params(0.2)
    assert lp["shift_amount"] == int(g["lrelu_shift"]) and lp["actual_slope"] == float(g["lrelu_slope"])
    wd = {"fc1_weight": g["bin_fc1_weight"], "fc1_bias": g["bin_fc1_bias"]}
    assert ofp.pack_weights(wd) == g["bin_blob_le"].tobytes()
    assert ofp.pack_weights(wd, "big") == g["bin_blob_be"].tobytes()


# ------------------------------------------------------------------ PA twin
def test_pa_twin():
    g = load_golden("g5_pa")
    re0, im0 = opa.init_coeffs()
    assert np.array_equal(re0.numpy(), g["base_re"]) and np.array_equal(im0.numpy(), g["base_im"])
    x = T(g["x"])
    for st in ("cold", "normal", "hot"):
        re, im = opa.thermal_coeffs(re0, im0, st)
        assert np.array_equal(re.numpy(), g[f"coef_re_{st}"]) and np.array_equal(im.numpy(), g[f"coef_im_{st}"])
        xg = x.clone().requires_grad_(True)
        y = opa.volterra_forward(xg, re, im)
        close(y.detach(), g[f"y_{st}"], rtol=1e-5, atol=1e-5)
        (dx,) = torch.autograd.grad((y ** 2).sum(), xg)
        close(dx, g[f"dx_{st}"], rtol=1e-4, atol=1e-2)
    xz = T(g["xz"]).requires_grad_(True)
    ren, imn = opa.thermal_coeffs(re0, im0, "normal")   # not the initial set: the linear row loses its imaginary part
    y0 = opa.volterra_forward(xz, ren, imn)
    close(y0.detach(), g["y_zero"], rtol=1e-5, atol=1e-5)
    (dx0,) = torch.autograd.grad(y0, xz, T(g["gy_zero"]))
    close(dx0, g["dx_zero"], rtol=1e-4, atol=1e-3)
    xc = torch.complex(x[..., 0], x[..., 1])
    close(opa.volterra_forward(xc, ren, imn), g["y_complex_in"], rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        opa.volterra_forward(torch.zeros(1, 4, 2), re0, im0)
    with pytest.raises(ValueError):
        opa.thermal_coeffs(re0, im0, "warm")


# ------------------------------------------------------------------ critic
def test_critic_forward_and_power_iteration():
    g = load_golden("g6_critic")
    layers = oc.layers_from_state_dict(sd(g, "sd0__"))
    x2, c2, x3, c3 = (T(g[k]) for k in ("x2", "c2", "x3", "c3"))
    close(oc.critic_forward(layers, x2, c2, training=False), g["eval_out2"])
    close(oc.critic_forward(layers, x3, c3, training=False), g["eval_out3"])
    close(oc.critic_forward(layers, x2, c2, training=True), g["train_out2"])
    for pfx, l in zip(oc.LAYER_PREFIXES, layers):
        close(l["u"], g["sd1__" + pfx.replace(".", "__") + "__weight_u"], rtol=1e-5, atol=1e-7)
        close(l["v"], g["sd1__" + pfx.replace(".", "__") + "__weight_v"], rtol=1e-5, atol=1e-7)
    for l in layers:
        l["weight_orig"].requires_grad_(True); l["bias"].requires_grad_(True)
    xg = x3.clone().requires_grad_(True)
    out3 = oc.critic_forward(layers, xg, c3, training=True)
    close(out3.detach(), g["train_out3"])
    ins = [xg] + [t for l in layers for t in (l["bias"], l["weight_orig"])]
    grads = torch.autograd.grad(out3, ins, T(g["train_gs3"]))
    close(grads[0], g["train_dx3"], rtol=1e-4, atol=1e-6)
    i = 1
    for pfx in oc.LAYER_PREFIXES:
        close(grads[i], g["grad3__" + pfx.replace(".", "__") + "__bias"], rtol=1e-4, atol=1e-4); i += 1
        close(grads[i], g["grad3__" + pfx.replace(".", "__") + "__weight_orig"], rtol=1e-4, atol=1e-4); i += 1
    nosn = sd(g, "nosn__")
    ln = [dict(weight_orig=nosn[p + ".weight"], bias=nosn[p + ".bias"]) for p in
          ("features.0", "features.2", "features.4", "output")]
    close(oc.critic_forward(ln, x3, c3, training=True, spectral_norm=False), g["nosn_out3"])


def test_wgan_gp_losses():
    g = load_golden("g7_wgan_gp")
    layers = oc.layers_from_state_dict(sd(g, "sd0__"))
    for l in layers:
        l["weight_orig"].requires_grad_(True); l["bias"].requires_grad_(True)
    real, fake, cond = (T(g[k]) for k in ("real", "fake", "cond"))
    gp = oc.gradient_penalty(layers, real, fake, cond, T(g["gp_alpha"]), training=False)
    close(gp.detach(), g["gp_eval"], rtol=1e-5)
    ws = [l["weight_orig"] for l in layers]
    for pfx, gr in zip(oc.LAYER_PREFIXES, torch.autograd.grad(gp, ws)):
        close(gr, g["gpgrad__" + pfx.replace(".", "__") + "__weight_orig"], rtol=1e-4, atol=1e-6)
    gp2 = oc.gradient_penalty(layers, T(g["gp2_real"]), T(g["gp2_fake"]), T(g["gp2_cond"]), T(g["gp2_alpha"]), training=False)
    close(gp2.detach(), g["gp2"], rtol=1e-5)
    total, info = oc.discriminator_loss(layers, real, fake, cond, T(g["dl_alpha"]), training=True)
    close(total.detach(), g["dl_total"], rtol=1e-5)
    for k in ("d_loss", "gp", "w_distance", "real_score", "fake_score"):
        close(info[k].detach(), g["dl_info__" + k], rtol=1e-4, atol=1e-6)
    ins = [t for l in layers for t in (l["bias"], l["weight_orig"])]
    grads = torch.autograd.grad(total, ins, allow_unused=True)
    i = 0
    for pfx in oc.LAYER_PREFIXES:
        for nm in ("bias", "weight_orig"):
            gr = grads[i] if grads[i] is not None else torch.zeros_like(ins[i])
            close(gr, g["dlgrad__" + pfx.replace(".", "__") + "__" + nm], rtol=2e-4, atol=2e-5); i += 1
    for pfx, l in zip(oc.LAYER_PREFIXES, layers):
        close(l["u"], g["sd1__" + pfx.replace(".", "__") + "__weight_u"], rtol=1e-5, atol=1e-7)
    fg = fake.clone().requires_grad_(True)
    gl = oc.generator_loss(layers, fg, cond, training=True)
    close(gl.detach(), g["gl"], rtol=1e-5)
    close(torch.autograd.grad(gl, fg)[0], g["gl_dfake"], rtol=1e-4, atol=1e-9)


# ------------------------------------------------------------------ spectral
def test_spectral_metrics_and_loss():
    g = load_golden("g8_spectral")
    for L in (256, 1024, 4096, 100):
        m = np.stack([t.numpy() for t in osp.band_masks(L)]).astype(np.uint8)
        assert np.array_equal(m, g[f"mask{L}"])
        if L & (L - 1) == 0:   # closed form for powers of two
            main = np.zeros(L, bool); main[:L // 4 + 1] = True; main[3 * L // 4:] = True
            up = np.zeros(L, bool); up[L // 4:L // 2] = True
            lo = np.zeros(L, bool); lo[L // 2:3 * L // 4 + 1] = True
            assert np.array_equal(m.astype(bool), np.stack([main, lo, up]))
    for L in (256, 1024):
        pred = T(g[f"L{L}_pred"]).requires_grad_(True)
        tgt = T(g[f"L{L}_target"])
        total, comps = osp.spectral_loss(pred, tgt)
        close(total.detach(), g[f"L{L}_total"], rtol=1e-5)
        for k in ("l1", "evm", "acpr"):
            close(comps[k].detach(), g[f"L{L}_{k}"], rtol=1e-5)
        close(torch.autograd.grad(total, pred)[0], g[f"L{L}_dpred"], rtol=1e-3, atol=1e-6)
        close(osp.evm(pred.detach(), tgt), g[f"L{L}_evm_db"], rtol=1e-5)
        close(osp.evm(pred.detach(), tgt, db=False), g[f"L{L}_evm_lin"], rtol=1e-5)
        close(osp.nmse(pred.detach(), tgt), g[f"L{L}_nmse_db"], rtol=1e-5)
        lo, up = osp.acpr(pred.detach())
        close(lo, g[f"L{L}_acpr_lo_db"], rtol=1e-5); close(up, g[f"L{L}_acpr_up_db"], rtol=1e-5)
        m = osp.metrics(pred.detach(), tgt)
        for k, v in m.items():
            close(v, g[f"L{L}_metric__{k}"], rtol=1e-5)
    pp = T(g["clamp_pred"]).requires_grad_(True)
    total, comps = osp.spectral_loss(pp, T(g["L256_target"]))
    close(total.detach(), g["clamp_total"], rtol=1e-5)
    close(torch.autograd.grad(total, pp)[0], g["clamp_dpred"], rtol=1e-3, atol=1e-6)


# ------------------------------------------------------------------ chain on measured data
def test_csv_chain():
    g = load_golden("g9_csv_chain")
    rows = g["csv_rows"]
    u = rows[:, 0] + 1j * rows[:, 1]
    u = (u * (0.7 / np.max(np.abs(u)))).astype(np.complex64)
    iq = T(np.stack([u.real, u.imag], -1).astype(np.float32)).unsqueeze(0)
    assert np.array_equal(iq.numpy(), g["iq"])
    p = {k: v for k, v in sd(g, "sd__").items() if k in og.PARAM_KEYS}
    dpd = og.gen_forward(iq, p, qat=True)
    close_q(dpd.numpy(), g["dpd"])
    y = opa.volterra_forward(dpd, *opa.init_coeffs())
    close(y, g["y"], rtol=1e-5, atol=1e-5)
    m = osp.metrics(y, iq[:, 9:, :] * (10 ** (30 / 20)))
    for k, v in m.items():
        assert abs(v - float(g["metric__" + k])) < 0.1 if k.endswith("_db") else abs(v - float(g["metric__" + k])) < 1e-4


# ------------------------------------------------------------------ training steps
def test_stage1_and_stage2_steps():
    g = load_golden("g10_steps")
    iq, target = T(g["iq"]), T(g["target"])
    cond = iq[:, 5:, :].contiguous()
    for tag, qat in (("f", False), ("q", True)):
        p0 = {k: v for k, v in sd(g, f"s1{tag}_g0__").items() if k in og.PARAM_KEYS}
        gs = ost.GenState(p0)
        loss = ost.stage1_step(gs, iq, target, qat)
        close(loss, g[f"s1{tag}_loss"], rtol=1e-5)
        p1 = sd(g, f"s1{tag}_g1__")
        for k in og.PARAM_KEYS:
            close(gs.p[k].detach(), p1[k], rtol=1e-5, atol=2e-7)
    gs = ost.GenState({k: v for k, v in sd(g, "s2_g0__").items() if k in og.PARAM_KEYS})
    ds = ost.CriticState(oc.layers_from_state_dict(sd(g, "s2_d0__")))
    info = ost.stage2_step(gs, ds, iq, target, cond, [T(a) for a in g["s2_alphas"]])
    for it in range(5):
        close(info[f"d_total_{it}"], g["s2_d_totals"][it], rtol=1e-4, atol=1e-5)
    close(info["g_total"], g["s2_g_total"], rtol=1e-5)
    g1, d2 = sd(g, "s2_g1__"), sd(g, "s2_d2__")
    for k in og.PARAM_KEYS:
        close(gs.p[k].detach(), g1[k], rtol=1e-5, atol=3e-7)
    for pfx, l in zip(oc.LAYER_PREFIXES, ds.layers):
        close(l["weight_orig"].detach(), d2[pfx + ".weight_orig"], rtol=1e-5, atol=3e-7)
        close(l["bias"].detach(), d2[pfx + ".bias"], rtol=1e-5, atol=3e-7)
        close(l["u"], d2[pfx + ".weight_u"], rtol=1e-4, atol=1e-6)
    for k in og.ACT_QUANT_KEYS + og.WEIGHT_QUANT_KEYS:
        close(gs.running[k], g1[k + ".running_scale"], rtol=1e-5)


# ------------------------------------------------------------------ Q8.8 integer inference
def test_q88_bit_exact():
    g = load_golden("g11_q88")
    out, mid = oq.infer(g["iq_i16"][0], g["w1"], g["b1"], g["w2"], g["b2"], g["w3"], g["b3"], g["lut"],
                        return_intermediates=True)
    for k in ("feat_q", "z1", "h1", "z2", "h2", "z3", "lut_idx"):
        assert np.array_equal(mid[k].astype(np.int64), g[k].astype(np.int64)), k
    assert np.array_equal(out, g["out_i16"])


def test_augmentation_and_pa_noise_with_recorded_draws():
    """oracle/augment.py against what the notebook's own augment_signal and PADigitalTwin.add_noise produced from the
    same random numbers (fixture g14; both are the same torch ops in the same order: equal to the last bits)."""
    from oracle import augment as oa
    g = load_golden("g14_augment")
    sig = torch.from_numpy(g["signal"])
    kinds = set()
    for i in range(int(g["n_cases"])):
        kind, strength = str(g[f"c{i}_kind"]), float(g[f"c{i}_strength"])
        snr, phase, gain, thermal = (float(v) for v in g[f"c{i}_scalars"])
        out = oa.augment_signal(sig, kind, strength, snr, torch.from_numpy(g[f"c{i}_noise"]), phase, gain, thermal)
        np.testing.assert_allclose(out.numpy(), g[f"c{i}_out"], rtol=1e-6, atol=1e-7)
        kinds.add((kind, "cold" if thermal < 0.33 else ("normal" if thermal < 0.67 else "hot")))
    assert {k for k, _ in kinds} == {"noise", "phase", "gain", "thermal", "all"}
    assert {t for k, t in kinds if k == "all"} == {"cold", "normal", "hot"}
    y = oa.pa_add_noise(torch.from_numpy(g["pa_x"]), torch.from_numpy(g["pa_awgn"]), torch.from_numpy(g["pa_phase"]),
                        float(g["pa_noise_floor_db"]), float(g["pa_phase_noise_deg"]))
    np.testing.assert_allclose(y.numpy(), g["pa_out"], rtol=1e-6, atol=1e-6)
