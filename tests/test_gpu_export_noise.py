"""Rows next to the hot path on the GPU: Q8.8 hex export in the FPGA bank layout, from_float, PA noise statistics."""
import numpy as np
import pytest
import torch

import dpd_amd  # noqa: F401
from gpu_util import DEV, T, close, close_q, load_golden, ofdm_rows, sd

pytestmark = pytest.mark.gpu


def test_q88_hex_export_byte_exact(tmp_path):
    """The six hex files against what the notebook's own to_q8_8 / export_to_hex wrote for the same weights
    (fixture g13: saturating values and exact ties included), and the one-file bank in the RTL's address order."""
    from dpd_amd.models import TDNNGeneratorQAT
    from dpd_amd.utils import Q88_BANK_OFFSETS, export_q88_flat, export_q88_hex, to_q8_8
    g = load_golden("g13_q88_hex")
    gen = TDNNGeneratorQAT()
    gen.load_state_dict(sd(g, "sd__"))
    gen = gen.to(DEV)
    files = export_q88_hex(gen, tmp_path)
    for i in (1, 2, 3):
        for kind in ("weights", "bias"):
            name = f"fc{i}_{kind}"
            assert files[name].read_bytes() == g[name + "_hex"].tobytes(), name
            fc = getattr(gen, f"fc{i}")
            assert np.array_equal(to_q8_8(fc.weight.data if kind == "weights" else fc.bias.data).cpu().numpy(), g[name + "_q"])
    flat = export_q88_flat(gen).numpy()
    assert flat.dtype == np.int16 and flat.shape == (1554,)
    assert Q88_BANK_OFFSETS == {"fc1_weights": 0, "fc1_bias": 960, "fc2_weights": 992, "fc2_bias": 1504,
                                "fc3_weights": 1520, "fc3_bias": 1552}
    for name, off in Q88_BANK_OFFSETS.items():
        q = g[name + "_q"].reshape(-1)
        assert np.array_equal(flat[off:off + q.size], q), name
    assert files["bank"].read_bytes() == b"".join(g[n + "_hex"].tobytes() for n in
                                                  ("fc1_weights", "fc1_bias", "fc2_weights", "fc2_bias", "fc3_weights", "fc3_bias"))


def test_from_float_copies_weights_and_quantises():
    """TDNNGeneratorQAT.from_float (models/tdnn_generator.py:502-533): same weights; with the quantisers off the
    output of the float model, with them on the oracle's QAT forward of those weights."""
    from oracle import generator as og
    from dpd_amd.models import TDNNGenerator, TDNNGeneratorQAT
    torch.manual_seed(4)
    fl = TDNNGenerator().to(DEV)
    with torch.no_grad():
        fl.fc1.bias.uniform_(-0.1, 0.1)
    q = TDNNGeneratorQAT.from_float(fl)
    assert q.fc1.weight.device == fl.fc1.weight.device and q.qat_enabled
    for name in ("fc1", "fc2", "fc3"):
        assert torch.equal(getattr(q, name).weight, getattr(fl, name).weight)
        assert torch.equal(getattr(q, name).bias, getattr(fl, name).bias)
    iq = ofdm_rows(4, 133, seed=8)
    params = {k: v.detach().cpu().clone() for k, v in fl.state_dict().items() if k in og.PARAM_KEYS}
    close_q(q.eval()(iq.to(DEV)), og.gen_forward(iq, params, qat=True).numpy())
    q.disable_qat()
    assert torch.equal(q(iq.to(DEV)), fl(iq.to(DEV)))


def test_pa_noise_statistics():
    """PADigitalTwin.add_noise (models/pa_digital_twin.py:280-309): additive noise of power
    signal_power * 10^(noise_floor_db / 10) per component and a per-sample phase rotation of standard deviation
    phase_noise_deg; the draws come from the device generator, so parity is that of the distributions."""
    from dpd_amd.models import PADigitalTwin
    x = ofdm_rows(64, 1029, seed=1).to(DEV)
    pa = PADigitalTwin(noise_floor_db=-30.0, phase_noise_deg=0.0).to(DEV)
    torch.manual_seed(0)
    d = pa.add_noise(x) - x
    want = float((x ** 2).mean()) * 10 ** (-30.0 / 10)
    assert abs(float((d ** 2).mean()) / want - 1) < 0.02 and abs(float(d.mean())) < 3 * (want / d.numel()) ** 0.5 * 2
    pa = PADigitalTwin(noise_floor_db=-200.0, phase_noise_deg=2.0).to(DEV)
    y = pa.add_noise(x)
    xc, yc = torch.view_as_complex(x.contiguous()), torch.view_as_complex(y.contiguous())
    keep = xc.abs() > 0.05
    ph = torch.angle(yc[keep] / xc[keep])
    assert abs(float(ph.std()) / (2.0 * np.pi / 180) - 1) < 0.02 and abs(float(ph.mean())) < 1e-3
    assert torch.allclose(yc.abs(), xc.abs(), rtol=1e-5, atol=1e-7)          # a pure rotation
    out = pa(x, add_noise=True)
    assert out.shape == (64, 1025, 2)


def test_augment_signal_matches_notebook_with_recorded_draws():
    """dpd_augment_rows through data.augment_signal against the notebook's own augment_signal run on the same random
    numbers (fixture g14: every augment_type, the three thermal outcomes, two strengths), and against the oracle on
    another length.  Tolerance: fp32 rounding of a different but equivalent operation order (rotation by the thermal
    phase shift instead of atan2 / cos / sin), 2e-6 of the signal's peak."""
    from dpd_amd.data import augment_signal
    from oracle import augment as oa
    g = load_golden("g14_augment")
    sig = torch.from_numpy(g["signal"])
    peak = float(sig.abs().max())
    worst = 0.0
    for i in range(int(g["n_cases"])):
        kind, strength = str(g[f"c{i}_kind"]), float(g[f"c{i}_strength"])
        snr, phase, gain, thermal = (float(v) for v in g[f"c{i}_scalars"])
        draws = dict(snr=snr, phase=phase, gain=gain, thermal=thermal, noise=torch.from_numpy(g[f"c{i}_noise"]).to(DEV))
        got = augment_signal(sig.to(DEV), kind, strength, draws=draws).cpu()
        err = float((got - torch.from_numpy(g[f"c{i}_out"])).abs().max())
        worst = max(worst, err)
        assert err <= 2e-6 * peak, (i, kind, err)
    print(f"augment_signal vs notebook: worst abs error {worst:.3e} (signal peak {peak:.3f})")
    torch.manual_seed(3)
    s2 = torch.randn(777, 2) * 0.4
    nz = torch.randn(777, 2)
    for kind, th in (("all", 0.1), ("all", 0.9), ("thermal", 0.5), ("noise", 0.0)):
        want = oa.augment_signal(s2, kind, 0.2, -0.7, nz, 0.8, 0.1, th)
        got = augment_signal(s2.to(DEV), kind, 0.2, draws=dict(snr=-0.7, phase=0.8, gain=0.1, thermal=th, noise=nz.to(DEV))).cpu()
        assert float((got - want).abs().max()) <= 2e-6 * float(s2.abs().max())
    with pytest.raises(ValueError):
        augment_signal(sig.to(DEV), "fade", 0.1)


def test_pa_add_noise_matches_reference_with_recorded_draws():
    """dpd_pa_add_noise against PADigitalTwin.add_noise of the reference run on the same normal draws (fixture g14)."""
    from dpd_amd.models import PADigitalTwin
    g = load_golden("g14_augment")
    pa = PADigitalTwin(noise_floor_db=float(g["pa_noise_floor_db"]), phase_noise_deg=float(g["pa_phase_noise_deg"])).to(DEV)
    x = torch.from_numpy(g["pa_x"]).to(DEV)
    got = pa.add_noise(x, draws=dict(awgn=torch.from_numpy(g["pa_awgn"]).to(DEV), phase=torch.from_numpy(g["pa_phase"]).to(DEV))).cpu()
    err = float((got - torch.from_numpy(g["pa_out"])).abs().max())
    print(f"add_noise vs reference: worst abs error {err:.3e} (signal peak {float(x.abs().max()):.2f})")
    assert err <= 2e-6 * float(x.abs().max())


def test_kernel_generator_draws_are_standard_normal():
    """The counter-based generator inside the impairment kernels: moments of 2^21 deviates per component, independence
    of the two components, reproducible per seed and different between seeds."""
    from dpd_amd.models import PADigitalTwin
    pa = PADigitalTwin(noise_floor_db=0.0, phase_noise_deg=0.0).to(DEV)     # noise power == signal power, no rotation
    x = torch.full((1, 1 << 21, 2), 0.5, device=DEV)                        # mean(x^2) = 0.25: noise std 0.5
    z = (pa.add_noise(x, seed=11) - x) / 0.5
    assert torch.equal(z, (pa.add_noise(x, seed=11) - x) / 0.5)
    assert not torch.equal(z, (pa.add_noise(x, seed=12) - x) / 0.5)
    zi, zq = z[0, :, 0].double(), z[0, :, 1].double()
    n = zi.numel()
    for v in (zi, zq):
        assert abs(float(v.mean())) < 4 / n ** 0.5
        assert abs(float(v.var()) - 1) < 4 * (2 / n) ** 0.5
        assert abs(float((v ** 4).mean()) - 3) < 0.05
        assert abs(float((v ** 3).mean())) < 0.02
    assert abs(float((zi * zq).mean())) < 4 / n ** 0.5
    assert abs(float((zi[1:] * zi[:-1]).mean())) < 4 / n ** 0.5               # no correlation between neighbours
