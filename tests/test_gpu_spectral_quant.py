"""GPU parity of the spectral metrics/loss, the fixed-point primitives and the Q8.8 integer inference.
Tolerances: dB metrics within 0.1 dB (asserted far tighter), loss scalars rel 1e-5, loss gradients
2e-6 of their largest entry; fixed-point and Q8.8 results bit-exact."""
import os
import tempfile

import numpy as np
import pytest
import torch

from gpu_util import DEV, T, close, load_golden, ofdm_rows

pytestmark = pytest.mark.gpu

# Gradients of the spectral losses: absolute tolerance as a fraction of the largest entry.  Two fp32 transforms
# of up to 4,096 points in different butterfly orders differ by a few 1e-7 of that (measured against an fp64
# evaluation: HIP path 1.0-2.1e-7, the fp32 oracle 1.6-2.9e-7, one against the other up to 3.2e-7).
GTOL = 2e-6


def test_spectral_golden():
    from dpd_amd.utils import (ACPRLoss, EVMLoss, SpectralLoss, compute_acpr, compute_evm, compute_nmse)
    g = load_golden("g8_spectral")
    sl = SpectralLoss()
    for n in (256, 1024):
        pred = T(g[f"L{n}_pred"]).requires_grad_(True)
        tgt = T(g[f"L{n}_target"])
        total, comps = sl(pred, tgt, return_components=True)
        close(total, g[f"L{n}_total"], rtol=1e-5)
        for k in ("l1", "evm", "acpr"):
            close(comps[k], g[f"L{n}_{k}"], rtol=1e-5, atol=1e-7)
        total.backward()
        ref = g[f"L{n}_dpred"]
        close(pred.grad, ref, rtol=1e-5, atol=GTOL * np.abs(ref).max())
        p = pred.detach()
        assert np.abs(compute_evm(p, tgt).cpu().numpy() - g[f"L{n}_evm_db"]).max() < 1e-3
        close(compute_evm(p, tgt, return_db=False), g[f"L{n}_evm_lin"], rtol=1e-5)
        assert np.abs(compute_nmse(p, tgt).cpu().numpy() - g[f"L{n}_nmse_db"]).max() < 1e-3
        close(compute_nmse(p, tgt, return_db=False), g[f"L{n}_nmse_lin"], rtol=1e-5)
        lo, up = compute_acpr(p)
        assert np.abs(lo.cpu().numpy() - g[f"L{n}_acpr_lo_db"]).max() < 1e-2   # bar: 0.1 dB
        assert np.abs(up.cpu().numpy() - g[f"L{n}_acpr_up_db"]).max() < 1e-2
        lo, up = compute_acpr(p, return_db=False)
        close(lo, g[f"L{n}_acpr_lo_lin"], rtol=1e-5); close(up, g[f"L{n}_acpr_up_lin"], rtol=1e-5)
        m = sl.compute_metrics(p, tgt)
        for k, v in m.items():
            ref = float(g[f"L{n}_metric__{k}"])
            assert abs(v - ref) < (1e-2 if k.endswith("_db") else 1e-6), (k, v, ref)
        # complex [B, L] input form
        pc = torch.complex(p[..., 0], p[..., 1])
        tc = torch.complex(tgt[..., 0], tgt[..., 1])
        assert np.abs(compute_evm(pc, tc).cpu().numpy() - g[f"L{n}_evm_db"]).max() < 1e-3
    tgt = T(g["L256_target"])
    pp = T(g["clamp_pred"]).requires_grad_(True)
    total, comps = sl(pp, tgt, return_components=True)
    close(total, g["clamp_total"], rtol=1e-5)
    close(comps["evm"], g["clamp_evm"], rtol=1e-5, atol=1e-7)
    total.backward()
    close(pp.grad, g["clamp_dpred"], rtol=1e-5, atol=GTOL * np.abs(g["clamp_dpred"]).max())
    p256 = T(g["L256_pred"]).requires_grad_(True)
    el = EVMLoss()(p256, tgt)
    close(el, g["evmloss"], rtol=1e-5)
    el.backward()
    close(p256.grad, g["evmloss_d"], rtol=1e-5, atol=GTOL * np.abs(g["evmloss_d"]).max())
    p256 = T(g["L256_pred"]).requires_grad_(True)
    al = ACPRLoss()(p256)
    close(al, g["acprloss"], rtol=1e-5)
    al.backward()
    close(p256.grad, g["acprloss_d"], rtol=1e-5, atol=GTOL * np.abs(g["acprloss_d"]).max())


def test_band_masks_match_reference():
    from dpd_amd.utils.spectral_loss import band_masks
    g = load_golden("g8_spectral")
    for n in (256, 1024, 4096, 100):
        assert np.array_equal(band_masks(n, 200e6, 100e6, 100e6, DEV).cpu().numpy(), g[f"mask{n}"])


@pytest.mark.parametrize("n,rows", [(2, 3), (8, 3), (64, 3), (512, 3), (4096, 3), (256, 1), (256, 7), (1024, 5), (1024, 9)])
def test_spectral_vs_oracle_lengths(n, rows):
    """Every supported FFT length (256 and 1,024 take the register-FFT kernels: four rows / one row per wave,
    row counts that leave lane groups and waves idle), forward statistics and gradient, against the oracle."""
    from oracle import spectral as osp
    from dpd_amd.utils import SpectralLoss
    gen = torch.Generator().manual_seed(n)
    tgt = torch.randn(rows, n, 2, generator=gen) * 0.3
    pred = (tgt + 0.05 * torch.randn(rows, n, 2, generator=gen)).requires_grad_(True)
    ref, comps = osp.spectral_loss(pred, tgt)
    (rg,) = torch.autograd.grad(ref, pred)
    p = pred.detach().to(DEV).requires_grad_(True)
    total = SpectralLoss()(p, tgt.to(DEV))
    close(total, ref.detach(), rtol=2e-5)
    total.backward()
    close(p.grad, rg, rtol=1e-5, atol=GTOL * float(rg.abs().max()))


def test_spectral_1024_rows_on_an_8_byte_boundary():
    """Rows of 1,024 samples take 16-byte loads when the tensors start on a 16-byte boundary and 8-byte loads
    otherwise (a view that starts at an odd sample): both forms against the oracle, and against each other."""
    from oracle import spectral as osp
    from dpd_amd import _lib as L
    from dpd_amd.utils.spectral_loss import band_masks
    gen = torch.Generator().manual_seed(7)
    rows, n = 6, 1024
    tgt = torch.randn(rows, n, 2, generator=gen) * 0.3
    pred = tgt + 0.05 * torch.randn(rows, n, 2, generator=gen)
    gst = torch.rand(rows, 8, generator=gen) * 1e-2
    band = band_masks(n, 200e6, 100e6, 100e6, DEV)
    res = []
    for shift in (0, 1):           # 1: every tensor one sample (8 bytes) off a 16-byte boundary
        def dev(t):
            buf = torch.zeros(t.numel() // 2 + 2, 2, device=DEV)
            view = buf[shift:shift + t.numel() // 2]
            view.copy_(t.reshape(-1, 2))
            assert view.data_ptr() % 16 == 8 * shift
            return view
        p, t, dp = dev(pred), dev(tgt), dev(torch.zeros_like(pred))
        stats = torch.empty(rows, 8, device=DEV)
        L.call("dpd_spectral_stats", L.fptr(p), L.fptr(t), rows, n, L.fptr(band, torch.uint8), L.fptr(stats), L.stream())
        L.call("dpd_spectral_backward", L.fptr(p), L.fptr(t), rows, n, L.fptr(band, torch.uint8), L.fptr(gst.to(DEV)), L.fptr(dp), L.stream())
        res.append((stats.cpu(), dp.cpu().reshape(rows, n, 2)))
    ref_stats = osp.row_stats(pred, tgt) if hasattr(osp, "row_stats") else None
    close(res[1][0], res[0][0], rtol=1e-6, atol=1e-7)
    close(res[1][1], res[0][1], rtol=1e-5, atol=1e-6 * float(res[0][1].abs().max()))
    if ref_stats is not None:
        close(res[0][0][:, :6], ref_stats[:, :6], rtol=2e-5, atol=1e-6)


def test_spectral_other_lengths():
    """Row lengths that are not a power of two: metrics through the direct DFT (any length up to 8192, as
    torch.fft.fft in the reference takes any length) against the oracle; longer rows raise."""
    from oracle import spectral as osp
    from dpd_amd.utils import SpectralLoss, compute_acpr, compute_evm
    for n in (100, 1000, 4087, 6):
        x = ofdm_rows(3, n + 5, seed=n)[:, 5:, :].contiguous() + 0.01 * torch.randn(3, n, 2, generator=torch.Generator().manual_seed(n))
        lo, up = compute_acpr(x.to(DEV))
        rlo, rup = osp.acpr(x)
        assert float((lo.cpu() - rlo).abs().max()) < 0.01 and float((up.cpu() - rup).abs().max()) < 0.01
        m = SpectralLoss().compute_metrics(x.to(DEV), (x * 0.9).to(DEV))
        r = osp.metrics(x, x * 0.9)
        for k in r:
            assert abs(m[k] - r[k]) < (0.01 if k.endswith("_db") else 1e-5), (n, k, m[k], r[k])
    x = torch.randn(2, 100, 2, device=DEV)
    with pytest.raises(NotImplementedError):
        compute_acpr(torch.randn(1, 8200, 2, device=DEV))
    compute_evm(x, x * 0.9)          # no transform needed: any length
    # the longest row of the direct-DFT path (its LDS request is above the 64 KiB a launch gets by default)
    x = ofdm_rows(1, 8191 + 5, seed=3)[:, 5:, :].contiguous()
    lo, up = compute_acpr(x.to(DEV))
    rlo, rup = osp.acpr(x)
    assert float((lo.cpu() - rlo).abs().max()) < 0.01 and float((up.cpu() - rup).abs().max()) < 0.01


@pytest.mark.parametrize("n,rows", [(100, 3), (1000, 2), (4087, 2), (6, 4), (3, 2), (8191, 1)])
def test_spectral_gradient_other_lengths(n, rows):
    """SpectralLoss and its gradient on rows whose length is not a power of two (both transforms as direct sums over
    the tabulated twiddles), against autograd through the oracle; the longest row asks for 128 KiB of LDS."""
    from oracle import spectral as osp
    from dpd_amd.utils import SpectralLoss, compute_acpr
    gen = torch.Generator().manual_seed(n)
    tgt = torch.randn(rows, n, 2, generator=gen) * 0.3
    pred = (tgt + 0.05 * torch.randn(rows, n, 2, generator=gen)).requires_grad_(True)
    ref, _ = osp.spectral_loss(pred, tgt)
    (rg,) = torch.autograd.grad(ref, pred)
    p = pred.detach().to(DEV).requires_grad_(True)
    total = SpectralLoss()(p, tgt.to(DEV))
    close(total, ref.detach(), rtol=2e-5)
    total.backward()
    # direct sums of up to 8,191 terms per output in fp32: the allowance grows with the square root of the length
    close(p.grad, rg, rtol=1e-5, atol=GTOL * max(1.0, (n / 1024) ** 0.5) * float(rg.abs().max()))
    # the ACPR functions carry a gradient as well (the reference's ACPRLoss path)
    q = pred.detach().to(DEV).requires_grad_(True)
    lo, up = compute_acpr(q, return_db=False)
    (lo.sum() + 2 * up.sum()).backward()
    pr = pred.detach().clone().requires_grad_(True)
    rlo, rup = osp.acpr(pr, db=False)
    (rlo.sum() + 2 * rup.sum()).backward()
    close(q.grad, pr.grad, rtol=1e-5, atol=GTOL * max(1.0, (n / 1024) ** 0.5) * float(pr.grad.abs().max()))


def test_fixed_point_primitives_bit_exact():
    from dpd_amd.utils import (QuantizationConfig, QuantizedLinear, dequantize_from_fixed_point, export_weights_binary,
                               fake_quantize, generate_leaky_relu_params, generate_tanh_lut, load_weights_binary,
                               quantize_to_fixed_point)
    g = load_golden("g4_fixed_point")
    cfg = QuantizationConfig()
    eq = lambda a, b: np.array_equal(a.cpu().numpy(), b)
    assert eq(quantize_to_fixed_point(T(g["edge15"]), cfg.weight_scale, 16), g["q15"])
    assert eq(quantize_to_fixed_point(T(g["edge8"]), cfg.activation_scale, 16), g["q8"])
    assert eq(quantize_to_fixed_point(T(g["rnd"]), cfg.activation_scale, 16), g["q8r"])
    assert eq(quantize_to_fixed_point(T(g["edge8"]), 16.0, 8, signed=False), g["qu8"])
    assert eq(fake_quantize(T(g["edge15"]), cfg.weight_scale, 16), g["fq15"])
    assert eq(fake_quantize(T(g["rnd"]), cfg.activation_scale, 16), g["fq8r"])
    assert eq(dequantize_from_fixed_point(T(g["q8"]), cfg.activation_scale), g["dq8"])
    assert cfg.weight_range == (-32768, 32767) and cfg.activation_range == (-32768, 32767) and cfg.io_scale == 32768
    ql = QuantizedLinear(30, 32, cfg)
    with torch.no_grad():
        ql.weight.copy_(torch.from_numpy(g["ql_weight"])); ql.bias.copy_(torch.from_numpy(g["ql_bias"]))
    ql = ql.to(DEV)
    x = T(g["ql_x"]).requires_grad_(True)
    y = ql(x)
    # the sum is formed exactly on the device (Q8.8 x Q1.15 products are integers / 2^23): bit-exact
    assert np.array_equal(y.detach().cpu().numpy(), g["ql_y"])
    y.backward(T(g["ql_gy"]))
    close(x.grad, g["ql_gx"], rtol=1e-4, atol=1e-6); close(ql.weight.grad, g["ql_gw"], rtol=1e-4, atol=1e-6)
    close(ql.bias.grad, g["ql_gb"], rtol=1e-5, atol=1e-6)
    qw = ql.get_quantized_weights()
    assert eq(qw["weight"], g["ql_w_int"]) and eq(qw["bias"], g["ql_b_int"])
    assert np.array_equal(generate_tanh_lut().numpy(), g["tanh_lut"])
    assert np.array_equal(generate_tanh_lut(num_entries=16).numpy(), g["tanh_lut16"])
    lp = generate_leaky_relu_params(0.2)
    assert lp["shift_amount"] == int(g["lrelu_shift"]) and lp["actual_slope"] == float(g["lrelu_slope"])
    wd = {"fc1_weight": torch.from_numpy(g["bin_fc1_weight"]), "fc1_bias": torch.from_numpy(g["bin_fc1_bias"])}
    with tempfile.TemporaryDirectory() as d:
        for en, ref in (("little", g["bin_blob_le"]), ("big", g["bin_blob_be"])):
            p = os.path.join(d, en + ".bin")
            export_weights_binary(wd, p, endianness=en)
            assert open(p, "rb").read() == ref.tobytes()
            back = load_weights_binary(p, endianness=en)
            assert all(torch.equal(back[k], wd[k]) for k in wd)
        open(os.path.join(d, "bad.bin"), "wb").write(b"\x00" * 16)
        with pytest.raises(ValueError):
            load_weights_binary(os.path.join(d, "bad.bin"))


@pytest.mark.parametrize("kernel", ["matrix", "valu32", "valu64"])
def test_q88_golden_bit_exact(kernel):
    """The three kernels of the integer path (int8 matrix cores, integer VALU with 32-bit and with 64-bit
    accumulators) against the fixture, bit for bit."""
    from dpd_amd.utils import Q88Inference
    g = load_golden("g11_q88")
    w = {k: T(g[k]) for k in ("w1", "b1", "w2", "b2", "w3", "b3", "lut")}
    eng = Q88Inference(w, force_acc64=kernel == "valu64", force_valu=kernel == "valu32")
    # these weights are provably safe for 32-bit accumulation and split into 8-bit digits: the default is the matrix form
    assert eng.mode == {"matrix": 2, "valu32": 0, "valu64": 1}[kernel] and eng.acc64 == (kernel == "valu64")
    out = eng(T(g["iq_i16"]))
    assert np.array_equal(out.cpu().numpy()[0], g["out_i16"])
    with pytest.raises(ValueError):
        eng(torch.zeros(1, 5, 2, dtype=torch.int16, device=DEV))


def test_q88_random_rows_and_extremes_vs_oracle():
    """Random full-range int16 IQ (exercises saturation), perfect-square envelopes (exact ties), ragged
    row length, and weights large enough that the 32-bit proof fails (64-bit path is selected)."""
    from oracle import q88 as oq
    from dpd_amd.utils import Q88Inference
    g = load_golden("g11_q88")
    rng = np.random.default_rng(5)
    iq = rng.integers(-32768, 32768, size=(3, 203, 2), dtype=np.int64).astype(np.int16)
    iq[0, :40, 0] = (np.arange(40) * 192 + 64).astype(np.int16); iq[0, :40, 1] = 0     # sqrt(s)/128 on .5 ties
    iq[1, 7] = (32767, -32768); iq[1, 8] = (-32768, -32768); iq[2, 9] = (1024, 1024)
    base = [g[k] for k in ("w1", "b1", "w2", "b2", "w3", "b3")]
    big = [rng.integers(-32768, 32768, size=a.shape, dtype=np.int64).astype(np.int16) for a in base]
    # biases at the ends of the int16 range: pre-activations beyond +-32768 (saturation on both sides, in both
    # layers) while the weights stay small enough for the 32-bit proof
    edge = [a.copy() for a in base]
    edge[1][:4] = (32767, -32768, 32767, -32768)
    edge[3][:4] = (32767, -32768, -32768, 32767)
    for ws, expect64 in ((base, False), (edge, False), (big, True)):
        w = {k: T(a) for k, a in zip(("w1", "b1", "w2", "b2", "w3", "b3"), ws)}
        w["lut"] = T(g["lut"])
        eng = Q88Inference(w)
        assert eng.acc64 == expect64
        out = eng(T(iq)).cpu().numpy()
        ref = oq.infer_rows(iq, *ws, g["lut"])
        assert np.array_equal(out, ref)
        if not expect64:      # the other two kernels on the same weights
            assert eng.mode == 2
            assert np.array_equal(Q88Inference(w, force_acc64=True)(T(iq)).cpu().numpy(), ref)
            assert np.array_equal(Q88Inference(w, force_valu=True)(T(iq)).cpu().numpy(), ref)
    # a weight above 32639 has no signed 8-bit high digit: the matrix form is not selected
    top = [a.copy() for a in base]
    top[0][3, 7] = 32767
    w = {k: T(a) for k, a in zip(("w1", "b1", "w2", "b2", "w3", "b3"), top)}
    w["lut"] = T(g["lut"])
    eng = Q88Inference(w)
    assert eng.mode == 0
    assert np.array_equal(eng(T(iq)).cpu().numpy(), oq.infer_rows(iq, *top, g["lut"]))
    # weights at the ends of what the digits hold (32639, -32768), ragged rows through the run walk of the matrix form
    ext = [a.copy() for a in base]
    ext[0][0, :4] = (32639, -32768, 32639, -32768)
    ext[2][1, :2] = (32639, -32768)
    ext[4][0, 0] = 32639
    w = {k: T(a) for k, a in zip(("w1", "b1", "w2", "b2", "w3", "b3"), ext)}
    w["lut"] = T(g["lut"])
    eng = Q88Inference(w)
    if eng.mode == 2:         # (the 32-bit proof may fail for such weights: then the 64-bit kernel serves them)
        for shape in ((3, 203), (70, 69), (5, 1029), (33, 6)):
            x = rng.integers(-32768, 32768, size=shape + (2,), dtype=np.int64).astype(np.int16)
            assert np.array_equal(eng(T(x)).cpu().numpy(), oq.infer_rows(x, *ext, g["lut"])), shape
    for shape in ((70, 69), (5, 1029), (33, 6), (1, 70), (2, 134)):
        x = rng.integers(-32768, 32768, size=shape + (2,), dtype=np.int64).astype(np.int16)
        w = {k: T(a) for k, a in zip(("w1", "b1", "w2", "b2", "w3", "b3"), base)}
        w["lut"] = T(g["lut"])
        assert np.array_equal(Q88Inference(w)(T(x)).cpu().numpy(), oq.infer_rows(x, *base, g["lut"])), shape


def test_q88_export_from_generator():
    from dpd_amd.models import TDNNGeneratorQAT
    from dpd_amd.utils import Q88Inference, export_q88_weights
    from oracle import q88 as oq
    torch.manual_seed(3)
    gen = TDNNGeneratorQAT().to(DEV)
    w = export_q88_weights(gen)
    assert w["w1"].dtype == torch.int16 and tuple(w["w1"].shape) == (32, 30) and tuple(w["lut"].shape) == (256,)
    iq = (ofdm_rows(2, 133, seed=4) * 32768).round().clamp(-32768, 32767).to(torch.int16)
    out = Q88Inference(w)(iq.to(DEV)).cpu().numpy()
    ref = oq.infer_rows(iq.numpy(), *(w[k].cpu().numpy() for k in ("w1", "b1", "w2", "b2", "w3", "b3", "lut")))
    assert np.array_equal(out, ref)


def test_q88_long_prefix_bit_exact_vs_oracle():
    """2^20 outputs (1024 rows of 1029) of the integer path against the numpy int64 oracle, bit for bit,
    with trained-like weights exported from a seeded generator."""
    from oracle import q88 as oq
    from dpd_amd.models import TDNNGeneratorQAT
    from dpd_amd.utils import Q88Inference, export_q88_weights
    from gpu_util import ofdm_rows
    torch.manual_seed(21)
    gen = TDNNGeneratorQAT()
    with torch.no_grad():
        for fc in (gen.fc1, gen.fc2, gen.fc3):
            fc.bias.uniform_(-0.2, 0.2)
    w = export_q88_weights(gen.to(DEV))
    iq = (ofdm_rows(1024, 1029, seed=8) * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    out = Q88Inference(w)(iq.to(DEV)).cpu().numpy()
    wn = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in w.items()}
    ref = oq.infer_rows(iq.numpy(), wn["w1"], wn["b1"], wn["w2"], wn["b2"], wn["w3"], wn["b3"], wn["lut"])
    assert out.shape == ref.shape and np.array_equal(out, ref)
