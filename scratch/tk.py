"""Kernel times of the library named by DPD_HIP_LIBRARY (default: the product build) at the bench shape, through the
C ABI with HIP events around every call: tk.py [critic] [gp] [gen] [q88] [spec] [reps=N]
Prints one line per kernel: median and minimum over the repetitions, and a checksum of the result."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dpd_amd  # noqa: E402,F401
from dpd_amd import _lib as L  # noqa: E402
from dpd_amd.data import ofdm_rows  # noqa: E402

what = [a for a in sys.argv[1:] if "=" not in a] or ["critic", "gp", "gen", "q88", "spec"]
reps = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("reps=")), 20)
dev = torch.device("cuda:0")
B, S = 16384, 1029
Lr = S - 5
N = B * Lr
name = os.path.basename(os.environ.get("DPD_HIP_LIBRARY", "product"))
st = L.stream()
for kv in os.environ.get("DPD_OPTS", "").split():      # library options for this run, e.g. DPD_OPTS="critic_deterministic=1"
    k, v = kv.split("=")
    L.call("dpd_set_option", k.encode(), int(v))
iq = ofdm_rows(B, S, seed=1234, device=dev)
tgt = iq[:, 5:, :].contiguous()


def timeit(label, fn, check=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    c = "" if check is None else "  check %.9g" % float(check().double().sum().item())
    print("%-22s %-34s median %.4f ms  min %.4f ms%s" % (name, label, statistics.median(ms), min(ms), c), flush=True)


if "critic" in what or "gp" in what:
    from dpd_amd.models import Discriminator
    torch.manual_seed(1)
    disc = Discriminator().to(dev).train()
    wb = disc.flat_weights().detach().contiguous()
    ws, wsp, wsn = L.workspace(dev, N)
    x = tgt.reshape(-1, 2).contiguous()
    fake = (tgt * 0.9 + 0.01).reshape(-1, 2).contiguous()
    grads = torch.zeros(L.CRITIC_NPARAM, device=dev)
    gx = torch.empty_like(x)
    ssum = torch.zeros(1, device=dev)
    if "critic" in what:
        timeit("critic_bwd<wgrad>", lambda: L.call("dpd_critic_backward", L.fptr(fake), L.fptr(x), N, L.fptr(wb), 0.2, L.c_p(0), 1.0 / N,
                                                   L.fptr(grads), L.c_p(0), L.fptr(ssum), wsp, wsn, st), lambda: grads.abs())
        timeit("critic_bwd<dx>", lambda: L.call("dpd_critic_backward", L.fptr(fake), L.fptr(x), N, L.fptr(wb), 0.2, L.c_p(0), 1.0 / N,
                                                L.c_p(0), L.fptr(gx), L.fptr(ssum), wsp, wsn, st), lambda: gx.abs())
    if "gp" in what:
        alpha = torch.rand(B, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        gp = torch.zeros(1, device=dev)
        rn = torch.empty(B, device=dev)
        timeit("critic_gp", lambda: L.call("dpd_critic_gp", L.fptr(x), L.fptr(fake), L.fptr(x), L.fptr(alpha), B, Lr, L.fptr(wb), 0.2,
                                           L.fptr(gp), L.fptr(rn), L.fptr(grads), wsp, wsn, st), lambda: grads.abs())

if "gen" in what:
    from dpd_amd.models import PADigitalTwin, TDNNGeneratorQAT
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(dev)
    pa = PADigitalTwin().to(dev)
    ps = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
    wq = torch.empty(L.GEN_NPARAM, device=dev)
    absmax = torch.zeros(L.GEN_NSCALE, device=dev)
    out = torch.empty(B, Lr, 2, device=dev)
    cre, cim = pa.pa_model.coeffs_real.contiguous(), pa.pa_model.coeffs_imag.contiguous()
    yb = torch.empty(B, S - 9, 2, device=dev)
    ws, wsp, wsn = L.workspace(dev, N)
    grads = torch.zeros(L.GEN_NPARAM + 2, device=dev)
    loss = grads[L.GEN_NPARAM:L.GEN_NPARAM + 1]
    L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), 0, L.fptr(wq), L.fptr(absmax), st)
    timeit("gen_fwd float", lambda: L.call("dpd_gen_forward", L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(out), 2, st), lambda: out)
    timeit("gen_pa_fwd float", lambda: L.call("dpd_gen_pa_forward", L.fptr(iq), B, S, L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(cre), L.fptr(cim), 4, 5,
                                              L.c_p(0), L.fptr(yb), st), lambda: yb)
    timeit("gen_mse_step float", lambda: L.call("dpd_gen_mse_step", L.fptr(iq), B, S, L.fptr(tgt), L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(grads),
                                                L.fptr(loss), wsp, wsn, 0, st), lambda: grads[:1554].abs())
    L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), 1, L.fptr(wq), L.fptr(absmax), st)
    for stage in range(4):
        timeit("qat pass %d" % stage, lambda: L.call("dpd_gen_absmax_pass", stage, L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.c_p(0), 0, st),
               lambda: absmax[:4])
    timeit("gen_pa_fwd qat", lambda: L.call("dpd_gen_pa_forward", L.fptr(iq), B, S, L.fptr(wq), 0.2, 1, L.fptr(absmax), L.fptr(cre), L.fptr(cim), 4, 5,
                                            L.c_p(0), L.fptr(yb), st), lambda: yb)
    tg2 = (tgt[:, 4:, :] * 31.0).contiguous()
    timeit("gen_pa_mse_step qat", lambda: L.call("dpd_gen_pa_mse_step", L.fptr(iq), B, S, L.fptr(tg2), L.fptr(wq), 0.2, 1, L.fptr(absmax), L.fptr(cre),
                                                 L.fptr(cim), 4, 5, L.fptr(grads), L.fptr(loss), wsp, wsn, st), lambda: grads[:1554].abs())
    timeit("gen_mse_step qat", lambda: L.call("dpd_gen_mse_step", L.fptr(iq), B, S, L.fptr(tgt), L.fptr(wq), 0.2, 1, L.fptr(absmax), L.fptr(grads),
                                              L.fptr(loss), wsp, wsn, 0, st), lambda: grads[:1554].abs())

if "q88" in what:
    from dpd_amd.models import TDNNGeneratorQAT
    from dpd_amd.utils import Q88Inference, export_q88_weights
    torch.manual_seed(0)
    eng = Q88Inference(export_q88_weights(TDNNGeneratorQAT().to(dev)))
    iq16 = (iq * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    o16 = torch.empty(B, Lr, 2, dtype=torch.int16, device=dev)
    timeit("q88 (mode %d)" % eng.mode, lambda: eng(iq16, out=o16), lambda: o16.to(torch.int64).abs())

if "spec" in what:
    from dpd_amd.utils import spectral_loss as _sl
    pred = (tgt * 0.9).contiguous()
    stats = torch.empty(B, 8, device=dev)
    gst = torch.rand(B, 8, device=dev) * 1e-3
    dp = torch.empty_like(pred)
    band = _sl.band_masks(Lr, 200e6, 100e6, 100e6, dev)
    timeit("spectral_stats", lambda: L.call("dpd_spectral_stats", L.fptr(pred), L.fptr(tgt), B, Lr, L.fptr(band, torch.uint8), L.fptr(stats), st), lambda: stats)
    timeit("spectral_backward", lambda: L.call("dpd_spectral_backward", L.fptr(pred), L.fptr(tgt), B, Lr, L.fptr(band, torch.uint8), L.fptr(gst), L.fptr(dp), st),
           lambda: dp.abs())
