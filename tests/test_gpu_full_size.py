"""Properties checked at the benchmark's full size (16384 rows x 1029 samples = 2^24 outputs per GPU), where
the CPU oracle is too slow to be the comparator: the two kernel families against each other, row-permutation
equivariance (rows are independent units), additivity of gradient sums over a split of the batch, exact
power-of-two scaling of the spectral statistics, and a 64-row prefix against the oracle."""
import numpy as np
import pytest
import torch

from gpu_util import DEV, close

pytestmark = pytest.mark.gpu

ROWS, SEQ = 16384, 1029


@pytest.fixture(scope="module")
def big():
    import dpd_amd  # noqa: F401
    from dpd_amd.data import ofdm_rows
    from dpd_amd.models import TDNNGeneratorQAT
    torch.manual_seed(17)
    gen = TDNNGeneratorQAT().to(DEV)
    with torch.no_grad():
        for fc in (gen.fc1, gen.fc2, gen.fc3):
            fc.bias.uniform_(-0.1, 0.1)
    iq = ofdm_rows(ROWS, SEQ, seed=77, device=DEV)
    perm = torch.randperm(ROWS, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    return gen, iq, perm


@pytest.mark.parametrize("qat", [False, True])
def test_generator_families_agree_and_rows_are_independent(big, qat):
    gen, iq, perm = big
    gen.qat_enabled = qat
    gen.eval()
    with torch.no_grad():
        gen.kernel_impl = 2
        y_mfma = gen(iq)
        y_perm = gen(iq[perm])
        gen.kernel_impl = 1
        y_valu = gen(iq[:2048])
        gen.kernel_impl = 0
    assert torch.equal(y_perm, y_mfma[perm])            # bit-exact: no state crosses rows (abs-max is order-free)
    if qat:                                             # the VALU run sees 2048 rows only: its scales differ
        return
    d = (y_mfma[:2048] - y_valu).abs().max()
    assert float(d) <= 1e-5 * float(y_valu.abs().max()) + 1e-6


def test_oracle_prefix_of_full_size_forward(big):
    from oracle import generator as og
    gen, iq, _ = big
    gen.qat_enabled = False
    with torch.no_grad():
        y = gen(iq)[:64].cpu()
    params = {k: v.detach().cpu() for k, v in gen.state_dict().items() if k in og.PARAM_KEYS}
    ref = og.gen_forward(iq[:64].cpu(), params, qat=False)
    close(y, ref.numpy(), rtol=1e-5, atol=1e-6)


def test_stage1_gradient_sums_are_additive_over_a_batch_split(big):
    from dpd_amd.training import Stage1Step
    gen, iq, _ = big
    gen.qat_enabled = False
    tgt = (iq[:, 5:, :] * 0.9).contiguous()
    step = Stage1Step(gen, optimizer=torch.optim.SGD(gen.parameters(), lr=0.0))

    def grads(a, b):
        loss = step(iq[a:b], tgt[a:b]).clone()
        return step.grads[:1554].clone(), loss
    g_all, l_all = grads(0, ROWS)
    g_lo, l_lo = grads(0, ROWS // 2)
    g_hi, l_hi = grads(ROWS // 2, ROWS)
    # mean-reduced loss: full = average of the halves
    close(l_all, (0.5 * (l_lo + l_hi)).cpu().numpy(), rtol=1e-5, atol=0)
    close(g_all, (0.5 * (g_lo + g_hi)).cpu().numpy(), rtol=1e-5, atol=1e-6 * float(g_all.abs().max()))


def test_critic_mean_score_additive_and_gp_row_independent(big):
    from dpd_amd.models import Discriminator, compute_gradient_penalty
    _, iq, perm = big
    torch.manual_seed(5)
    disc = Discriminator().to(DEV).eval()
    x, c = iq[:, 5:, :].contiguous(), (iq[:, 5:, :] * 0.8).contiguous()
    with torch.no_grad():
        m_all = disc.mean_score(x, c)
        m_lo, m_hi = disc.mean_score(x[:ROWS // 2], c[:ROWS // 2]), disc.mean_score(x[ROWS // 2:], c[ROWS // 2:])
    close(m_all, (0.5 * (m_lo + m_hi)).cpu().numpy(), rtol=1e-5, atol=1e-6)
    B = 4096
    alpha = torch.rand(B, 1, 1, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    p = perm[perm < B]
    gp = compute_gradient_penalty(disc, x[:B], c[:B] * 1.1, c[:B], DEV, alpha=alpha)
    gp_p = compute_gradient_penalty(disc, x[:B][p], (c[:B] * 1.1)[p], c[:B][p], DEV, alpha=alpha[p])
    close(gp, gp_p.detach().cpu().numpy(), rtol=1e-5, atol=0)


def test_gradient_penalty_one_pass_and_three_pass_agree_at_full_size(big):
    """dpd_critic_gp on the benchmark's 16384 rows x 1024 samples: the one-pass kernel against the three-pass form,
    and the weight gradient additive over a split of the rows (the penalty is a mean over rows)."""
    from dpd_amd import _lib as L
    _, iq, _ = big
    real = iq[:, 5:, :].contiguous()
    fake, cond = (real * 1.1).contiguous(), (real * 0.8).contiguous()
    B, Lr = real.shape[0], real.shape[1]
    alpha = torch.rand(B, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    torch.manual_seed(6)
    wb = torch.randn(2945, device=DEV) * 0.2
    ws, wsp, wsn = L.workspace(DEV, B * Lr)
    st = L.stream()

    def run(lo, hi):
        n = hi - lo
        gp, rn, gg = torch.zeros(1, device=DEV), torch.empty(n, device=DEV), torch.empty(2945, device=DEV)
        L.call("dpd_critic_gp", L.fptr(real[lo:hi]), L.fptr(fake[lo:hi]), L.fptr(cond[lo:hi]), L.fptr(alpha[lo:hi]), n, Lr,
               L.fptr(wb), 0.2, L.fptr(gp), L.fptr(rn), L.fptr(gg), wsp, wsn, st)
        torch.cuda.synchronize()
        return gp.cpu(), rn.cpu(), gg.cpu()
    one = run(0, B)
    L.call("dpd_set_option", b"gp_fused", 0)
    try:
        three = run(0, B)
    finally:
        L.call("dpd_set_option", b"gp_fused", 1)
    assert torch.equal(one[1], three[1])
    close(one[0], three[0].numpy(), rtol=1e-6)
    scale = float(three[2].abs().max())
    assert scale > 0
    close(one[2], three[2].numpy(), rtol=5e-5, atol=5e-6 * scale)
    lo, hi = run(0, B // 2), run(B // 2, B)
    close(one[0], (0.5 * (lo[0] + hi[0])).numpy(), rtol=1e-5)
    close(one[2], (0.5 * (lo[2] + hi[2])).numpy(), rtol=1e-5, atol=5e-7 * scale)


def test_spectral_statistics_scale_exactly(big):
    from dpd_amd import _lib as L
    from dpd_amd.utils.spectral_loss import band_masks
    _, iq, _ = big
    p = iq[:, 5:, :].contiguous()
    t = (p * 0.5).contiguous()
    band = band_masks(1024, 200e6, 100e6, 100e6, p.device)

    def stats(a, b):
        s = torch.empty(ROWS, 8, device=DEV)
        L.call("dpd_spectral_stats", L.fptr(a), L.fptr(b), ROWS, 1024, L.fptr(band, torch.uint8), L.fptr(s), L.stream())
        return s
    s1, s2 = stats(p, t), stats((p * 2).contiguous(), (t * 2).contiguous())
    assert torch.equal(s2[:, 0], s1[:, 0] * 2)           # sums of |.|: scale by 2, exactly (power of two)
    assert torch.equal(s2[:, 1:6], s1[:, 1:6] * 4)       # energies and band powers: by 4, exactly
    # signal confined to the main channel: adjacent-channel power far below
    assert float((s1[:, 4] / s1[:, 3]).max()) < 1e-3 and float((s1[:, 5] / s1[:, 3]).max()) < 1e-3


def test_q88_full_size_chunk_invariance_and_prefix(big):
    from oracle import q88 as oq
    from dpd_amd.utils import Q88Inference, export_q88_weights
    gen, iq, perm = big
    w = export_q88_weights(gen)
    eng = Q88Inference(w)
    iq16 = (iq * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    out = eng(iq16)
    assert torch.equal(eng(iq16[perm]), out[perm])
    assert torch.equal(torch.cat([eng(iq16[:5000]), eng(iq16[5000:])]), out)
    wn = {k: v.cpu().numpy() for k, v in w.items()}
    ref = oq.infer_rows(iq16[:32].cpu().numpy(), wn["w1"], wn["b1"], wn["w2"], wn["b2"], wn["w3"], wn["b3"], wn["lut"])
    assert np.array_equal(out[:32].cpu().numpy(), ref)


@pytest.mark.parametrize("qat", [False, True])
def test_fused_chain_row_walks_agree_at_full_size(big, qat):
    """Fused generator -> PA forward: whole rows per wave with carried outputs (what the default picks at this
    shape) against overlapping chunks, bit for bit (the arithmetic per output is the same), and a 32-row prefix
    against the PA module applied to the generator module's output."""
    from dpd_amd import _lib as L
    from dpd_amd.models import PADigitalTwin, dpd_pa_chain
    gen, iq, _ = big
    gen.qat_enabled = qat
    gen.eval()
    pa = PADigitalTwin().to(DEV)
    try:
        L.call("dpd_set_option", b"chain_carry", 0)
        d0, y0 = dpd_pa_chain(gen, pa, iq)
        L.call("dpd_set_option", b"chain_carry", 1)
        d1, y1 = dpd_pa_chain(gen, pa, iq)
    finally:
        L.call("dpd_set_option", b"chain_carry", 2)
    d2, y2 = dpd_pa_chain(gen, pa, iq)
    assert torch.equal(d0, d1) and torch.equal(y0, y1)
    assert torch.equal(d2, d1) and torch.equal(y2, y1)
    if not qat:     # (with QAT the modules on 32 rows see other scales than the whole batch)
        with torch.no_grad():
            y_mod = pa(gen(iq[:32]), add_noise=False)
        close(y1[:32], y_mod.cpu().numpy(), rtol=1e-5, atol=2e-6 * float(y_mod.abs().max()))
