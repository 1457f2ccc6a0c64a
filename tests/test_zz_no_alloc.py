"""The allocation-free Stage-1 entry point (dpd_gen_mse_step_noalloc) against dpd_gen_mse_step, called
through the C ABI: same gradients and loss, on two calls in a row (the second on a smaller input, so a
stale slab row or chunk counter of the first call would show), for both kernel families, float and QAT."""
import ctypes

import pytest
import torch

import dpd_amd  # noqa: F401
from dpd_amd import _lib as L
from gpu_util import DEV, ofdm_rows

pytestmark = pytest.mark.gpu


def prepared(qat, iq):
    from dpd_amd.models import TDNNGeneratorQAT
    torch.manual_seed(0)
    gen = TDNNGeneratorQAT().to(DEV)
    with torch.no_grad():
        gen.fc1.bias.uniform_(-0.05, 0.05)
        gen.fc2.bias.uniform_(-0.05, 0.05)
    st = L.stream()
    wq = torch.empty(L.GEN_NPARAM, dtype=torch.float32, device=DEV)
    absmax = torch.zeros(L.GEN_NSCALE, dtype=torch.float32, device=DEV)
    ps = [gen.fc1.weight, gen.fc1.bias, gen.fc2.weight, gen.fc2.bias, gen.fc3.weight, gen.fc3.bias]
    L.call("dpd_gen_prepare_weights", *(L.fptr(p.detach()) for p in ps), int(qat), L.fptr(wq), L.fptr(absmax), st)
    if qat:
        B, S = iq.shape[0], iq.shape[1]
        out = torch.empty(B, S - 5, 2, dtype=torch.float32, device=DEV)
        for stage in range(4):
            L.call("dpd_gen_absmax_pass", stage, L.fptr(iq), B, S, 0, L.fptr(wq), 0.2, L.fptr(absmax), L.fptr(out), 0, st)
    return wq, absmax


def both(iq, target, qat, impl):
    """(grads, loss) of the workspace entry point and of the allocation-free one."""
    B, S = iq.shape[0], iq.shape[1]
    wq, absmax = prepared(qat, iq)
    st = L.stream()
    ref = torch.full((L.GEN_NPARAM + 1,), float("nan"), device=DEV)
    new = torch.full((L.GEN_NPARAM + 1,), float("nan"), device=DEV)
    _, wsp, wsn = L.workspace(iq.device)
    L.call("dpd_gen_mse_step", L.fptr(iq), B, S, L.fptr(target), L.fptr(wq), 0.2, int(qat), L.fptr(absmax), L.fptr(ref),
           L.fptr(ref[L.GEN_NPARAM:]), wsp, wsn, impl, st)
    L.call("dpd_gen_mse_step_noalloc", L.fptr(iq), B, S, L.fptr(target), L.fptr(wq), 0.2, int(qat), L.fptr(absmax),
           L.fptr(new), L.fptr(new[L.GEN_NPARAM:]), impl, st)
    torch.cuda.synchronize()
    return ref.cpu(), new.cpu()


@pytest.mark.parametrize("qat", [False, True])
@pytest.mark.parametrize("impl", [L.IMPL_MFMA, L.IMPL_VALU])
def test_noalloc_matches_workspace_entry_point_twice(qat, impl):
    L.call("dpd_set_option", b"gen_deterministic", 1)      # fixed chunk order: the two entry points agree bit for bit
    try:
        for rows, seq in ((96, 517), (7, 133)):            # second call: smaller input, ragged last chunk
            iq = ofdm_rows(rows, seq, seed=rows).to(DEV)
            target = (iq[:, 5:, :] * 0.9).contiguous()
            ref, new = both(iq, target, qat, impl)
            assert torch.isfinite(new).all()
            assert torch.equal(ref, new), (ref - new).abs().max()
    finally:
        L.call("dpd_set_option", b"gen_deterministic", 0)


def test_noalloc_default_chunk_queue_and_empty_batch():
    """First-come chunk hand-out (the default): equal to fp32 summation order; B = 0 zero-fills."""
    for rows, seq in ((256, 1029), (33, 261)):
        iq = ofdm_rows(rows, seq, seed=3).to(DEV)
        target = (iq[:, 5:, :] * 0.9).contiguous()
        ref, new = both(iq, target, True, L.IMPL_AUTO)
        scale = ref[:L.GEN_NPARAM].abs().max()
        assert (ref - new).abs().max() <= 1e-5 * scale
    g = torch.full((L.GEN_NPARAM + 1,), 7.0, device=DEV)
    wq, absmax = prepared(False, None)
    L.call("dpd_gen_mse_step_noalloc", L.c_p(0), 0, 37, L.c_p(0), L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(g),
           L.fptr(g[L.GEN_NPARAM:]), 0, L.stream())
    assert float(g.abs().max()) == 0.0


def test_noalloc_rejects_oversize_and_never_allocates():
    lib = L.lib()
    wq, absmax = prepared(False, None)
    g = torch.zeros(L.GEN_NPARAM + 1, device=DEV)
    small = torch.zeros(1, 37, 2, device=DEV)
    # a shape beyond the stated maximum is refused before any device work (the pointers are never read)
    rc = lib.dpd_gen_mse_step_noalloc(L.fptr(small), (1 << 20) + 1, 1029, L.fptr(small), L.fptr(wq), 0.2, 0, L.fptr(absmax),
                                      L.fptr(g), L.fptr(g[L.GEN_NPARAM:]), 0, L.stream())
    assert rc == 4 and b"maximum" in lib.dpd_last_error()
    # no device allocation by the call: the allocator's and the driver's free memory stay where they were
    iq = ofdm_rows(64, 517, seed=5).to(DEV)
    target = iq[:, 5:, :].contiguous()
    both(iq, target, False, 0)                      # first use on this device done
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    res0 = torch.cuda.memory_reserved()
    for _ in range(3):
        L.call("dpd_gen_mse_step_noalloc", L.fptr(iq), 64, 517, L.fptr(target), L.fptr(wq), 0.2, 0, L.fptr(absmax), L.fptr(g),
               L.fptr(g[L.GEN_NPARAM:]), 0, L.stream())
    torch.cuda.synchronize()
    assert torch.cuda.memory_reserved() == res0
    assert torch.cuda.mem_get_info()[0] == free0
