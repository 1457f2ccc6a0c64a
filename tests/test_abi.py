"""CPU-side checks of the C ABI and the host layer: the built library loads without a GPU and
exports every symbol the header declares; the host layer refuses CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dpd_hip.h")
LIB = os.path.join(ROOT, "6g-pa-gan-dpd_amd", "libdpd_hip.so")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dpd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for name in ("dpd_gen_forward", "dpd_gen_backward", "dpd_gen_mse_step", "dpd_gen_absmax_pass", "dpd_tap_assembly",
                 "dpd_pa_forward", "dpd_pa_backward", "dpd_quant_pa_forward", "dpd_gen_pa_forward", "dpd_critic_sn_forward", "dpd_critic_sn_backward", "dpd_critic_forward", "dpd_critic_backward",
                 "dpd_critic_gp", "dpd_spectral_stats", "dpd_spectral_backward", "dpd_spectral_rowdb", "dpd_spectral_loss_final", "dpd_spectral_terms_backward", "dpd_quantize_fixed",
                 "dpd_fake_quantize_dynamic", "dpd_q88_infer", "dpd_q88_int32_safe", "dpd_q88_digits_fit", "dpd_q88_matrix_image_bytes",
                 "dpd_q88_matrix_prepare", "dpd_q88_matrix_infer", "dpd_workspace_bytes"):
        assert name in syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "libdpd_hip.so is not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.dpd_last_error.restype = ctypes.c_char_p
    assert lib.dpd_version() >= 1
    lib.dpd_workspace_bytes.restype = ctypes.c_size_t
    lib.dpd_workspace_bytes.argtypes = [ctypes.c_int64]
    assert lib.dpd_workspace_bytes(1000) - lib.dpd_workspace_bytes(0) == 28000


def test_binding_table_matches_header():
    import dpd_amd  # noqa: F401
    from dpd_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_symbols()
    _lib.lib()


def test_argument_errors_without_gpu():
    """Argument validation happens before any device work, so it can be exercised on the CPU."""
    lib = ctypes.CDLL(LIB)
    lib.dpd_last_error.restype = ctypes.c_char_p
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    lib.dpd_tap_assembly.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.dpd_tap_assembly(p, 1, 5, p, null) == 2          # DPD_ERR_SHAPE: S <= memory depth
    assert b"must be > memory depth" in lib.dpd_last_error()
    assert lib.dpd_tap_assembly(null, 1, 16, p, null) == 1      # DPD_ERR_INVALID_ARG
    lib.dpd_pa_forward.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.dpd_pa_forward(p, 1, 4, p, p, 4, 5, p, null) == 2
    assert lib.dpd_pa_forward(p, 1, 40, p, p, 9, 5, p, null) == 1
    lib.dpd_spectral_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]
    assert lib.dpd_spectral_stats(p, p, 1, 9000, p, p, null) == 1  # neither a power of two nor within the direct-DFT limit
    i16 = (ctypes.c_int16 * 1024)()
    q = ctypes.cast(i16, ctypes.c_void_p)
    lib.dpd_q88_int32_safe.argtypes = [ctypes.c_void_p] * 6
    assert lib.dpd_q88_int32_safe(q, q, q, q, q, q) == 1          # all-zero weights cannot overflow
    lib.dpd_q88_digits_fit.argtypes = [ctypes.c_void_p] * 3
    assert lib.dpd_q88_digits_fit(q, q, q) == 1
    lib.dpd_q88_matrix_image_bytes.restype = ctypes.c_size_t
    assert lib.dpd_q88_matrix_image_bytes() == 15 * 64 * 16
    for i in range(1024):
        i16[i] = 32767
    assert lib.dpd_q88_int32_safe(q, q, q, q, q, q) == 0
    assert lib.dpd_q88_digits_fit(q, q, q) == 0                   # 32767 has no signed 8-bit high digit


def test_host_layer_has_no_cpu_path():
    import dpd_amd  # noqa: F401
    from dpd_amd.models import Discriminator, PADigitalTwin, TDNNGeneratorQAT
    from dpd_amd.utils import SpectralLoss, quantize_to_fixed_point
    x = torch.randn(2, 37, 2)
    with pytest.raises(RuntimeError, match="MI355X"):
        TDNNGeneratorQAT()(x)
    with pytest.raises(RuntimeError, match="MI355X"):
        PADigitalTwin()(x, add_noise=False)
    with pytest.raises(RuntimeError, match="MI355X"):
        Discriminator()(x, x)
    with pytest.raises(RuntimeError, match="MI355X"):
        SpectralLoss()(x[:, :32], x[:, :32])
    with pytest.raises(RuntimeError, match="MI355X"):
        quantize_to_fixed_point(x, 256.0, 16)
    # shape errors surface as ValueError like the reference, before any device work
    with pytest.raises(ValueError):
        TDNNGeneratorQAT().memory_assembly(torch.zeros(1, 5, 2))
    with pytest.raises(ValueError):
        PADigitalTwin().set_temperature_state("warm")


def test_state_dict_keys_match_reference_fixtures():
    from conftest import load_golden
    import dpd_amd  # noqa: F401
    from dpd_amd.models import Discriminator, TDNNGenerator, TDNNGeneratorQAT
    g3, g2, g6 = load_golden("g3_qat_gen"), load_golden("g2_float_gen"), load_golden("g6_critic")
    keys = lambda g, p: [k[len(p):].replace("__", ".") for k in g.files if k.startswith(p)]
    assert list(TDNNGeneratorQAT().state_dict().keys()) == keys(g3, "sd0__")
    assert list(TDNNGenerator().state_dict().keys()) == keys(g2, "sd__")
    assert list(Discriminator().state_dict().keys()) == keys(g6, "sd0__")
    assert list(Discriminator(use_spectral_norm=False).state_dict().keys()) == keys(g6, "nosn__")
    assert TDNNGeneratorQAT().get_param_count()["total"] == 1554
    assert sum(p.numel() for p in Discriminator().parameters()) == 2945
