"""ctypes binding of libdpd_hip.so (the C ABI in include/dpd_hip.h)."""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPD_HIP_LIBRARY") or os.path.join(_HERE, "libdpd_hip.so")   # override: another build of the same ABI

GEN_NPARAM = 1554
GEN_NSCALE = 8
CRITIC_NPARAM = 2945
IMPL_AUTO, IMPL_VALU, IMPL_MFMA = 0, 1, 2
ERR_SHAPE = 2

_lib = None
_lock = threading.Lock()

c_p, c_i, c_l, c_f, c_sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t

# name -> argument types (all return int unless listed in _RESTYPES)
PROTOTYPES = {
    "dpd_tap_assembly": [c_p, c_l, c_l, c_p, c_p],
    "dpd_tap_assembly_backward": [c_p, c_l, c_l, c_p, c_p, c_p],
    "dpd_gen_prepare_weights": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p],
    "dpd_gen_absmax_pass": [c_i, c_p, c_l, c_l, c_i, c_p, c_f, c_p, c_p, c_i, c_p],
    "dpd_gen_uses_mfma": [c_l, c_l, c_i, c_i],
    "dpd_gen_quantize_output": [c_p, c_l, c_p, c_p],
    "dpd_gen_forward": [c_p, c_l, c_l, c_i, c_p, c_f, c_i, c_p, c_p, c_i, c_p],
    "dpd_gen_backward": [c_p, c_l, c_l, c_i, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_i, c_p],
    "dpd_gen_mse_step": [c_p, c_l, c_l, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_sz, c_i, c_p],
    "dpd_gen_mse_step_noalloc": [c_p, c_l, c_l, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_i, c_p],
    "dpd_pa_forward": [c_p, c_l, c_l, c_p, c_p, c_i, c_i, c_p, c_p],
    "dpd_pa_backward": [c_p, c_l, c_l, c_p, c_p, c_i, c_i, c_p, c_p, c_p],
    "dpd_quant_pa_forward": [c_p, c_l, c_l, c_p, c_p, c_p, c_i, c_i, c_p, c_p],
    "dpd_gen_pa_forward": [c_p, c_l, c_l, c_p, c_f, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p],
    "dpd_gen_pa_backward": [c_p, c_l, c_l, c_p, c_f, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_sz, c_p],
    "dpd_gen_pa_mse_step": [c_p, c_l, c_l, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_sz, c_p],
    "dpd_impair_workspace_bytes": [c_l, c_l],
    "dpd_augment_rows": [c_p, c_l, c_l, c_p, c_p, ctypes.c_uint64, c_p, c_p, c_sz, c_p],
    "dpd_pa_add_noise": [c_p, c_l, c_l, c_p, c_f, c_p, c_p, ctypes.c_uint64, c_p, c_p, c_sz, c_p],
    "dpd_critic_sn_forward": [c_p, c_p, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_p],
    "dpd_critic_sn_backward": [c_p, c_p, c_p, c_p, c_p, c_p],
    "dpd_critic_sn_forward3": [c_p, c_p, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_p],
    "dpd_critic_sn_backward3": [c_p, c_p, c_p, c_p, c_f, c_p, c_p],
    "dpd_qat_observe": [c_p, c_p, c_i, c_f, c_i, c_i, c_p],
    "dpd_critic_forward": [c_p, c_p, c_l, c_p, c_f, c_p, c_p, c_p],
    "dpd_critic_backward": [c_p, c_p, c_l, c_p, c_f, c_p, c_f, c_p, c_p, c_p, c_p, c_sz, c_p],
    "dpd_critic_gp": [c_p, c_p, c_p, c_p, c_l, c_l, c_p, c_f, c_p, c_p, c_p, c_p, c_sz, c_p],
    "dpd_spectral_stats": [c_p, c_p, c_l, c_l, c_p, c_p, c_p],
    "dpd_spectral_backward": [c_p, c_p, c_l, c_l, c_p, c_p, c_p, c_p],
    "dpd_spectral_rowdb": [c_p, c_l, c_l, c_p, c_p, c_p],
    "dpd_spectral_loss_final": [c_p, c_l, c_f, c_f, c_f, c_p, c_p],
    "dpd_spectral_terms_backward": [c_p, c_p, c_p, c_l, c_l, c_p, c_p],
    "dpd_quantize_fixed": [c_p, c_l, c_f, c_i, c_i, c_i, c_p, c_p],
    "dpd_quantized_linear": [c_p, c_p, c_p, c_l, c_i, c_i, c_f, c_i, c_i, c_f, c_i, c_i, c_p, c_p],
    "dpd_quantized_linear_backward": [c_p, c_p, c_p, c_l, c_i, c_i, c_f, c_i, c_i, c_f, c_i, c_i, c_p, c_p, c_p, c_p],
    "dpd_fake_quantize_dynamic": [c_p, c_l, c_l, c_i, c_i, c_p, c_p, c_p],
    "dpd_q88_infer": [c_p, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p],
    "dpd_q88_int32_safe": [c_p, c_p, c_p, c_p, c_p, c_p],
    "dpd_q88_digits_fit": [c_p, c_p, c_p],
    "dpd_q88_matrix_image_bytes": [],
    "dpd_q88_matrix_prepare": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "dpd_q88_matrix_infer": [c_p, c_l, c_l, c_p, c_p, c_p, c_p],
    "dpd_workspace_bytes": [c_l],
    "dpd_set_option": [ctypes.c_char_p, c_i],
    "dpd_last_error": [],
    "dpd_version": [],
}
_RESTYPES = {"dpd_workspace_bytes": c_sz, "dpd_impair_workspace_bytes": c_sz, "dpd_q88_matrix_image_bytes": c_sz, "dpd_last_error": ctypes.c_char_p}


def lib():
    """The loaded library.  Raises if it has not been built: the product has no CPU fallback."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc, gfx950). dpd_amd has no CPU or eager fallback.")
                handle = ctypes.CDLL(LIB_PATH)
                for name, args in PROTOTYPES.items():
                    if not hasattr(handle, name):   # reported when the symbol is first called
                        continue
                    fn = getattr(handle, name)
                    fn.argtypes = args
                    fn.restype = _RESTYPES.get(name, c_i)
                _lib = handle
    return _lib


def check(rc):
    if rc == 0:
        return
    msg = lib().dpd_last_error().decode("utf-8", "replace")
    if rc == ERR_SHAPE:
        raise ValueError(msg)
    raise RuntimeError(f"libdpd_hip error {rc}: {msg}")


def call(name, *args):
    handle = lib()
    if not hasattr(handle, name):
        raise RuntimeError(f"libdpd_hip.so does not export {name}: rebuild the library")
    check(getattr(handle, name)(*args))


def need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dpd_amd ops run on the MI355X only: got a tensor on %s (no CPU path)" % t.device)


def fptr(t, dtype=torch.float32):
    """Device pointer of a contiguous tensor of the expected dtype (None -> NULL)."""
    if t is None:
        return c_p(0)
    if t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise RuntimeError(f"expected a contiguous {dtype} GPU tensor, got {t.dtype} contiguous={t.is_contiguous()} on {t.device}")
    return c_p(t.data_ptr())


def as_f32(t):
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


def stream():
    return c_p(torch.cuda.current_stream().cuda_stream)


_workspaces = {}


def workspace(device, n_samples=0):
    """One scratch buffer per device, grown on demand and reused by every call on that device (calls
    on one device and different streams must not overlap)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    nbytes = int(lib().dpd_workspace_bytes(int(n_samples)))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", key))
        _workspaces[key] = ws
    return ws, c_p(ws.data_ptr()), c_sz(ws.numel())
