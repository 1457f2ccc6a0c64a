"""MI355X-native batched DPD hot path: hand-written gfx950 HIP kernels behind a C ABI
(libdpd_hip.so, declared in include/dpd_hip.h) and a PyTorch-ROCm host layer that mirrors the
reference's `models/` and `utils/` module API.  There is no CPU path: every op raises if the
HIP library is missing or a tensor is not on the GPU."""
from . import _lib  # noqa: F401
from . import models, utils  # noqa: F401

__all__ = ["models", "utils", "_lib"]
