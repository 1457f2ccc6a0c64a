"""Host-side mirror of the reference's models/pa_digital_twin.py (VolterraPA :36-170, PADigitalTwin
:173-378, create_pa_digital_twin :381-394) on top of libdpd_hip.so.  Coefficient tables, thermal
states and error behaviour follow the reference; the memory polynomial and its gradient run in
the HIP kernels of csrc/pa.hip."""
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L


class _PAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, coef_re, coef_im):
        L.need_gpu(x, coef_re)
        xc = L.as_f32(x)
        B, S, _ = xc.shape
        K, M = coef_re.shape
        cre, cim = L.as_f32(coef_re), L.as_f32(coef_im)
        y = torch.empty(B, max(S - M + 1, 0), 2, dtype=torch.float32, device=xc.device)
        L.call("dpd_pa_forward", L.fptr(xc), B, S, L.fptr(cre), L.fptr(cim), K, M, L.fptr(y), L.stream())
        ctx.save_for_backward(xc, cre, cim)
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, cre, cim = ctx.saved_tensors
        B, S, _ = xc.shape
        K, M = cre.shape
        gx = torch.empty_like(xc)
        L.call("dpd_pa_backward", L.fptr(xc), B, S, L.fptr(cre), L.fptr(cim), K, M, L.fptr(L.as_f32(gy)), L.fptr(gx),
               L.stream())
        return gx, None, None


class VolterraPA(nn.Module):
    """Memory polynomial y(n) = sum_k sum_m h[k,m] x(n-m) |x(n-m)|^(k-1), k odd, m < M."""

    def __init__(self, memory_depth: int = 5, nonlinear_order: int = 7, gain_db: float = 30.0, p1db_dbm: float = 40.0):
        super().__init__()
        self.memory_depth, self.nonlinear_order, self.gain_db, self.p1db_dbm = memory_depth, nonlinear_order, gain_db, p1db_dbm
        self.orders = list(range(1, nonlinear_order + 1, 2))
        self.num_orders = len(self.orders)
        self.num_coeffs = self.num_orders * memory_depth
        self.register_buffer("coeffs_real", torch.zeros(self.num_orders, memory_depth))
        self.register_buffer("coeffs_imag", torch.zeros(self.num_orders, memory_depth))
        self._init_coefficients()

    def _init_coefficients(self):
        g = 10 ** (self.gain_db / 20)
        re, im = self.coeffs_real, self.coeffs_imag
        re[0, 0] = g
        im[0, 0] = 0.0
        for m in range(1, self.memory_depth):
            d = 0.1 * np.exp(-m / 2)
            re[0, m] = d * g
            im[0, m] = 0.02 * d * g
        for ki, k in enumerate(self.orders[1:], start=1):
            am_am, am_pm = -0.05 * g / (k ** 1.5), 0.02 * g / (k ** 1.5)
            re[ki, 0], im[ki, 0] = am_am, am_pm
            for m in range(1, self.memory_depth):
                d = 0.3 * np.exp(-m / 1.5)
                re[ki, m] = am_am * d
                im[ki, m] = am_pm * d * 0.5

    def get_coefficients(self) -> torch.Tensor:
        return torch.complex(self.coeffs_real, self.coeffs_imag)

    def set_coefficients(self, coeffs: torch.Tensor):
        self.coeffs_real.copy_(coeffs.real)
        self.coeffs_imag.copy_(coeffs.imag)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not (x.dim() == 3 and x.shape[-1] == 2):
            x = torch.view_as_real(x.to(torch.complex64))       # complex [B, S] input form
        seq_len = x.shape[1]
        if seq_len < self.memory_depth:
            raise ValueError(f"Sequence length {seq_len} must be >= memory depth {self.memory_depth}")
        if self.num_orders > 8 or self.memory_depth > 16:
            raise NotImplementedError("the HIP kernel supports up to 8 odd orders and 16 taps")
        return _PAFn.apply(x, self.coeffs_real, self.coeffs_imag)


class PADigitalTwin(nn.Module):
    """Volterra PA with temperature-dependent coefficient sets and optional AWGN / phase noise."""

    def __init__(self, memory_depth: int = 5, nonlinear_order: int = 7, gain_db: float = 30.0, p1db_dbm: float = 40.0,
                 noise_floor_db: float = -60.0, phase_noise_deg: float = 0.5):
        super().__init__()
        self.memory_depth, self.noise_floor_db, self.phase_noise_deg = memory_depth, noise_floor_db, phase_noise_deg
        self.pa_model = VolterraPA(memory_depth, nonlinear_order, gain_db, p1db_dbm)
        self.temp_states = ["cold", "normal", "hot"]
        self.current_temp_state = "normal"
        self.temp_coefficients = {
            "cold": dict(gain_scale=1.02, phase_offset_deg=-5, am_am_scale=1.05, am_pm_scale=0.95),
            "normal": dict(gain_scale=1.0, phase_offset_deg=0, am_am_scale=1.0, am_pm_scale=1.0),
            "hot": dict(gain_scale=0.95, phase_offset_deg=10, am_am_scale=0.92, am_pm_scale=1.15),
        }
        self.register_buffer("base_coeffs_real", self.pa_model.coeffs_real.clone())
        self.register_buffer("base_coeffs_imag", self.pa_model.coeffs_imag.clone())

    def set_temperature_state(self, state: str):
        if state not in self.temp_states:
            raise ValueError(f"Invalid temperature state: {state}")
        self.current_temp_state = state
        t = self.temp_coefficients[state]
        ph = t["phase_offset_deg"] * np.pi / 180
        re, im = self.base_coeffs_real.clone(), self.base_coeffs_imag.clone()
        # the linear row is rebuilt from the base REAL part only (reference :269-270)
        re[0] = self.base_coeffs_real[0] * t["gain_scale"] * np.cos(ph)
        im[0] = self.base_coeffs_real[0] * t["gain_scale"] * np.sin(ph)
        for k in range(1, self.pa_model.num_orders):
            re[k] = self.base_coeffs_real[k] * t["am_am_scale"]
            im[k] = self.base_coeffs_imag[k] * t["am_pm_scale"]
        self.pa_model.coeffs_real.copy_(re)
        self.pa_model.coeffs_imag.copy_(im)

    def add_noise(self, x: torch.Tensor, draws: Optional[dict] = None, seed: Optional[int] = None) -> torch.Tensor:
        """AWGN at noise_floor_db relative to the power of the whole tensor plus per-sample phase noise
        (models/pa_digital_twin.py:280-309), one kernel pass.  `draws` = dict(awgn [B, S, 2], phase [B, S]) of standard
        normals fixes the random numbers (the reference takes them from torch.randn_like / torch.randn); otherwise they
        come from the kernel's counter-based generator, keyed by `seed` (default: a fresh seed from torch's generator).
        Evaluation-time impairment: no gradient flows through it (the reference's scripts use it for validation only)."""
        L.need_gpu(x)
        xc = L.as_f32(x)
        n = xc.numel() // 2
        params = torch.tensor([[10.0 ** (self.noise_floor_db / 10.0), 1.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0]],
                              dtype=torch.float32, device=xc.device)
        pn_rad = float(self.phase_noise_deg * np.pi / 180) if self.phase_noise_deg > 0 else 0.0
        awgn = pn = None
        if draws is not None:
            awgn, pn = L.as_f32(draws["awgn"]).reshape(-1), L.as_f32(draws["phase"]).reshape(-1)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        nbytes = int(L.lib().dpd_impair_workspace_bytes(1, n))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xc.device)
        out = torch.empty_like(xc)
        L.call("dpd_pa_add_noise", L.fptr(xc), 1, n, L.fptr(params), pn_rad, L.fptr(awgn), L.fptr(pn),
               int(seed) & 0xFFFFFFFFFFFFFFFF, L.fptr(out), L.c_p(ws.data_ptr()), L.c_sz(ws.numel()), L.stream())
        return out

    def forward(self, x: torch.Tensor, add_noise: bool = True, temp_state: Optional[str] = None) -> torch.Tensor:
        if temp_state is not None:
            self.set_temperature_state(temp_state)
        y = self.pa_model(x)
        return self.add_noise(y) if add_noise else y

    def get_transfer_function(self, num_points: int = 256, max_amplitude: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """AM-AM / AM-PM of a constant-envelope input, all amplitudes in one batched call."""
        dev = self.pa_model.coeffs_real.device
        amp = torch.linspace(0.01, max_amplitude, num_points)
        x = torch.zeros(num_points, self.memory_depth + 1, 2, device=dev)
        x[:, :, 0] = amp.to(dev).unsqueeze(1)
        with torch.no_grad():
            y = self.pa_model(x)[:, -1, :].cpu()
        yc = torch.complex(y[:, 0], y[:, 1])
        return amp, yc.abs(), torch.angle(yc) * 180 / np.pi


def create_pa_digital_twin(config: dict) -> PADigitalTwin:
    pa = config["pa"]
    return PADigitalTwin(memory_depth=pa.get("memory_depth", 5), nonlinear_order=pa.get("nonlinear_order", 7),
                         gain_db=pa.get("gain_db", 30.0), p1db_dbm=pa.get("p1db_dbm", 40.0),
                         noise_floor_db=pa.get("noise", {}).get("floor_db", -60),
                         phase_noise_deg=pa.get("noise", {}).get("phase_noise_deg", 0.5))
