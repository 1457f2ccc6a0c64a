"""Fixed-scale quantisation primitives (oracle; test infrastructure only).

Follows reference utils/quantization.py:
  * quantize_to_fixed_point      :88-120   clamp(round(x * scale), qmin, qmax), round-half-even
  * dequantize_from_fixed_point  :123-137
  * fake_quantize                :140-161
  * QuantizedLinear              :213-256  fq(x, 2^8) . fq(W, 2^15)^T + bias -> fq(., 2^8);
                                           export: W int16 Q1.15, bias int16 Q8.8
  * weight container 'DPD1'      :259-362
  * generate_tanh_lut            :365-393  round(tanh(linspace(-4, 4, n)) * 2^15), clamped
  * generate_leaky_relu_params   :396-430
"""
import struct
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def qrange(bits: int, signed: bool = True):
    return (-(2 ** (bits - 1)), 2 ** (bits - 1) - 1) if signed else (0, 2 ** bits - 1)


def quantize_fixed(x: torch.Tensor, scale: float, bits: int, signed: bool = True) -> torch.Tensor:
    lo, hi = qrange(bits, signed)
    return torch.clamp(torch.round(x * scale), lo, hi)


def dequantize_fixed(q: torch.Tensor, scale: float) -> torch.Tensor:
    return q.float() / scale


def fake_quantize_fixed(x: torch.Tensor, scale: float, bits: int, signed: bool = True) -> torch.Tensor:
    return dequantize_fixed(quantize_fixed(x, scale, bits, signed), scale)


class _FQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, bits):
        return fake_quantize_fixed(x, scale, bits)

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def quantized_linear(x, weight, bias, act_scale=256.0, w_scale=32768.0, bits=16):
    out = F.linear(_FQ.apply(x, act_scale, bits), _FQ.apply(weight, w_scale, bits), bias)
    return _FQ.apply(out, act_scale, bits)


def tanh_lut(num_entries: int = 256, input_range: float = 4.0, bits: int = 16, frac_bits: int = 15) -> torch.Tensor:
    x = torch.linspace(-input_range, input_range, num_entries)
    return quantize_fixed(torch.tanh(x), 2 ** frac_bits, bits).to(torch.int16)


def leaky_relu_params(negative_slope: float = 0.2) -> Dict[str, float]:
    best_shift, best_err = 2, float("inf")
    for s in range(1, 8):
        e = abs(1.0 / 2 ** s - negative_slope)
        if e < best_err:
            best_shift, best_err = s, e
    return dict(shift_amount=best_shift, actual_slope=1.0 / 2 ** best_shift, target_slope=negative_slope, error=best_err)


def pack_weights(weights: Dict[str, np.ndarray], endianness: str = "little") -> bytes:
    """'DPD1' container: u32 magic 0x44504431, u32 total size, then per tensor a 32-byte
    NUL-padded name, u32 rows, u32 cols (1 for vectors), int16 data."""
    bo = "<" if endianness == "little" else ">"
    body = b""
    for name, t in weights.items():
        a = np.asarray(t)
        shape = (a.shape[0], 1) if a.ndim == 1 else (a.shape[0], a.shape[1])
        body += name.encode("utf-8")[:31].ljust(32, b"\x00") + struct.pack(bo + "II", *shape)
        body += a.flatten().astype(np.int16).tobytes()     # native (little-endian) data, as the reference writes it
    return struct.pack(bo + "I", 0x44504431) + struct.pack(bo + "I", 8 + len(body)) + body
