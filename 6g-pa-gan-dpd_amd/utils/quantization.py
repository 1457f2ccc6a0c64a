"""Host-side mirror of the reference's utils/quantization.py (QuantizationConfig :36-85,
quantize_to_fixed_point :88-120, dequantize_from_fixed_point :123-137, fake_quantize :140-161,
FakeQuantizeFunction :164-173, QuantizedLinear :186-256, export/load_weights_binary :259-362,
generate_tanh_lut :365-393, generate_leaky_relu_params :396-430) plus the Q8.8 integer inference
entry that the reference only describes (README Q-format table).  Element-wise quantisation runs in
csrc/quant.hip, integer inference in csrc/q88.hip."""
import struct
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L


@dataclass
class QuantizationConfig:
    weight_bits: int = 16
    weight_frac_bits: int = 15
    weight_signed: bool = True
    activation_bits: int = 16
    activation_frac_bits: int = 8
    activation_signed: bool = True
    accumulator_bits: int = 32
    accumulator_frac_bits: int = 16
    io_bits: int = 16
    io_frac_bits: int = 15

    @property
    def weight_scale(self) -> float:
        return 2 ** self.weight_frac_bits

    @property
    def activation_scale(self) -> float:
        return 2 ** self.activation_frac_bits

    @property
    def io_scale(self) -> float:
        return 2 ** self.io_frac_bits

    @property
    def weight_range(self) -> Tuple[int, int]:
        if self.weight_signed:
            return (-(2 ** (self.weight_bits - 1)), 2 ** (self.weight_bits - 1) - 1)
        return (0, 2 ** self.weight_bits - 1)

    @property
    def activation_range(self) -> Tuple[int, int]:
        if self.activation_signed:
            return (-(2 ** (self.activation_bits - 1)), 2 ** (self.activation_bits - 1) - 1)
        return (0, 2 ** self.activation_bits - 1)


def _quantize(tensor: torch.Tensor, scale: float, bits: int, signed: bool, dequant: bool) -> torch.Tensor:
    L.need_gpu(tensor)
    x = L.as_f32(tensor.detach())
    out = torch.empty_like(x)
    L.call("dpd_quantize_fixed", L.fptr(x), x.numel(), float(scale), int(bits), int(signed), int(dequant), L.fptr(out),
           L.stream())
    return out


def quantize_to_fixed_point(tensor: torch.Tensor, scale: float, bits: int, signed: bool = True) -> torch.Tensor:
    """clamp(round_half_even(x * scale), qmin, qmax) as an integral-valued fp32 tensor."""
    return _quantize(tensor, scale, bits, signed, False)


def dequantize_from_fixed_point(tensor: torch.Tensor, scale: float) -> torch.Tensor:
    return tensor.float() / scale


def fake_quantize(tensor: torch.Tensor, scale: float, bits: int, signed: bool = True) -> torch.Tensor:
    return _quantize(tensor, scale, bits, signed, True)


class FakeQuantizeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, scale: float, bits: int, signed: bool) -> torch.Tensor:
        return fake_quantize(x, scale, bits, signed)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None, None, None


def fake_quantize_grad(tensor: torch.Tensor, scale: float, bits: int, signed: bool = True) -> torch.Tensor:
    return FakeQuantizeFunction.apply(tensor, scale, bits, signed)


class _QuantizedLinearFn(torch.autograd.Function):
    """fq_a(fq_a(x) fq_w(W)^T + b) in one kernel with an exactly formed sum; straight-through backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, cfg):
        L.need_gpu(x, weight)
        xc, wc = L.as_f32(x), L.as_f32(weight)
        bc = L.as_f32(bias) if bias is not None else None
        out_f, in_f = wc.shape
        x2 = xc.reshape(-1, in_f)
        out = torch.empty(x2.shape[0], out_f, dtype=torch.float32, device=xc.device)
        ctx.q = (float(cfg.activation_scale), int(cfg.activation_bits), int(cfg.activation_signed),
                 float(cfg.weight_scale), int(cfg.weight_bits), int(cfg.weight_signed))
        L.call("dpd_quantized_linear", L.fptr(x2), L.fptr(wc), L.fptr(bc), x2.shape[0], in_f, out_f, *ctx.q, L.fptr(out),
               L.stream())
        ctx.save_for_backward(x2, wc)
        ctx.has_bias = bias is not None
        return out.reshape(*xc.shape[:-1], out_f)

    @staticmethod
    def backward(ctx, gy):
        x2, wc = ctx.saved_tensors
        out_f, in_f = wc.shape
        g2 = L.as_f32(gy).reshape(-1, out_f)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        gx = torch.empty_like(x2) if need_x else None
        gw = torch.empty_like(wc) if (need_w or need_b) else None
        gb = torch.empty(out_f, dtype=torch.float32, device=wc.device) if need_b else None
        L.call("dpd_quantized_linear_backward", L.fptr(x2), L.fptr(wc), L.fptr(g2), x2.shape[0], in_f, out_f, *ctx.q,
               L.fptr(gx), L.fptr(gw), L.fptr(gb), L.stream())
        return (gx.reshape(*gy.shape[:-1], in_f) if need_x else None), (gw if need_w else None), gb, None


class QuantizedLinear(nn.Module):
    """fq(x, 2^8) @ fq(W, 2^15)^T + bias, then fq(., 2^8): the fixed-scale datapath statement."""

    def __init__(self, in_features: int, out_features: int, config: QuantizationConfig, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features, self.config = in_features, out_features, config
        self.weight = nn.Parameter(torch.randn(out_features, in_features) * 0.1)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features))
        else:
            self.register_parameter("bias", None)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _QuantizedLinearFn.apply(x, self.weight, self.bias, self.config)

    def get_quantized_weights(self) -> Dict[str, torch.Tensor]:
        c = self.config
        result = {"weight": quantize_to_fixed_point(self.weight.data, c.weight_scale, c.weight_bits, c.weight_signed).to(torch.int16)}
        if self.bias is not None:
            result["bias"] = quantize_to_fixed_point(self.bias.data, c.activation_scale, c.activation_bits,
                                                     c.activation_signed).to(torch.int16)
        return result


def export_weights_binary(weights: Dict[str, torch.Tensor], output_path: Union[str, Path], endianness: str = "little") -> None:
    """'DPD1' container: u32 magic 0x44504431, u32 total size, then per tensor a 32-byte NUL-padded
    name, u32 rows, u32 cols (1 for vectors) and the int16 data."""
    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    bo = "<" if endianness == "little" else ">"
    body = b""
    for name, tensor in weights.items():
        t = tensor.detach().cpu()
        shape = (t.shape[0], 1) if t.dim() == 1 else (t.shape[0], t.shape[1])
        body += name.encode("utf-8")[:31].ljust(32, b"\x00") + struct.pack(bo + "II", *shape)
        body += t.flatten().numpy().astype(np.int16).tobytes()
    with open(output_path, "wb") as f:
        f.write(struct.pack(bo + "I", 0x44504431) + struct.pack(bo + "I", 8 + len(body)) + body)


def load_weights_binary(input_path: Union[str, Path], endianness: str = "little") -> Dict[str, torch.Tensor]:
    bo = "<" if endianness == "little" else ">"
    blob = Path(input_path).read_bytes()
    magic, total = struct.unpack_from(bo + "II", blob, 0)
    if magic != 0x44504431:
        raise ValueError(f"Invalid magic number: {hex(magic)}")
    weights, pos = {}, 8
    while pos < total:
        name = blob[pos:pos + 32].rstrip(b"\x00").decode("utf-8")
        rows, cols = struct.unpack_from(bo + "II", blob, pos + 32)
        pos += 40
        data = np.frombuffer(blob, dtype=np.int16, count=rows * cols, offset=pos).copy()
        pos += rows * cols * 2
        weights[name] = torch.from_numpy(data).reshape(rows) if cols == 1 else torch.from_numpy(data).reshape(rows, cols)
    return weights


def generate_tanh_lut(num_entries: int = 256, input_range: float = 4.0, output_bits: int = 16,
                      output_frac_bits: int = 15) -> torch.Tensor:
    """Host-side table: round(tanh(linspace(-r, r, n)) * 2^frac), clamped, int16 (CPU tensor)."""
    x = torch.linspace(-input_range, input_range, num_entries)
    lo, hi = -(2 ** (output_bits - 1)), 2 ** (output_bits - 1) - 1
    return torch.clamp(torch.round(torch.tanh(x) * (2 ** output_frac_bits)), lo, hi).to(torch.int16)


def generate_leaky_relu_params(negative_slope: float = 0.2, bits: int = 16) -> Dict[str, float]:
    best_shift, best_error = 2, float("inf")
    for shift in range(1, 8):
        error = abs(1.0 / (2 ** shift) - negative_slope)
        if error < best_error:
            best_error, best_shift = error, shift
    return {"shift_amount": best_shift, "actual_slope": 1.0 / (2 ** best_shift), "target_slope": negative_slope,
            "error": best_error}


# ------------------------------------------------------------------------------------------
# Q8.8 integer inference (no end-to-end counterpart in the reference; contract in csrc/q88.hip)
# ------------------------------------------------------------------------------------------
def export_q88_weights(generator) -> Dict[str, torch.Tensor]:
    """int16 weights Q1.15 and biases Q8.8 of a generator, plus the tanh table (all on the generator's device)."""
    dev = generator.fc1.weight.device
    out = {}
    for i, fc in ((1, generator.fc1), (2, generator.fc2), (3, generator.fc3)):
        out[f"w{i}"] = quantize_to_fixed_point(fc.weight.data, 32768.0, 16).to(torch.int16)
        out[f"b{i}"] = quantize_to_fixed_point(fc.bias.data, 256.0, 16).to(torch.int16)
    out["lut"] = generate_tanh_lut().to(dev)
    return out


def to_q8_8(values: torch.Tensor) -> torch.Tensor:
    """clip(round_half_even(v * 256), -32768, 32767) as int16: the notebook's `to_q8_8` (training_colab.ipynb,
    weight-export cell), which agrees with quantize_to_fixed_point(v, 256, 16)."""
    return quantize_to_fixed_point(values, 256.0, 16).to(torch.int16)


Q88_BANK_OFFSETS = {"fc1_weights": 0, "fc1_bias": 960, "fc2_weights": 992, "fc2_bias": 1504, "fc3_weights": 1520, "fc3_bias": 1552}
Q88_BANK_SIZE = 1554


def export_q88_flat(generator) -> torch.Tensor:
    """One weight bank of the FPGA inference engine as int16 [1554] (host tensor): all six parameter tensors in Q8.8,
    each flattened row-major ([out, in]), in the order and at the offsets of rtl/src/tdnn_generator.v:70-79
    (FC1 weights 0, b1 960, FC2 weights 992, b2 1504, FC3 weights 1520, b3 1552)."""
    parts = []
    for fc in (generator.fc1, generator.fc2, generator.fc3):
        parts += [to_q8_8(fc.weight.data).reshape(-1), to_q8_8(fc.bias.data).reshape(-1)]
    flat = torch.cat(parts).cpu()
    if flat.numel() != Q88_BANK_SIZE:
        raise ValueError(f"the bank layout is that of the 30 -> 32 -> 16 -> 2 generator ({Q88_BANK_SIZE} values), got {flat.numel()}")
    return flat


def _hex_lines(q: torch.Tensor) -> bytes:
    return "".join(f"{int(v) & 0xFFFF:04x}\n" for v in q.reshape(-1).tolist()).encode("ascii")


def export_q88_hex(generator, out_dir: Union[str, Path]) -> Dict[str, Path]:
    """The notebook's hex export (one 16-bit two's-complement word per line, four lower-case hex digits):
    fc{1,2,3}_weights.hex and fc{1,2,3}_bias.hex as the notebook writes them, plus weights_q88.hex, the whole bank
    of `export_q88_flat` in one file for $readmemh."""
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    flat = export_q88_flat(generator)
    ends = sorted(Q88_BANK_OFFSETS.values()) + [Q88_BANK_SIZE]
    written = {}
    for name, off in Q88_BANK_OFFSETS.items():
        path = out_dir / f"{name}.hex"
        path.write_bytes(_hex_lines(flat[off:ends[ends.index(off) + 1]]))
        written[name] = path
    path = out_dir / "weights_q88.hex"
    path.write_bytes(_hex_lines(flat))
    written["bank"] = path
    return written


class Q88Inference:
    """Bit-exact fixed-point DPD: iq int16 Q1.15 [B, S, 2] -> int16 Q1.15 [B, S-5, 2]."""

    MODE_VALU32, MODE_VALU64, MODE_MATRIX = 0, 1, 2

    def __init__(self, weights: Dict[str, torch.Tensor], force_acc64: bool = False, force_valu: bool = False):
        """Three kernels give the same bits; the fastest one the weights allow is picked: the int8 matrix cores (every
        int16 as two 8-bit digits) when 32-bit accumulation is provably exact and every weight is <= 32639, the integer
        VALU with 32-bit accumulators when only the proof holds, with 64-bit accumulators otherwise.  `force_acc64` /
        `force_valu` select the slower kernels (tests)."""
        self.w = {k: v.contiguous() for k, v in weights.items()}
        host = [self.w[k].cpu().contiguous() for k in ("w1", "b1", "w2", "b2", "w3", "b3")]
        safe = L.lib().dpd_q88_int32_safe(*(L.c_p(t.data_ptr()) for t in host))
        fit = L.lib().dpd_q88_digits_fit(*(L.c_p(host[i].data_ptr()) for i in (0, 2, 4)))
        self.acc64 = bool(force_acc64 or not safe)
        self.mode = self.MODE_VALU64 if self.acc64 else (self.MODE_MATRIX if (fit and not force_valu) else self.MODE_VALU32)
        self.image = None      # weight image of the matrix-core kernel, built at the first call

    def _prepare_image(self):
        i16 = torch.int16
        self.image = torch.empty(int(L.lib().dpd_q88_matrix_image_bytes()), dtype=torch.uint8, device=self.w["w1"].device)
        L.call("dpd_q88_matrix_prepare", *(L.fptr(self.w[k], i16) for k in ("w1", "b1", "w2", "b2", "w3", "b3")),
               L.c_p(self.image.data_ptr()), L.stream())

    def __call__(self, iq: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`out` (optional, int16 [B, S-5, 2], contiguous) receives the result in place of a new tensor: a
        streaming caller writes chunk after chunk into one resident buffer without a copy."""
        L.need_gpu(iq)
        if iq.dtype != torch.int16:
            raise RuntimeError("Q88Inference expects int16 Q1.15 IQ")
        x = iq.contiguous()
        B, S = x.shape[0], x.shape[1]
        if S <= 5:
            raise ValueError(f"Sequence length {S} must be > memory depth 5")
        if out is None:
            out = torch.empty(B, S - 5, 2, dtype=torch.int16, device=x.device)
        elif tuple(out.shape) != (B, S - 5, 2) or out.dtype != torch.int16 or not out.is_contiguous():
            raise RuntimeError(f"out must be a contiguous int16 tensor of shape {(B, S - 5, 2)}")
        i16 = torch.int16
        if self.mode == self.MODE_MATRIX:
            if self.image is None:
                self._prepare_image()
            L.call("dpd_q88_matrix_infer", L.fptr(x, i16), B, S, L.c_p(self.image.data_ptr()), L.fptr(self.w["lut"], i16),
                   L.fptr(out, i16), L.stream())
            return out
        L.call("dpd_q88_infer", L.fptr(x, i16), B, S, *(L.fptr(self.w[k], i16) for k in ("w1", "b1", "w2", "b2", "w3", "b3")),
               L.fptr(self.w["lut"], i16), int(self.acc64), L.fptr(out, i16), L.stream())
        return out
