"""Host-side mirror of the reference's models/tdnn_generator.py on top of libdpd_hip.so.

Same classes, constructor arguments, state-dict keys and error behaviour as the reference
(models/tdnn_generator.py: MemoryTapAssembly :139-195, fake_quantize :74-103, FakeQuantizeModule
:106-132, TDNNGenerator :202-299, TDNNGeneratorQAT :306-533, create_generator :540-567).  The
arithmetic runs in the fused gfx950 kernels: the [B, L, 30] feature tensor is never materialised
in a generator forward, and the backward recomputes the forward from the IQ rows.
"""
import math
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from .. import _lib as L

_SUPPORTED = "the HIP kernels are built for memory_depth=5, hidden_dims=[32, 16] (the reference defaults)"


# --------------------------------------------------------------------------------------------
# dynamic-range fake quantisation (stand-alone op)
# --------------------------------------------------------------------------------------------
class StraightThroughQuantize(torch.autograd.Function):
    """out = clamp(round(x / scale), qmin, qmax) * scale with scale = max(absmax, 1e-8) / qmax taken
    from x itself (whole tensor, or per row of dim 0); the gradient passes through unchanged."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, n_bits: int, per_channel: bool) -> torch.Tensor:
        L.need_gpu(x)
        xc = L.as_f32(x)
        rows = xc.shape[0] if (per_channel and xc.dim() > 0) else 1
        cols = xc.numel() // max(rows, 1)
        out = torch.empty_like(xc)
        if xc.numel():
            amax = torch.empty(max(rows, 1), dtype=torch.float32, device=xc.device)
            L.call("dpd_fake_quantize_dynamic", L.fptr(xc), rows, cols, int(per_channel), int(n_bits), L.fptr(out),
                   L.fptr(amax), L.stream())
        return out.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None, None


def fake_quantize(x: torch.Tensor, n_bits: int = 16, per_channel: bool = False, channel_dim: int = 0) -> torch.Tensor:
    if per_channel and channel_dim != 0:
        xm = x.movedim(channel_dim, 0).contiguous()
        return StraightThroughQuantize.apply(xm, n_bits, True).movedim(0, channel_dim)
    return StraightThroughQuantize.apply(x, n_bits, per_channel)


class FakeQuantizeModule(nn.Module):
    """Fake quantiser with a running-scale buffer (updated in training mode, never used for the
    quantisation itself, exactly as in the reference)."""

    def __init__(self, n_bits: int = 16, per_channel: bool = False, channel_dim: int = 0, momentum: float = 0.1):
        super().__init__()
        self.n_bits, self.per_channel, self.channel_dim, self.momentum = n_bits, per_channel, channel_dim, momentum
        self.register_buffer("running_scale", torch.tensor(1.0))
        self.register_buffer("running_zero_point", torch.tensor(0.0))

    def observe(self, absmax: torch.Tensor):
        """EMA update from a per-tensor abs-max held on the device (no host sync)."""
        q_max = 2 ** (self.n_bits - 1) - 1
        scale = torch.clamp(absmax, min=1e-8) / q_max
        self.running_scale = (1 - self.momentum) * self.running_scale + self.momentum * scale

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            self.observe(x.detach().abs().max())
        return fake_quantize(x, self.n_bits, self.per_channel, self.channel_dim)


# --------------------------------------------------------------------------------------------
# tap assembly
# --------------------------------------------------------------------------------------------
class _TapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, iq):
        L.need_gpu(iq)
        x = L.as_f32(iq)
        B, S, _ = x.shape
        feat = torch.empty(B, max(S - 5, 0), 30, dtype=torch.float32, device=x.device)
        L.call("dpd_tap_assembly", L.fptr(x), B, S, L.fptr(feat), L.stream())
        ctx.save_for_backward(x)
        return feat

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, S, _ = x.shape
        dx = torch.empty_like(x)
        L.call("dpd_tap_assembly_backward", L.fptr(x), B, S, L.fptr(L.as_f32(g)), L.fptr(dx), L.stream())
        return dx


class MemoryTapAssembly(nn.Module):
    """[B, S, 2] IQ -> [B, S-M, 30] memory-tap features (blocked order, see the kernel header)."""

    def __init__(self, memory_depth: int = 5):
        super().__init__()
        self.memory_depth = memory_depth
        self.input_dim = 2 + 3 * (memory_depth + 1) + 2 * memory_depth

    def forward(self, iq_sequence: torch.Tensor) -> torch.Tensor:
        if self.memory_depth != 5:
            raise NotImplementedError(_SUPPORTED)
        _, seq_len, _ = iq_sequence.shape
        if seq_len <= self.memory_depth:
            raise ValueError(f"Sequence length {seq_len} must be > memory depth {self.memory_depth}")
        return _TapFn.apply(iq_sequence)


# --------------------------------------------------------------------------------------------
# fused generator
# --------------------------------------------------------------------------------------------
class _GenFn(torch.autograd.Function):
    """Fused tap assembly + 30->32->16->2 generator.  Returns (out, absmax[8])."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, slope, qat, pre, impl, absmax_sync):
        L.need_gpu(x, w1)
        xc = L.as_f32(x)
        dev = xc.device
        wq = torch.empty(L.GEN_NPARAM, dtype=torch.float32, device=dev)
        absmax = torch.zeros(L.GEN_NSCALE, dtype=torch.float32, device=dev)
        st = L.stream()
        L.call("dpd_gen_prepare_weights", *(L.fptr(L.as_f32(t.detach())) for t in (w1, b1, w2, b2, w3, b3)), int(qat),
               L.fptr(wq), L.fptr(absmax), st)
        if pre:
            B, S = xc.shape[0], 0
            if xc.dim() != 2 or xc.shape[1] != 30:
                raise RuntimeError(f"pre-assembled input must be [N, 30], got {tuple(xc.shape)}")
            out = torch.empty(B, 2, dtype=torch.float32, device=dev)
        else:
            B, S = xc.shape[0], xc.shape[1]
            if S <= 5:
                raise ValueError(f"Sequence length {S} must be > memory depth 5")
            out = torch.empty(B, S - 5, 2, dtype=torch.float32, device=dev)
        if qat and absmax_sync is not None:
            # data-parallel form: each abs-max is max-reduced across ranks before the next pass
            for stage in range(4):
                L.call("dpd_gen_absmax_pass", stage, L.fptr(xc), B, S, int(pre), L.fptr(wq), float(slope),
                       L.fptr(absmax), L.fptr(out), impl, st)
                absmax_sync(absmax[stage:stage + 1])
            L.call("dpd_gen_quantize_output", L.fptr(out), out.numel(), L.fptr(absmax), st)
        else:
            L.call("dpd_gen_forward", L.fptr(xc), B, S, int(pre), L.fptr(wq), float(slope), int(qat), L.fptr(absmax),
                   L.fptr(out), impl, st)
        ctx.save_for_backward(xc, wq, absmax)
        ctx.cfg = (B, S, bool(pre), float(slope), bool(qat), impl, x.requires_grad)
        ctx.shapes = [t.shape for t in (w1, b1, w2, b2, w3, b3)]
        ctx.mark_non_differentiable(absmax)
        return out, absmax

    @staticmethod
    def backward(ctx, gout, _gabs):
        xc, wq, absmax = ctx.saved_tensors
        B, S, pre, slope, qat, impl, x_needs_grad = ctx.cfg
        grads = torch.empty(L.GEN_NPARAM, dtype=torch.float32, device=xc.device)
        _, wsp, wsn = L.workspace(xc.device)
        if x_needs_grad and not pre:
            gx = iq_rows_backward(xc, B, S, wq, slope, qat, absmax, L.as_f32(gout), grads)
        else:
            gx = torch.empty_like(xc) if (pre and x_needs_grad) else None
            L.call("dpd_gen_backward", L.fptr(xc), B, S, int(pre), L.fptr(wq), slope, int(qat), L.fptr(absmax),
                   L.fptr(L.as_f32(gout)), L.fptr(grads), L.fptr(gx), wsp, wsn, impl, L.stream())
        sizes = [960, 32, 512, 16, 32, 2]
        parts = [g.view(s) for g, s in zip(torch.split(grads, sizes), ctx.shapes)]
        return (gx, *parts, None, None, None, None, None)


def iq_rows_backward(x, B, S, wq, slope, qat, absmax, gout, grads):
    """Backward of the generator when the IQ rows themselves ask for a gradient (no caller in the reference does; its
    generator is differentiable torch, models/tdnn_generator.py:158-195, :257-287).  Three launches instead of the
    fused kernel: features to HBM, the backward on pre-assembled features (parameter gradients into `grads`, gradient of
    the features; the scales are those of the forward), the adjoint of the tap assembly.  Returns dx [B, S, 2]."""
    st = L.stream()
    feat = torch.empty(B * (S - 5), 30, dtype=torch.float32, device=x.device)
    L.call("dpd_tap_assembly", L.fptr(x), B, S, L.fptr(feat), st)
    gfeat = torch.empty_like(feat)
    _, wsp, wsn = L.workspace(x.device)
    L.call("dpd_gen_backward", L.fptr(feat), B * (S - 5), 0, 1, L.fptr(wq), slope, int(qat), L.fptr(absmax),
           L.fptr(gout), L.fptr(grads), L.fptr(gfeat), wsp, wsn, L.IMPL_AUTO, st)
    dx = torch.empty_like(x)
    L.call("dpd_tap_assembly_backward", L.fptr(x), B, S, L.fptr(gfeat), L.fptr(dx), st)
    return dx


class _GeneratorBase(nn.Module):
    def _check(self):
        if self.memory_depth != 5 or list(self.hidden_dims) != [32, 16]:
            raise NotImplementedError(_SUPPORTED)

    def _run(self, x, qat: bool, pre_assembled: bool):
        self._check()
        out, absmax = _GenFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, self.fc3.weight,
                                   self.fc3.bias, self.leaky_slope, qat, pre_assembled, self.kernel_impl,
                                   self.absmax_sync if qat else None)
        return out, absmax

    def get_param_count(self) -> Dict[str, int]:
        return {
            "fc1_weights": self.fc1.weight.numel(), "fc1_bias": self.fc1.bias.numel(),
            "fc2_weights": self.fc2.weight.numel(), "fc2_bias": self.fc2.bias.numel(),
            "fc3_weights": self.fc3.weight.numel(), "fc3_bias": self.fc3.bias.numel(),
            "total": sum(p.numel() for p in self.parameters() if p.requires_grad),
        }


class TDNNGenerator(_GeneratorBase):
    """Float generator: tanh(W3 lrelu(W2 lrelu(W1 f + b1) + b2) + b3), Xavier-uniform weights."""

    def __init__(self, memory_depth: int = 5, hidden_dims: List[int] = [32, 16], leaky_slope: float = 0.2):
        super().__init__()
        self.memory_depth = memory_depth
        self.input_dim = 2 + 3 * (memory_depth + 1) + 2 * memory_depth
        self.hidden_dims = hidden_dims
        self.leaky_slope = leaky_slope
        self.memory_assembly = MemoryTapAssembly(memory_depth)
        self.fc1 = nn.Linear(self.input_dim, hidden_dims[0])
        self.act1 = nn.LeakyReLU(negative_slope=leaky_slope)
        self.fc2 = nn.Linear(hidden_dims[0], hidden_dims[1])
        self.act2 = nn.LeakyReLU(negative_slope=leaky_slope)
        self.fc3 = nn.Linear(hidden_dims[1], 2)
        self.output_act = nn.Tanh()
        self.kernel_impl = L.IMPL_AUTO
        self.absmax_sync = None
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x: torch.Tensor, pre_assembled: bool = False) -> torch.Tensor:
        return self._run(x, False, pre_assembled)[0]


class TDNNGeneratorQAT(_GeneratorBase):
    """Generator with dynamic-range 16-bit fake quantisation of inputs, weights (per row) and
    activations; quantises in eval mode too while `qat_enabled` is set."""

    ACT_QUANT = ("input_quant", "fc1_act_quant", "fc2_act_quant", "output_quant")
    WEIGHT_QUANT = ("fc1_weight_quant", "fc2_weight_quant", "fc3_weight_quant")

    def __init__(self, memory_depth: int = 5, hidden_dims: List[int] = [32, 16], leaky_slope: float = 0.2,
                 weight_bits: int = 16, activation_bits: int = 16):
        super().__init__()
        self.memory_depth = memory_depth
        self.input_dim = 2 + 3 * (memory_depth + 1) + 2 * memory_depth
        self.hidden_dims = hidden_dims
        self.leaky_slope = leaky_slope
        self.weight_bits = weight_bits
        self.activation_bits = activation_bits
        self.memory_assembly = MemoryTapAssembly(memory_depth)
        self.input_quant = FakeQuantizeModule(n_bits=activation_bits)
        self.fc1 = nn.Linear(self.input_dim, hidden_dims[0])
        self.fc1_weight_quant = FakeQuantizeModule(n_bits=weight_bits, per_channel=True)
        self.fc1_act_quant = FakeQuantizeModule(n_bits=activation_bits)
        self.act1 = nn.LeakyReLU(negative_slope=leaky_slope)
        self.fc2 = nn.Linear(hidden_dims[0], hidden_dims[1])
        self.fc2_weight_quant = FakeQuantizeModule(n_bits=weight_bits, per_channel=True)
        self.fc2_act_quant = FakeQuantizeModule(n_bits=activation_bits)
        self.act2 = nn.LeakyReLU(negative_slope=leaky_slope)
        self.fc3 = nn.Linear(hidden_dims[1], 2)
        self.fc3_weight_quant = FakeQuantizeModule(n_bits=weight_bits, per_channel=True)
        self.fc3_act_quant = FakeQuantizeModule(n_bits=activation_bits)
        self.output_act = nn.Tanh()
        self.output_quant = FakeQuantizeModule(n_bits=16)
        self.qat_enabled = True
        self.kernel_impl = L.IMPL_AUTO
        self.absmax_sync: Optional[Callable[[torch.Tensor], None]] = None
        for m in self.modules():
            if isinstance(m, nn.Linear):
                bound = 0.5 / math.sqrt(m.in_features)
                nn.init.uniform_(m.weight, -bound, bound)
                nn.init.zeros_(m.bias)

    def enable_qat(self):
        self.qat_enabled = True

    def disable_qat(self):
        self.qat_enabled = False

    def forward(self, x: torch.Tensor, pre_assembled: bool = False) -> torch.Tensor:
        if self.qat_enabled and (self.weight_bits != 16 or self.activation_bits != 16):
            raise NotImplementedError("the HIP kernels quantise to 16 bits (the reference default)")
        out, absmax = self._run(x, self.qat_enabled, pre_assembled)
        self._last_absmax = absmax if (self.qat_enabled and self.training) else None
        self.repeat_observation()
        return out

    def repeat_observation(self, times: int = 1):
        """Advance the running-scale buffers as `times` more training-mode forwards on the same input and
        weights would (used when a caller reuses the output instead of recomputing it)."""
        absmax = getattr(self, "_last_absmax", None)
        if absmax is not None and times > 0:
            self.observe_all(absmax, times)

    def observe_all(self, absmax: torch.Tensor, times: int = 1):
        """The EMA update of every quantiser's running scale from the abs-max vector of a forward (order: features,
        act1, act2, output, W1, W2, W3), all seven in one launch (dpd_qat_observe: the arithmetic of
        FakeQuantizeModule.observe element by element, `times` times over).  The buffers become views of one
        seven-element tensor, which the kernel updates in place."""
        mods = [getattr(self, name) for name in self.ACT_QUANT + self.WEIGHT_QUANT]
        m0 = mods[0]
        if any(m.n_bits != m0.n_bits or m.momentum != m0.momentum for m in mods) or not absmax.is_cuda:
            for _ in range(times):
                for i, m in enumerate(mods):
                    m.observe(absmax[i])
            return
        rs = getattr(self, "_running_all", None)
        bound = rs is not None and all(m.running_scale.data_ptr() == rs[i].data_ptr() for i, m in enumerate(mods))
        if not bound:          # first use, or the buffers were replaced (load_state_dict copies in place and keeps them)
            rs = torch.stack([m.running_scale.detach().to(torch.float32) for m in mods]).to(absmax.device)
            self._running_all = rs
            for m, r in zip(mods, rs.unbind(0)):
                m.running_scale = r
        L.call("dpd_qat_observe", L.fptr(L.as_f32(absmax)), L.fptr(rs), len(mods), float(m0.momentum), int(m0.n_bits),
               int(times), L.stream())

    def get_quantized_weights(self) -> Dict[str, torch.Tensor]:
        weights = {}
        for i, fc in ((1, self.fc1), (2, self.fc2), (3, self.fc3)):
            w = fc.weight.data
            scale = w.abs().max() / (2 ** 15 - 1)
            weights[f"fc{i}_weight"] = torch.round(w / scale).to(torch.int16)
            weights[f"fc{i}_weight_scale"] = scale
            weights[f"fc{i}_bias"] = fc.bias.data
        return weights

    @classmethod
    def from_float(cls, float_model: TDNNGenerator, weight_bits: int = 16, activation_bits: int = 16) -> "TDNNGeneratorQAT":
        q = cls(memory_depth=float_model.memory_depth, hidden_dims=float_model.hidden_dims,
                leaky_slope=float_model.leaky_slope, weight_bits=weight_bits, activation_bits=activation_bits)
        q = q.to(float_model.fc1.weight.device)
        with torch.no_grad():
            for name in ("fc1", "fc2", "fc3"):
                getattr(q, name).weight.copy_(getattr(float_model, name).weight)
                getattr(q, name).bias.copy_(getattr(float_model, name).bias)
        return q


def create_generator(config: dict, qat: bool = False) -> nn.Module:
    gen = config["model"]["generator"]
    quant = config.get("quantization", {})
    kw = dict(memory_depth=gen.get("memory_depth", 5), hidden_dims=gen.get("hidden_dims", [32, 16]),
              leaky_slope=gen.get("leaky_slope", 0.2))
    if qat and quant.get("enabled", False):
        return TDNNGeneratorQAT(weight_bits=quant.get("weight", {}).get("bits", 16),
                                activation_bits=quant.get("activation", {}).get("bits", 16), **kw)
    return TDNNGenerator(**kw)
