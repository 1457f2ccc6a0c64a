"""Drop-in counterparts of the reference's `utils` hot-path modules."""
from .spectral_loss import (  # noqa: F401
    ACPRLoss, EVMLoss, SpectralLoss, compute_acpr, compute_evm, compute_nmse, create_spectral_loss,
)
from .quantization import (  # noqa: F401
    FakeQuantizeFunction, Q88Inference, QuantizationConfig, QuantizedLinear, dequantize_from_fixed_point,
    export_q88_flat, export_q88_hex, export_q88_weights, export_weights_binary, fake_quantize, fake_quantize_grad, generate_leaky_relu_params,
    Q88_BANK_OFFSETS, Q88_BANK_SIZE, generate_tanh_lut, load_weights_binary, quantize_to_fixed_point, to_q8_8,
)
