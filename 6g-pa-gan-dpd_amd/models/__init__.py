"""Drop-in counterparts of the reference's `models` package."""
from .tdnn_generator import (  # noqa: F401
    FakeQuantizeModule, MemoryTapAssembly, StraightThroughQuantize, TDNNGenerator, TDNNGeneratorQAT,
    create_generator, fake_quantize,
)
from .pa_digital_twin import PADigitalTwin, VolterraPA, create_pa_digital_twin  # noqa: F401
from .discriminator import (  # noqa: F401
    Discriminator, SpectralNormLinear, WassersteinLoss, compute_gradient_penalty, create_discriminator,
)
from .chain import dpd_pa_chain  # noqa: F401  (no counterpart in the reference: the fused validation chain)
