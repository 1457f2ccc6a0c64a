"""Drop-in counterparts of the reference's `models` package."""
from .tdnn_generator import (  # noqa: F401
    FakeQuantizeModule, MemoryTapAssembly, StraightThroughQuantize, TDNNGenerator, TDNNGeneratorQAT,
    create_generator, fake_quantize,
)
