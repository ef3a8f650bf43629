"""cips3d_amd — MI355X-native (gfx950) implementation of the CIPS-3D generator / discriminator
hot path behind the reference's Python API.  See DESIGN.md and include/cips3d_hip.h."""
from .generator import GeneratorNerfINR, GeneratorNerfINR_freeze_NeRF  # noqa: F401
from .discriminator import Discriminator_MultiScale, Discriminator_MultiScale_Aux  # noqa: F401

__all__ = ["GeneratorNerfINR", "GeneratorNerfINR_freeze_NeRF", "Discriminator_MultiScale", "Discriminator_MultiScale_Aux"]
