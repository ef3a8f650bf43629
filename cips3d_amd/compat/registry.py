"""Registration of the drop-in models in the reference's model registry.

The reference builds its networks by name: `build_model(cfg)` imports `cfg.register_modules` and looks `cfg.name` up
in `tl2.proj.fvcore.MODEL_REGISTRY` (exp/cips3d/scripts/train.py:228-229; YAML keys exp/cips3d/configs/ffhq_exp.yaml:
44-46, 89-91); its own classes register as `@MODEL_REGISTRY.register(name_prefix=__name__)`
(exp/cips3d/models/generator.py:1158, 1954; discriminator.py:588).  Importing THIS module does the same for the
MI355X classes, so the only reference-side change is in the YAML:

    G_cfg_3D2D:
      register_modules: [cips3d_amd.compat.registry]
      name: cips3d_amd.compat.registry.GeneratorNerfINR            # or ...GeneratorNerfINR_freeze_NeRF
    D_cfg:
      register_modules: [cips3d_amd.compat.registry]
      name: cips3d_amd.compat.registry.Discriminator_MultiScale_Aux
"""
from ..generator import GeneratorNerfINR, GeneratorNerfINR_freeze_NeRF
from ..discriminator import Discriminator_MultiScale, Discriminator_MultiScale_Aux

CLASSES = (GeneratorNerfINR, GeneratorNerfINR_freeze_NeRF, Discriminator_MultiScale, Discriminator_MultiScale_Aux)


def register(registry=None, name_prefix=__name__):
    """Register the four classes in `registry` (default: tl2.proj.fvcore.MODEL_REGISTRY) -> list of registered names."""
    if registry is None:
        from tl2.proj.fvcore import MODEL_REGISTRY as registry
    for cls in CLASSES:
        registry.register(name_prefix=name_prefix)(cls)
    return [f"{name_prefix}.{cls.__name__}" for cls in CLASSES]


try:                                   # `register_modules: [cips3d_amd.compat.registry]` -> registered on import
    REGISTERED = register()
except ImportError:                    # tl2 not installed (this repo's own tests / bench): call register(registry) yourself
    REGISTERED = []
