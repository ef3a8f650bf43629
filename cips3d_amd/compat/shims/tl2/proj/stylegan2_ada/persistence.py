"""tl2.proj.stylegan2_ada.persistence (exp/dev/nerf_inr/models/generator_nerf_inr.py:15: decorator only)"""


def persistent_class(orig_class):
    return orig_class
