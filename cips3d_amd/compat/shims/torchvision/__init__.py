"""torchvision stand-in: `utils.make_grid / save_image` are real (train.py:22, 98-140 — on cips3d_amd.evaluation, which
mirrors torchvision's arithmetic); `datasets` / `transforms` exist so that exp/pigan/datasets.py and exp/comm/comm_utils.py
import (their torchvision-based datasets are not on the cips3d path)."""
__version__ = "0.0+cips3d.shim"
from . import utils, datasets, transforms  # noqa: F401,E402
