"""tl2.proj.pil.pil_utils.np_to_pil (setup_evaluation.py:85: uint8 (C,H,W) array of a real image -> PIL)"""
import numpy as np
from PIL import Image


def np_to_pil(np_img, channel_first=False, range01=False):
    a = np.asarray(np_img)
    if channel_first:
        a = a.transpose(1, 2, 0)
    if range01:
        a = a * 255.0
    a = np.clip(a, 0, 255).astype(np.uint8)
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    return Image.fromarray(a)
