"""Base class exp/pigan/datasets.py:17 derives a dataset from at import time (the cips3d configs never build it)."""
from torch.utils.data import Dataset


class Danbooru2019_Portraits(Dataset):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("tl2 shim: Danbooru2019_Portraits is not part of the cips3d path")
