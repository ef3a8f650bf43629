"""tl2...multi_process_main.dataset.ImageListDataset (imported by exp/cips3d/scripts/setup_evaluation.py; a list of image
files as a Dataset of (uint8 CHW tensor, index))"""
import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


class ImageListDataset(Dataset):
    def __init__(self, meta_file=None, image_list=None, transform=None, **kwargs):
        if image_list is None:
            with open(meta_file) as f:
                image_list = [l.strip() for l in f if l.strip()]
        self.image_list, self.transform = list(image_list), transform

    def __len__(self):
        return len(self.image_list)

    def __getitem__(self, idx):
        img = Image.open(self.image_list[idx]).convert("RGB")
        if self.transform is not None:
            return self.transform(img), idx
        return torch.from_numpy(np.asarray(img).transpose(2, 0, 1).copy()), idx
