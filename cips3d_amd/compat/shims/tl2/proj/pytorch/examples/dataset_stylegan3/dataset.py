"""tl2.proj.pytorch.examples.dataset_stylegan3.dataset — reader of the StyleGAN-style training set.

On-disk format (what the reference's `scripts/dataset_tool.py:398-541` writes, a copy of StyleGAN3's tool): a folder or an
uncompressed zip of images `<idx // 1000 :05d>/img<idx:08d>.png` (any PIL-readable extension is accepted, files are taken
in sorted order) plus an optional `dataset.json` `{"labels": [[fname, label], ...]}` — integer labels become one-hot
vectors, float vectors pass through, `null` means no labels.  All images must share one square power-of-two resolution.

API, from the call sites (exp/cips3d/configs/ffhq_exp.yaml:103-114, exp/cips3d/scripts/train.py:300-317, setup_evaluation.py):
  ImageFolderDataset_of_stylegan(path, use_labels=False, max_size=None, xflip=False, resize_resolution=None, random_seed=0)
      registered in MODEL_REGISTRY under its bare class name; `dataset[i] -> (uint8 CHW array in [0, 255], label, i)`
      (train.py:301 `# imgs, label, idx = dataset[0] # [0, 255]`); `max_size` keeps a seeded random subset, `xflip` doubles the
      set with mirrored copies (StyleGAN3 semantics); `resize_resolution` resamples every image (Lanczos, the filter
      dataset_tool.py uses when it resizes at creation time — tl2's own choice cannot be checked offline);
  get_training_dataloader(dataset, rank, num_gpus, batch_size, num_workers, shuffle=True, sampler_seed=0)
      an endless DataLoader over an InfiniteSampler (rank-strided, windowed re-shuffling) with the GLOBAL batch size
      (train.py:302-305 passes batch_size * world_size), i.e. batch_size // num_gpus images per rank and step;
  to_norm_tensor(imgs, device) -> float32 in [-1, 1] (train.py:317).

Provenance: the class layout (`Dataset` with `_raw_idx` / `_xflip` / `_get_raw_labels`, `ImageFolderDataset`, `InfiniteSampler`)
restates NVIDIA StyleGAN3's `training/dataset.py` and `torch_utils/misc.py` from memory — that is the format's de-facto
definition, and what tl2 itself wraps; it is not part of /root/reference and not on the measured path.
"""
import json
import os
import zipfile

import numpy as np
import PIL.Image
import torch

from tl2.proj.fvcore import MODEL_REGISTRY


class Dataset(torch.utils.data.Dataset):
    def __init__(self, name, raw_shape, max_size=None, use_labels=False, xflip=False, random_seed=0):
        self._name = name
        self._raw_shape = list(raw_shape)
        self._use_labels = use_labels
        self._raw_labels = None
        self._label_shape = None
        self._raw_idx = np.arange(self._raw_shape[0], dtype=np.int64)
        if (max_size is not None) and (self._raw_idx.size > max_size):
            np.random.RandomState(random_seed).shuffle(self._raw_idx)
            self._raw_idx = np.sort(self._raw_idx[:max_size])
        self._xflip = np.zeros(self._raw_idx.size, dtype=np.uint8)
        if xflip:
            self._raw_idx = np.tile(self._raw_idx, 2)
            self._xflip = np.concatenate([self._xflip, np.ones_like(self._xflip)])

    def _get_raw_labels(self):
        if self._raw_labels is None:
            self._raw_labels = self._load_raw_labels() if self._use_labels else None
            if self._raw_labels is None:
                self._raw_labels = np.zeros([self._raw_shape[0], 0], dtype=np.float32)
            assert isinstance(self._raw_labels, np.ndarray) and self._raw_labels.shape[0] == self._raw_shape[0]
            assert self._raw_labels.dtype in [np.float32, np.int64]
            if self._raw_labels.dtype == np.int64:
                assert self._raw_labels.ndim == 1 and np.all(self._raw_labels >= 0)
        return self._raw_labels

    def close(self):
        pass

    def _load_raw_image(self, raw_idx):
        raise NotImplementedError

    def _load_raw_labels(self):
        raise NotImplementedError

    def __getstate__(self):
        return dict(self.__dict__, _raw_labels=None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self._raw_idx.size

    def __getitem__(self, idx):
        image = self._load_raw_image(self._raw_idx[idx])
        assert isinstance(image, np.ndarray) and image.dtype == np.uint8 and list(image.shape) == self.image_shape
        if self._xflip[idx]:
            image = image[:, :, ::-1]
        return image.copy(), self.get_label(idx), idx

    def get_label(self, idx):
        label = self._get_raw_labels()[self._raw_idx[idx]]
        if label.dtype == np.int64:
            onehot = np.zeros(self.label_shape, dtype=np.float32)
            onehot[label] = 1
            label = onehot
        return label.copy()

    @property
    def name(self):
        return self._name

    @property
    def image_shape(self):
        return list(self._raw_shape[1:])

    @property
    def num_channels(self):
        return self.image_shape[0]

    @property
    def resolution(self):
        return self.image_shape[1]

    @property
    def label_shape(self):
        if self._label_shape is None:
            raw = self._get_raw_labels()
            self._label_shape = [int(np.max(raw)) + 1] if raw.dtype == np.int64 else raw.shape[1:]
        return list(self._label_shape)

    @property
    def label_dim(self):
        return self.label_shape[0]

    @property
    def has_labels(self):
        return any(x != 0 for x in self.label_shape)


@MODEL_REGISTRY.register()
class ImageFolderDataset_of_stylegan(Dataset):
    def __init__(self, path, resolution=None, resize_resolution=None, **super_kwargs):
        self._path = path
        self._zipfile = None
        self._resize = int(resize_resolution) if resize_resolution else None
        if os.path.isdir(self._path):
            self._type = 'dir'
            self._all_fnames = {os.path.relpath(os.path.join(root, fname), start=self._path)
                                for root, _dirs, files in os.walk(self._path) for fname in files}
        elif os.path.splitext(self._path)[1].lower() == '.zip':
            self._type = 'zip'
            self._all_fnames = set(self._get_zipfile().namelist())
        else:
            raise IOError(f"'{path}': the dataset path must point to a directory or a zip")
        PIL.Image.init()
        self._image_fnames = sorted(f for f in self._all_fnames if os.path.splitext(f)[1].lower() in PIL.Image.EXTENSION)
        if len(self._image_fnames) == 0:
            raise IOError(f"'{path}': no image files found")
        name = os.path.splitext(os.path.basename(self._path))[0]
        self._raw_shape = None
        raw_shape = [len(self._image_fnames)] + list(self._load_raw_image(0).shape)
        if resolution is not None and (raw_shape[2] != resolution or raw_shape[3] != resolution):
            raise IOError('Image files do not match the specified resolution')
        super().__init__(name=name, raw_shape=raw_shape, **super_kwargs)

    def _get_zipfile(self):
        if self._zipfile is None:
            self._zipfile = zipfile.ZipFile(self._path)
        return self._zipfile

    def _open_file(self, fname):
        if self._type == 'dir':
            return open(os.path.join(self._path, fname), 'rb')
        return self._get_zipfile().open(fname, 'r')

    def close(self):
        try:
            if self._zipfile is not None:
                self._zipfile.close()
        finally:
            self._zipfile = None

    def __getstate__(self):
        return dict(super().__getstate__(), _zipfile=None)     # a worker process re-opens the archive

    def _load_raw_image(self, raw_idx):
        fname = self._image_fnames[raw_idx]
        with self._open_file(fname) as f:
            img = PIL.Image.open(f)
            img.load()
        if img.mode not in ("RGB", "L"):
            # palette / RGBA / 16-bit / CMYK files: the training set is RGB (or grey) uint8 — normalise here instead of handing D a
            # wrong channel count or tripping the uint8 assert in the middle of training
            img = img.convert("L" if img.mode in ("1", "I;16", "I", "F", "LA") else "RGB")
        if self._resize is not None and img.size != (self._resize, self._resize):
            img = img.resize((self._resize, self._resize), PIL.Image.LANCZOS)
        image = np.array(img)
        if image.ndim == 2:
            image = image[:, :, np.newaxis]
        return np.ascontiguousarray(image.transpose(2, 0, 1))   # HWC -> CHW

    def _load_raw_labels(self):
        fname = 'dataset.json'
        if fname not in self._all_fnames:
            return None
        with self._open_file(fname) as f:
            labels = json.load(f)['labels']
        if labels is None:
            return None
        labels = dict(labels)
        labels = [labels[fname.replace('\\', '/')] for fname in self._image_fnames]
        labels = np.array(labels)
        return labels.astype({1: np.int64, 2: np.float32}[labels.ndim])


ImageFolderDataset = ImageFolderDataset_of_stylegan


class InfiniteSampler(torch.utils.data.Sampler):
    """endless index stream: rank r of `num_replicas` takes every num_replicas-th position of a shuffled order that keeps
    re-shuffling inside a sliding window (StyleGAN3's training sampler)"""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(len(self.dataset))
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        idx = 0
        while True:
            i = idx % order.size
            if idx % self.num_replicas == self.rank:
                yield int(order[i])
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            idx += 1


def get_training_dataloader(dataset, rank, num_gpus, batch_size, num_workers=0, shuffle=True, sampler_seed=0, pin_memory=True,
                            **kwargs):
    if batch_size % num_gpus:
        raise ValueError(f"global batch size {batch_size} is not a multiple of {num_gpus} ranks")
    sampler = InfiniteSampler(dataset=dataset, rank=rank, num_replicas=num_gpus, shuffle=shuffle, seed=sampler_seed)
    extra = dict(prefetch_factor=2, persistent_workers=True) if num_workers > 0 else {}
    return torch.utils.data.DataLoader(dataset=dataset, sampler=sampler, batch_size=batch_size // num_gpus,
                                       num_workers=num_workers, pin_memory=pin_memory and torch.cuda.is_available(), **extra)


def to_norm_tensor(imgs, device):
    """uint8 [0, 255] -> float32 [-1, 1] on `device` (train.py:317)"""
    return imgs.to(device, non_blocking=True).to(torch.float32) / 127.5 - 1.0
