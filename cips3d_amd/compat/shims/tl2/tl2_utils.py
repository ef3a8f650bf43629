"""tl2.tl2_utils — the helpers the reference's cips3d scripts and models call."""
import json
import os
import sys


def get_class_repr(self):
    """exp/cips3d/models/generator.py:23 (repr of a module with its `repr_str`)"""
    return f"{self.__class__.__name__}({getattr(self, 'repr_str', '')})"


def dict2string(dict_obj=None, **kwargs):
    return str(dict_obj)


def parser_args_from_list(name, argv_list, type='list'):
    """exp/tests/test_cips3d.py:36, 45: the values following flag `name` in argv_list up to the next `--flag`"""
    if name not in argv_list:
        return [] if type == 'list' else None
    i = list(argv_list).index(name) + 1
    vals = []
    while i < len(argv_list) and not str(argv_list[i]).startswith('--'):
        vals.append(argv_list[i])
        i += 1
    if type == 'list':
        return vals
    return vals[0] if vals else None


class MaxToKeep:
    """train.py:65-66: `MaxToKeep.get_named_max_to_keep(name='ckpt', use_circle_number=True)
    .step_and_ret_circle_dir(global_cfg.tl_ckptdir)` — a ring of checkpoint directories <root>/<name>_<k>"""
    _named = {}

    def __init__(self, name, max_to_keep=4):
        self.name, self.max_to_keep, self.count = name, max_to_keep, 0

    @classmethod
    def get_named_max_to_keep(cls, name, max_to_keep=4, use_circle_number=True):
        if name not in cls._named:
            cls._named[name] = cls(name, max_to_keep)
        return cls._named[name]

    def step_and_ret_circle_dir(self, root):
        d = os.path.join(root, f"{self.name}_{self.count % self.max_to_keep:02d}")
        self.count += 1
        return d


def write_info_msg(saved_dir, info_msg):
    """train.py:71"""
    os.makedirs(saved_dir, exist_ok=True)
    with open(os.path.join(saved_dir, "0info.txt"), "w") as f:
        f.write(str(info_msg) + "\n")


class TL_tqdm:
    """train.py:296: `pbar = TL_tqdm(total=..., start=...)`, `pbar.update()` once per step"""

    def __init__(self, total, start=0, desc=''):
        self.total, self.n = total, start

    def update(self, n=1):
        self.n += n

    def get_string(self):
        return f"{self.n}/{self.total}"


def get_print_dict_str(ddict, outdir=None, suffix_str='', float_format="+.6f"):
    """train.py:509: one log line from a dict of dicts of scalars"""
    parts = []
    for k, v in ddict.items():
        if isinstance(v, dict):
            parts.append(f"{k}: " + ", ".join(f"{kk} {vv:{float_format}}" if isinstance(vv, float) else f"{kk} {vv}" for kk, vv in v.items()))
        else:
            parts.append(f"{k} {v}")
    s = "[" + "] [".join(parts) + "]"
    if outdir:
        s += f" [{outdir}]"
    return s + (f" {suffix_str}" if suffix_str else "")


class AverageMeter:
    def __init__(self):
        self.sum, self.count = 0.0, 0

    def update(self, val, n=1):
        self.sum += float(val) * n
        self.count += n

    @property
    def avg(self):
        return self.sum / max(self.count, 1)


def read_image_list_from_files(files, compress=False, ext=None):
    """exp/pigan/datasets.py:16: image paths listed one per line in text files"""
    if isinstance(files, str):
        files = [files]
    out = []
    for f in files:
        with open(f) as fh:
            out += [line.strip() for line in fh if line.strip()]
    return out


def json_dump(obj, path):
    with open(path, "w") as f:
        json.dump(obj, f, indent=1)


class Worker:
    """exp/tests/test_cips3d.py:899 (a background process around a shell command); not needed by the training scripts"""
    def __init__(self, name=None, args=()):
        self.args = args

    def start(self):
        raise NotImplementedError("tl2 shim: Worker is a ModelArts convenience, not provided")
