"""On-disk checkpoint layout of the reference's training script.

exp/cips3d/scripts/train.py:249-256 collects

    model_dict = {'generator': G, 'G_ema': G_ema, 'discriminator': D, 'state_dict': {cur_fid, best_fid, worst_fid, step}}

and hands it to tl2's `torch_utils.save_models(save_dir, model_dict)` (train.py:72) / `load_models(save_dir, model_dict,
strict=False, rank=rank)` (:262, :272); gen_images.py:102 and eval_fid.py load one network from its file with
`Checkpointer(G_ema).load_state_dict_from_file(network_pkl)`.  tl2 is an unpinned PyPI package that is not vendored in
the reference; the layout it produces — one `torch.save`d `state_dict()` per entry, `<save_dir>/<name>.pth`, plain dicts
saved as they are — is what the released checkpoints (`G_ema.pth`, README.md:96-100) look like and what these functions
read and write, so a checkpoint directory written by the reference loads into the MI355X classes and vice versa (the
172 / 160 state_dict keys and shapes are identical, tests/test_compat_cpu.py)."""
import os

import torch


def _unwrap(obj):
    return obj.module if isinstance(obj, torch.nn.parallel.DistributedDataParallel) else obj


def _to_cpu(obj):
    """tensors anywhere inside nested dicts / lists / tuples (an optimiser's state -> idx -> exp_avg ...) -> CPU copies"""
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return type(obj)((k, _to_cpu(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def save_models(save_dir, model_dict, info_msg=None):
    """<save_dir>/<name>.pth for every entry: modules / optimisers / EMA helpers -> their state_dict(), plain dicts
    (the 'state_dict' entry: step, FID bookkeeping) as they are.  Tensors are moved to the CPU first."""
    os.makedirs(save_dir, exist_ok=True)
    for name, obj in model_dict.items():
        obj = _unwrap(obj)
        sd = obj.state_dict() if hasattr(obj, "state_dict") else obj
        torch.save(_to_cpu(sd), os.path.join(save_dir, f"{name}.pth"))
    if info_msg is not None:
        with open(os.path.join(save_dir, "0info.txt"), "w") as f:
            f.write(f"{info_msg}\n")


def load_models(save_dir, model_dict, strict=True, rank=0, verbose=False):
    """Inverse of save_models: load <save_dir>/<name>.pth into every entry in place.  Missing files are skipped when
    strict is False (train.py loads with strict=False so that e.g. a generator-only directory fine-tunes)."""
    for name, obj in model_dict.items():
        path = os.path.join(save_dir, f"{name}.pth")
        if not os.path.exists(path):
            if strict:
                raise FileNotFoundError(path)
            continue
        obj = _unwrap(obj)
        loaded = torch.load(path, map_location="cpu", weights_only=False)
        if isinstance(obj, torch.nn.Module):
            res = obj.load_state_dict(loaded, strict=strict)
            from .discriminator import invalidate_weight_cache
            invalidate_weight_cache(obj)
            if verbose and rank == 0:
                print(f"{name}: {res}")
        elif hasattr(obj, "load_state_dict"):
            obj.load_state_dict(loaded)
        elif isinstance(obj, dict):
            obj.clear()
            obj.update(loaded)
        else:
            raise TypeError(f"cannot load into {type(obj)}")


class Checkpointer:
    """`Checkpointer(G_ema).load_state_dict_from_file(network_pkl, rank=rank)` (gen_images.py:102)."""

    def __init__(self, model):
        self.model = _unwrap(model)

    def load_state_dict_from_file(self, path, rank=0, strict=True, key=None):
        """key: which state dict of a combined file ({'generator': ..., 'G_ema': ..., 'state_dict': ...}) to load; without it
        a file that holds several is ambiguous and raises"""
        sd = torch.load(path, map_location="cpu", weights_only=False)
        if key is not None:
            if not (isinstance(sd, dict) and isinstance(sd.get(key), dict)):
                raise KeyError(f"{path}: no state dict under key '{key}' (keys: {list(sd.keys()) if isinstance(sd, dict) else type(sd)})")
            sd = sd[key]
        # a file written as {'model': state_dict} / {'state_dict': state_dict} (common wrappers) loads too
        # — but only when exactly ONE candidate key is present: a file holding both 'generator' and 'G_ema' names two
        # networks, and picking one silently (with strict=False nothing would complain) loads the wrong one
        if isinstance(sd, dict) and sd and not any(torch.is_tensor(v) for v in sd.values()):
            cands = [k for k in ("model", "state_dict", "G_ema", "generator") if isinstance(sd.get(k), dict)]
            if len(cands) > 1:
                raise KeyError(f"{path}: several state dicts in one file ({cands}; keys: {list(sd.keys())}) — "
                               f"name the one to load: load_state_dict_from_file(path, key='{cands[0]}')")
            if cands:
                if rank == 0:
                    print(f"Checkpointer: loading the state dict under key '{cands[0]}' of {path}")
                sd = sd[cands[0]]
        res = self.model.load_state_dict(sd, strict=strict)
        from .discriminator import invalidate_weight_cache
        invalidate_weight_cache(self.model)
        return res

    def save_state_dict_to_file(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save({k: v.detach().cpu() for k, v in self.model.state_dict().items()}, path)
