"""tl2.proj.fvcore — MODEL_REGISTRY / build_model (call sites: exp/cips3d/models/generator.py:17, 1158, 1954;
discriminator.py:10, 588; train.py:228-229, 300).

`@MODEL_REGISTRY.register(name_prefix=__name__)` registers a class as "<module>.<ClassName>"; a bare
`@MODEL_REGISTRY.register()` as "<ClassName>" (the dataset: ffhq_exp.yaml:104-106).  `build_model(cfg, kwargs_priority=False,
cfg_to_kwargs=True, **kwargs)` imports `cfg.register_modules`, looks `cfg.name` up and calls it with the remaining keys of
cfg merged with kwargs — kwargs win when kwargs_priority (train.py:229: diffaug=..., :300: resize_resolution=...)."""
import importlib


class Registry:
    def __init__(self, name="MODEL"):
        self._name, self._d = name, {}

    def register(self, obj=None, name=None, name_prefix=None):
        def deco(cls):
            key = (name_prefix + "." if name_prefix else "") + (name or cls.__name__)
            self._d[key] = cls
            return cls
        if obj is not None and callable(obj):
            return deco(obj)
        return deco

    def get(self, name):
        if name not in self._d:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry! (have: {sorted(self._d)})")
        return self._d[name]

    def __contains__(self, name):
        return name in self._d


MODEL_REGISTRY = Registry("MODEL")


def build_model(cfg, kwargs_priority=False, cfg_to_kwargs=True, **kwargs):
    cfg = dict(cfg)
    for m in cfg.pop("register_modules", []) or []:
        importlib.import_module(m)
    name = cfg.pop("name")
    cls = MODEL_REGISTRY.get(name)
    merged = {**cfg, **kwargs} if kwargs_priority else {**kwargs, **cfg}
    return cls(**merged) if cfg_to_kwargs else cls(cfg, **kwargs)
