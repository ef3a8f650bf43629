"""tl2.launch.launch_utils — the reference's configuration plumbing, re-implemented from its call sites.

Command line of every exp/cips3d script (exp/tests/test_cips3d.py:884-916):
    --tl_config_file <yaml> --tl_command <section> --tl_outdir <dir> [--tl_resume --tl_resumedir <dir>] [--tl_debug]
    --tl_opts key value [key value ...]
YAML (exp/cips3d/configs/ffhq_exp.yaml): one top-level section per command; `base: <section>` inherits another section
(recursively; nested dicts merge key by key, `ffhq_exp.yaml:192-210`: `D_cfg: {diffaug: true}` overrides one key of the
inherited D_cfg); `--tl_opts` then overrides single keys, dotted for nested ones (`G_kwargs.num_steps 24`,
exp/cips3d/bash/afhq_exp/train_afhq_r128.sh:73), values parsed as YAML scalars.

`update_parser_defaults_from_yaml(parser)` (train.py:214, 595) fills `global_cfg` — an attribute dict with `.get`, `.dump`,
`.dump_to_file_with_command` — and sets the defaults of `parser` from the same-named keys.
"""
import argparse
import copy
import os
import sys

import yaml


class TLCfgNode(dict):
    """attribute-style dict (nested dicts are converted on assignment)"""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, TLCfgNode):
            return TLCfgNode(v)
        if isinstance(v, list):
            return [TLCfgNode._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return TLCfgNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        def un(v):
            if isinstance(v, dict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, list):
                return [un(x) for x in v]
            return v
        return un(self)

    def clone(self):
        return copy.deepcopy(self)

    def dump(self, *args, **kwargs):
        return yaml.safe_dump(self.to_dict(), *args, **kwargs)

    def dump_to_file(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            yaml.safe_dump(self.to_dict(), f)

    def dump_to_file_with_command(self, path, command):
        """train.py:68: the resolved configuration under its command name, so that the file can be fed back as --tl_config_file"""
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        body = {k: v for k, v in self.to_dict().items() if not str(k).startswith("tl_")}
        with open(path, "w") as f:
            yaml.safe_dump({command: body}, f)


global_cfg = TLCfgNode(tl_debug=False)


def _merge(base, over):
    """nested merge: dict values merge key by key, everything else is replaced"""
    out = copy.deepcopy(base)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def resolve_command(all_cfg, command, _seen=()):
    """the section `command` of a loaded YAML with its `base:` chain applied (ffhq_exp.yaml:130-224)"""
    if command not in all_cfg:
        raise KeyError(f"command '{command}' is not a section of the configuration (have: {sorted(map(str, all_cfg))[:20]} ...)")
    if command in _seen:
        raise ValueError(f"cyclic `base:` chain at '{command}'")
    sec = dict(all_cfg[command] or {})
    base = sec.pop("base", None)
    if base is None:
        return copy.deepcopy(sec)
    bases = base if isinstance(base, (list, tuple)) else [base]
    out = {}
    for b in bases:
        out = _merge(out, resolve_command(all_cfg, b, _seen + (command,)))
    return _merge(out, sec)


def apply_opts(cfg, opts):
    """--tl_opts key value ...: dotted keys address nested dicts, values are YAML scalars / flow collections"""
    if len(opts) % 2:
        raise ValueError(f"--tl_opts needs key value pairs, got {opts}")
    for k, v in zip(opts[0::2], opts[1::2]):
        node = cfg
        parts = str(k).split(".")
        for p in parts[:-1]:
            if not isinstance(node.get(p), dict):
                node[p] = {}
            node = node[p]
        node[parts[-1]] = yaml.safe_load(v) if isinstance(v, str) else v
    return cfg


def _tl_parser():
    p = argparse.ArgumentParser(add_help=False)
    p.add_argument("--tl_config_file", type=str, default=None)
    p.add_argument("--tl_command", type=str, default=None)
    p.add_argument("--tl_outdir", type=str, default="results/temp")
    p.add_argument("--tl_opts", type=str, nargs="*", default=[])
    p.add_argument("--tl_resume", action="store_true", default=False)
    p.add_argument("--tl_resumedir", type=str, default=None)
    p.add_argument("--tl_debug", action="store_true", default=False)
    p.add_argument("--tl_time_str", type=str, default="")
    return p


def build_cfg(argv=None):
    """-> (tl args namespace, resolved TLCfgNode) from a command line (default sys.argv)"""
    args, _ = _tl_parser().parse_known_args(sys.argv[1:] if argv is None else argv)
    cfg = {}
    if args.tl_config_file and args.tl_config_file != "none":
        with open(args.tl_config_file) as f:
            all_cfg = yaml.safe_load(f) or {}
        if args.tl_command and args.tl_command != "none":
            cfg = resolve_command(all_cfg, args.tl_command)
    cfg = apply_opts(cfg, list(args.tl_opts))
    node = TLCfgNode(cfg)
    node.tl_config_file = args.tl_config_file
    node.tl_command = args.tl_command
    node.tl_outdir = args.tl_outdir
    node.tl_ckptdir = os.path.join(args.tl_outdir, "ckptdir")
    node.tl_imgdir = os.path.join(args.tl_outdir, "imgdir")
    node.tl_logfile = os.path.join(args.tl_outdir, "log.txt")
    node.tl_resume = bool(args.tl_resume)
    node.tl_resumedir = args.tl_resumedir
    node.tl_debug = bool(args.tl_debug)
    node.tl_opts = list(args.tl_opts)
    return args, node


def update_parser_defaults_from_yaml(parser=None, is_main_process=True, use_cfg_as_args=False, append_local_rank=False, **kwargs):
    """train.py:214 (inside every rank: parser=None) and :595 (main: the script's own parser).  Fills `global_cfg` in place
    (the scripts hold a reference to it) and makes the configuration's scalars the defaults of same-named parser arguments."""
    args, node = build_cfg()
    global_cfg.clear()
    for k, v in node.items():
        global_cfg[k] = v
    if is_main_process:
        os.makedirs(global_cfg.tl_outdir, exist_ok=True)
        os.makedirs(global_cfg.tl_ckptdir, exist_ok=True)
        global_cfg.dump_to_file(os.path.join(global_cfg.tl_outdir, "config_command.yaml"))
    if parser is not None:
        for flag in ("tl_config_file", "tl_command", "tl_outdir", "tl_resumedir", "tl_time_str"):
            try:
                parser.add_argument(f"--{flag}", type=str, default=getattr(args, flag))
            except argparse.ArgumentError:
                pass
        for flag in ("tl_resume", "tl_debug"):
            try:
                parser.add_argument(f"--{flag}", action="store_true", default=getattr(args, flag))
            except argparse.ArgumentError:
                pass
        try:
            parser.add_argument("--tl_opts", type=str, nargs="*", default=list(args.tl_opts))
        except argparse.ArgumentError:
            pass
        known = {a.dest for a in parser._actions}
        parser.set_defaults(**{k: v for k, v in global_cfg.items() if k in known and not isinstance(v, dict)})
    return global_cfg


def get_append_cmd_str(args):
    """exp/tests/test_cips3d.py:915: the --tl_* flags of a parsed namespace as a command-line fragment"""
    s = f"--tl_config_file {args.tl_config_file} --tl_command {args.tl_command} --tl_outdir {args.tl_outdir}"
    if getattr(args, "tl_resume", False):
        s += f" --tl_resume --tl_resumedir {args.tl_resumedir}"
    if getattr(args, "tl_opts", None):
        s += " --tl_opts " + " ".join(map(str, args.tl_opts))
    return s
