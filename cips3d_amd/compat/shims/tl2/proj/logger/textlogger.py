"""tl2.proj.logger.textlogger — scalar logging to text files (train.py:508, 545): one `<outdir>/textdir/<prefix>.<group>.<key>.log`
per scalar with "step value" lines (what tl2 turns into its txt figures)."""
import os


class TextLogger:
    def __init__(self):
        self.log_root = None

    def _root(self):
        if self.log_root is None:
            from tl2.launch.launch_utils import global_cfg
            self.log_root = os.path.join(global_cfg.get("tl_outdir", "results/temp"), "textdir")
        os.makedirs(self.log_root, exist_ok=True)
        return self.log_root

    def log(self, name, step, value):
        with open(os.path.join(self._root(), f"{name}.log"), "a") as f:
            f.write(f"{step} {value}\n")


global_textlogger = TextLogger()


def _scalar(v):
    try:
        return float(v)
    except Exception:
        return None


def summary_dict2txtfig(dict_data, prefix, step, textlogger=None, in_one_axe=False, **kwargs):
    tl = textlogger or global_textlogger
    for k, v in dict_data.items():
        s = _scalar(v)
        if s is not None:
            tl.log(f"{prefix}.{k}", step, s)


def summary_defaultdict2txtfig(default_dict, prefix, step, textlogger=None, in_one_figure=True, **kwargs):
    tl = textlogger or global_textlogger
    for group, d in default_dict.items():
        for k, v in d.items():
            s = _scalar(v)
            if s is not None:
                tl.log(f"{prefix}.{group}.{k}", step, s)
