"""tl2.proj.pytorch.pytorch_hook.VerboseModel (generator.py:20: debug printing of layer shapes; a no-op here)"""


class VerboseModel:
    @staticmethod
    def forward_verbose(*args, **kwargs):
        return None
