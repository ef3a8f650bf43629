"""torchvision.transforms.functional (exp/comm/comm_utils.py:15 imports the module; only demo helpers call into it)"""


def __getattr__(name):
    raise AttributeError(f"torchvision shim: transforms.functional.{name} is not part of the cips3d path")
