"""tl2.proj.fvcore.checkpoint.Checkpointer (gen_images.py:102: `Checkpointer(G_ema).load_state_dict_from_file(pkl, rank=rank)`)"""
from cips3d_amd.checkpoint import Checkpointer  # noqa: F401
