"""tl2.modelarts.moxing_utils — OBS bucket copies (train.py:216-217, 272, 299, 572): local paths are already in place, so
every call is a no-op; `copy_data` checks that the local path exists when one is given, because a missing dataset is the
error the user needs to see."""
import os


def setup_tl_outdir_obs(cfg=None, unzip_code=False, **kwargs):
    return None


def modelarts_sync_results_dir(cfg=None, join=False, **kwargs):
    return None


def copy_data(rank=0, global_cfg=None, datapath_obs=None, datapath=None, disable=False, overwrite=False, unzip=False,
              **kwargs):
    if datapath and not disable and not os.path.exists(os.path.expanduser(datapath)):
        print(f"[tl2 shim] moxing_utils.copy_data: '{datapath}' does not exist locally and there is no OBS here")
    return None


def moxing_copy_parallel(*args, **kwargs):
    return None
