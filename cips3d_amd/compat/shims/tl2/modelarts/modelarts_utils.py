"""tl2.modelarts.modelarts_utils — Huawei ModelArts / OBS plumbing: not applicable off that cloud, every entry is a no-op."""


def setup_tl_outdir_obs(*args, **kwargs):
    return None


def modelarts_sync_results_dir(*args, **kwargs):
    return None


def modelarts_finetune(*args, **kwargs):
    return None
