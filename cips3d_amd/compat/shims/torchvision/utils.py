from cips3d_amd.evaluation import make_grid, save_image  # noqa: F401
