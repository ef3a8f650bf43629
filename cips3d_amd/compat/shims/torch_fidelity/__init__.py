"""torch_fidelity stand-in.  exp/cips3d/scripts/eval_fid.py:9, 42-48 computes FID / KID with the torch-fidelity package (a
git submodule of the reference whose directory is empty in the checkout, `torch_fidelity_lib/`) on top of the pretrained
Inception-v3 weights (`weights-inception-2015-12-05-6726825d.pth`, downloaded at run time).  Neither the package nor the
weights exist offline, so the metric itself cannot be provided: everything up to its input — the generated uint8 JPEGs
(cips3d_amd.evaluation.gen_images, bit-exact quantiser) and the real-image folder (setup_evaluation.py) — is.  Install the
real package to evaluate; this module only lets train.py import and fails loudly if the metric is requested."""


def calculate_metrics(*args, **kwargs):
    raise NotImplementedError("torch_fidelity is not installed: FID / KID need the torch-fidelity package and its Inception "
                              "weights (cips3d_amd/compat/shims/torch_fidelity/__init__.py)")
