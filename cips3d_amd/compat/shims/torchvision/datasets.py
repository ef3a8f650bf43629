class ImageFolder:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("torchvision shim: ImageFolder is not part of the cips3d path (the training set is the "
                                  "StyleGAN-style zip, tl2...dataset_stylegan3.dataset)")
