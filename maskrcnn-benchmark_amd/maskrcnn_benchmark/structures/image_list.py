"""ImageList / to_image_list (reference structures/image_list.py:7-72): a batch of differently
sized images zero-padded into one `[N,C,H,W]` tensor (H, W rounded up to `size_divisible`), with
the un-padded `(h, w)` of each image kept alongside."""
import torch


class ImageList(object):
    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes  # list of (h, w)

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def to_image_list(tensors, size_divisible=0):
    if isinstance(tensors, torch.Tensor) and size_divisible > 0:
        tensors = [tensors] if tensors.dim() == 3 else list(tensors)
    if isinstance(tensors, ImageList):
        return tensors
    if isinstance(tensors, torch.Tensor):
        if tensors.dim() == 3:
            tensors = tensors[None]
        assert tensors.dim() == 4
        return ImageList(tensors, [t.shape[-2:] for t in tensors])
    if isinstance(tensors, (tuple, list)):
        C = tensors[0].shape[0]
        H = max(t.shape[1] for t in tensors)
        W = max(t.shape[2] for t in tensors)
        if size_divisible > 0:
            d = int(size_divisible)
            H, W = (H + d - 1) // d * d, (W + d - 1) // d * d
        batch = tensors[0].new_zeros((len(tensors), C, H, W))
        for img, slot in zip(tensors, batch):
            slot[:, :img.shape[1], :img.shape[2]].copy_(img)
        return ImageList(batch, [t.shape[-2:] for t in tensors])
    raise TypeError("Unsupported type for to_image_list: {}".format(type(tensors)))
