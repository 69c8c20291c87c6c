"""Instance masks of one image (reference structures/segmentation_mask.py).

Only the dense representation is built: `BinaryMaskList` (reference :30-204) — the synthetic
COCO-shaped dataset produces binary masks directly, and polygon rasterisation needs pycocotools,
which is a third-party dependency outside the hot path.  `SegmentationMask(masks, size,
mode="mask")` is the reference's wrapper name (:444-557) and is kept so targets look the same.
"""
import torch
from torch.nn.functional import interpolate

FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1


class BinaryMaskList(object):
    """`masks` [n, H, W] (uint8 or bool or float) for an image of `size` = (W, H)."""

    def __init__(self, masks, size):
        assert isinstance(size, (list, tuple)) and len(size) == 2
        if isinstance(masks, BinaryMaskList):
            masks = masks.masks.clone()
        elif isinstance(masks, (list, tuple)):
            masks = torch.stack(list(masks), dim=0).clone() if len(masks) else torch.empty([0, size[1], size[0]])
        elif not isinstance(masks, torch.Tensor):
            raise RuntimeError("Type of `masks` argument could not be interpreted: %s" % type(masks))
        if masks.dim() == 2:
            masks = masks[None]
        assert masks.dim() == 3
        assert masks.shape[1] == size[1], "%s != %s" % (masks.shape[1], size[1])
        assert masks.shape[2] == size[0], "%s != %s" % (masks.shape[2], size[0])
        self.masks = masks
        self.size = tuple(size)

    def transpose(self, method):
        return BinaryMaskList(self.masks.flip(1 if method == FLIP_TOP_BOTTOM else 2), self.size)

    @staticmethod
    def crop_window(box, width, height):
        """Integer crop window of the reference's `crop` (:111-131): rounded, clamped, >= 1 px."""
        xmin, ymin, xmax, ymax = [round(float(b)) for b in box]
        assert xmin <= xmax and ymin <= ymax, str(box)
        xmin = min(max(xmin, 0), width - 1)
        ymin = min(max(ymin, 0), height - 1)
        xmax = max(min(max(xmax, 0), width), xmin + 1)
        ymax = max(min(max(ymax, 0), height), ymin + 1)
        return xmin, ymin, xmax, ymax

    def crop(self, box):
        xmin, ymin, xmax, ymax = self.crop_window(box, *self.size)
        return BinaryMaskList(self.masks[:, ymin:ymax, xmin:xmax], (xmax - xmin, ymax - ymin))

    def resize(self, size):
        if not isinstance(size, (list, tuple)):
            size = (size, size)
        width, height = map(int, size)
        assert width > 0 and height > 0
        out = interpolate(self.masks[None].float(), size=(height, width), mode="bilinear",
                          align_corners=False)[0].type_as(self.masks)
        return BinaryMaskList(out, (width, height))

    def to(self, *args, **kwargs):
        return BinaryMaskList(self.masks.to(*args, **kwargs), self.size)

    def get_mask_tensor(self):
        return self.masks.squeeze(0) if self.masks.shape[0] == 1 else self.masks

    def __len__(self):
        return len(self.masks)

    def __getitem__(self, index):
        if self.masks.numel() == 0:
            raise RuntimeError("Indexing empty BinaryMaskList")
        return BinaryMaskList(self.masks[index], self.size)

    def __iter__(self):
        return iter(self.masks)

    def __repr__(self):
        return "{}(num_instances={}, image_width={}, image_height={})".format(
            self.__class__.__name__, len(self.masks), self.size[0], self.size[1])


class SegmentationMask(object):
    def __init__(self, instances, size, mode="mask"):
        if mode != "mask":
            raise NotImplementedError("only mode='mask' is built (polygons need pycocotools)")
        self.instances = instances if isinstance(instances, BinaryMaskList) else BinaryMaskList(instances, size)
        self.size = tuple(size)
        self.mode = mode

    def transpose(self, method):
        return SegmentationMask(self.instances.transpose(method), self.size, self.mode)

    def crop(self, box):
        c = self.instances.crop(box)
        return SegmentationMask(c, c.size, self.mode)

    def resize(self, size, *args, **kwargs):
        r = self.instances.resize(size)
        return SegmentationMask(r, r.size, self.mode)

    def to(self, *args, **kwargs):
        return SegmentationMask(self.instances.to(*args, **kwargs), self.size, self.mode)

    def convert(self, mode):
        if mode != "mask":
            raise NotImplementedError("only mode='mask' is built")
        return self

    def get_mask_tensor(self):
        return self.instances.get_mask_tensor()

    def __len__(self):
        return len(self.instances)

    def __getitem__(self, item):
        return SegmentationMask(self.instances[item], self.size, self.mode)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i:i + 1]

    def __repr__(self):
        return "{}(num_instances={}, image_width={}, image_height={}, mode={})".format(
            self.__class__.__name__, len(self.instances), self.size[0], self.size[1], self.mode)
