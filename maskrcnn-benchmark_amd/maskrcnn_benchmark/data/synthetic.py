"""Synthetic COCO-shaped data: there is no network and no dataset in this
environment, so the training path is fed with generated samples of the shape the reference's
COCODataset + transforms + BatchCollator pipeline produces (reference data/datasets/coco.py:38-101,
data/transforms/build.py:5-42, data/collate_batch.py:5-20):

  image   float32 [3, H, W] (unit-variance noise), H x W = a COCO-like size after the
          min-800 / max-1333 resize (default 800 x 1333 -> padded to 800 x 1344 by the collator);
  target  BoxList (xyxy, size (W, H)) with fields "labels" int64 in 1..80 and "masks"
          SegmentationMask (binary, filled ellipses inscribed in the boxes), 8-20 objects.

Samples are a pure function of (seed, index): every rank / worker regenerates the same item.
"""
import math

import torch
import torch.utils.data

from maskrcnn_benchmark.structures.bounding_box import BoxList
from maskrcnn_benchmark.structures.image_list import to_image_list
from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask


class SyntheticCOCODataset(torch.utils.data.Dataset):
    def __init__(self, length=1024, height=800, width=1333, num_classes=81, min_objects=8, max_objects=20,
                 with_masks=True, seed=0, pixel_std=1.0):
        self.length, self.height, self.width = length, height, width
        self.num_classes = num_classes
        self.min_objects, self.max_objects = min_objects, max_objects
        self.with_masks = with_masks
        self.seed = seed
        # unit-variance pixels: with random-init weights and frozen (identity) batch-norm the
        # activations keep the input scale, so COCO's ~58-unit pixel std would blow the losses up
        self.pixel_std = pixel_std

    def __len__(self):
        return self.length

    def get_img_info(self, index):
        return {"height": self.height, "width": self.width}

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        H, W = self.height, self.width
        image = torch.randn(3, H, W, generator=g) * self.pixel_std
        n = int(torch.randint(self.min_objects, self.max_objects + 1, (1,), generator=g))
        # sqrt(area) log-uniform in [16, 0.6*H], aspect ratio in [0.5, 2]
        s = torch.exp(torch.empty(n).uniform_(math.log(16.0), math.log(0.6 * H), generator=g))
        r = torch.empty(n).uniform_(0.5, 2.0, generator=g)
        bw, bh = (s * r.sqrt()).clamp(max=W - 2), (s / r.sqrt()).clamp(max=H - 2)
        cx = torch.rand(n, generator=g) * (W - bw) + bw / 2
        cy = torch.rand(n, generator=g) * (H - bh) + bh / 2
        boxes = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2 - 1, cy + bh / 2 - 1], dim=1)
        target = BoxList(boxes, (W, H), mode="xyxy")
        target.add_field("labels", torch.randint(1, self.num_classes, (n,), generator=g))
        if self.with_masks:
            yy = torch.arange(H, dtype=torch.float32)[None, :, None]
            xx = torch.arange(W, dtype=torch.float32)[None, None, :]
            ell = (((xx - cx[:, None, None]) / (bw[:, None, None] / 2)) ** 2 +
                   ((yy - cy[:, None, None]) / (bh[:, None, None] / 2)) ** 2) <= 1.0
            target.add_field("masks", SegmentationMask(ell.to(torch.uint8), (W, H), mode="mask"))
        target = target.clip_to_image(remove_empty=True)
        return image, target, index


class BatchCollator(object):
    """[(image, target, id)] -> (ImageList padded to size_divisible, tuple(targets), tuple(ids))."""

    def __init__(self, size_divisible=0):
        self.size_divisible = size_divisible

    def __call__(self, batch):
        images, targets, ids = zip(*batch)
        return to_image_list(list(images), self.size_divisible), targets, ids


def make_data_loader(cfg, is_train=True, is_distributed=False, start_iter=0, images_per_gpu=None, length=None):
    """Synthetic replacement for the reference's data/build.py:105-171: per-rank batches of
    IMS_PER_BATCH / world_size images."""
    from maskrcnn_benchmark.utils.comm import get_rank, get_world_size
    world = get_world_size()
    if images_per_gpu is None:
        total = cfg.SOLVER.IMS_PER_BATCH if is_train else cfg.TEST.IMS_PER_BATCH
        assert total % world == 0, "IMS_PER_BATCH ({}) must be divisible by the number of GPUs ({})".format(total, world)
        images_per_gpu = total // world
    size = cfg.INPUT.MIN_SIZE_TRAIN[0] if is_train else cfg.INPUT.MIN_SIZE_TEST
    max_size = cfg.INPUT.MAX_SIZE_TRAIN if is_train else cfg.INPUT.MAX_SIZE_TEST
    ds = SyntheticCOCODataset(length=length or max(cfg.SOLVER.MAX_ITER * images_per_gpu, 64), height=size,
                              width=max_size, num_classes=cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES,
                              with_masks=cfg.MODEL.MASK_ON, seed=get_rank())
    return torch.utils.data.DataLoader(ds, batch_size=images_per_gpu, shuffle=False,
                                       num_workers=0, collate_fn=BatchCollator(cfg.DATALOADER.SIZE_DIVISIBILITY))
