"""Anchors (reference modeling/rpn/anchor_generator.py:34-291).

Cell anchors follow the Detectron recipe the reference inherits: start from the window
(0, 0, stride-1, stride-1), enumerate aspect ratios with ROUNDED widths/heights
(w = round(sqrt(area/ratio)), h = round(w*ratio)), then scale each about its centre by
size/stride.  Grid anchors are the cell anchors shifted by stride over the feature map, ordered
(y, x, anchor) — the order `permute_and_flatten` gives the head outputs.

The grid depends only on the feature-map shapes, so it is cached per (shapes, device) instead of
being rebuilt every iteration; `forward` still returns the reference's list (per image) of lists
(per level) of BoxLists with a "visibility" field, all sharing the cached tensors.
"""
import math

import numpy as np
import torch
from torch import nn

from maskrcnn_benchmark.structures.bounding_box import BoxList


def _window(ws, hs, x_ctr, y_ctr):
    ws, hs = ws[:, None], hs[:, None]
    return np.hstack((x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1),
                      x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1)))


def _center_form(a):
    w, h = a[2] - a[0] + 1, a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    """[len(ratios)*len(sizes), 4] float64 cell anchors (x1, y1, x2, y2), ratio-major."""
    scales = np.array(sizes, dtype=np.float64) / stride
    ratios = np.array(aspect_ratios, dtype=np.float64)
    base = np.array([0, 0, stride - 1, stride - 1], dtype=np.float64)
    w, h, cx, cy = _center_form(base)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    per_ratio = _window(ws, hs, cx, cy)
    rows = []
    for a in per_ratio:
        w, h, cx, cy = _center_form(a)
        rows.append(_window(w * scales, h * scales, cx, cy))
    return torch.from_numpy(np.vstack(rows))


class BufferList(nn.Module):
    """nn.ParameterList for buffers (reference :12-31)."""

    def __init__(self, buffers=None):
        super(BufferList, self).__init__()
        if buffers is not None:
            self.extend(buffers)

    def extend(self, buffers):
        offset = len(self)
        for i, b in enumerate(buffers):
            self.register_buffer(str(offset + i), b)
        return self

    def __len__(self):
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=(128, 256, 512), aspect_ratios=(0.5, 1.0, 2.0), anchor_strides=(8, 16, 32),
                 straddle_thresh=0):
        super(AnchorGenerator, self).__init__()
        if len(anchor_strides) == 1:
            cells = [generate_anchors(anchor_strides[0], sizes, aspect_ratios).float()]
        else:
            if len(anchor_strides) != len(sizes):
                raise RuntimeError("FPN should have #anchor_strides == #sizes")
            cells = [generate_anchors(s, z if isinstance(z, (tuple, list)) else (z,), aspect_ratios).float()
                     for s, z in zip(anchor_strides, sizes)]
        self.strides = anchor_strides
        self.cell_anchors = BufferList(cells)
        self.straddle_thresh = straddle_thresh
        self._grid_cache = {}
        self._vis_cache = {}          # insertion-ordered: least recently used first

    def num_anchors_per_location(self):
        return [len(c) for c in self.cell_anchors]

    def grid_anchors(self, grid_sizes):
        out = []
        for (gh, gw), stride, base in zip(grid_sizes, self.strides, self.cell_anchors):
            dev = base.device
            sx = torch.arange(0, gw * stride, step=stride, dtype=torch.float32, device=dev)
            sy = torch.arange(0, gh * stride, step=stride, dtype=torch.float32, device=dev)
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
            out.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
        return out

    def cached_grid_anchors(self, grid_sizes):
        key = (tuple((int(h), int(w)) for h, w in grid_sizes), str(self.cell_anchors._buffers["0"].device))
        if key not in self._grid_cache:
            self._grid_cache[key] = self.grid_anchors(grid_sizes)
        return self._grid_cache[key]

    def visibility(self, anchors, image_width, image_height):
        if self.straddle_thresh >= 0:
            t = self.straddle_thresh
            return ((anchors[..., 0] >= -t) & (anchors[..., 1] >= -t) &
                    (anchors[..., 2] < image_width + t) & (anchors[..., 3] < image_height + t))
        return torch.ones(anchors.shape[0], dtype=torch.bool, device=anchors.device)

    def add_visibility_to(self, boxlist):
        w, h = boxlist.size
        boxlist.add_field("visibility", self._cached_visibility(boxlist.bbox, w, h))

    def _cached_visibility(self, anchors, image_width, image_height):
        """visibility is a function of (the cached per-level anchor tensor, the image size): seven elementwise launches
        per (image, level) and step when recomputed — 70 per step at 2 images x 5 levels.  Bounded cache (real data
        brings many image sizes): least recently used entries go first."""
        key = (anchors.data_ptr(), tuple(anchors.shape), str(anchors.device), int(image_width), int(image_height))
        hit = self._vis_cache.pop(key, None)
        if hit is None or hit[0] is not anchors:
            hit = (anchors, self.visibility(anchors, image_width, image_height))
            while len(self._vis_cache) >= 256:
                self._vis_cache.pop(next(iter(self._vis_cache)))
        self._vis_cache[key] = hit
        return hit[1]

    def forward(self, image_list, feature_maps):
        per_level = self.cached_grid_anchors([f.shape[-2:] for f in feature_maps])
        anchors = []
        for (image_height, image_width) in image_list.image_sizes:
            in_image = []
            for a in per_level:
                b = BoxList(a, (image_width, image_height), mode="xyxy")
                self.add_visibility_to(b)
                in_image.append(b)
            anchors.append(in_image)
        return anchors


def make_anchor_generator(config):
    R = config.MODEL.RPN
    if R.USE_FPN:
        assert len(R.ANCHOR_STRIDE) == len(R.ANCHOR_SIZES), "FPN should have len(ANCHOR_STRIDE) == len(ANCHOR_SIZES)"
    else:
        assert len(R.ANCHOR_STRIDE) == 1, "Non-FPN should have a single ANCHOR_STRIDE"
    return AnchorGenerator(R.ANCHOR_SIZES, R.ASPECT_RATIOS, R.ANCHOR_STRIDE, R.STRADDLE_THRESH)


def make_anchor_generator_retinanet(config):
    R = config.MODEL.RETINANET
    assert len(R.ANCHOR_STRIDES) == len(R.ANCHOR_SIZES), "Only support FPN now"
    sizes = tuple(tuple(size * R.OCTAVE ** (k / float(R.SCALES_PER_OCTAVE)) for k in range(R.SCALES_PER_OCTAVE))
                  for size in R.ANCHOR_SIZES)
    return AnchorGenerator(sizes, R.ASPECT_RATIOS, R.ANCHOR_STRIDES, R.STRADDLE_THRESH)


del math
