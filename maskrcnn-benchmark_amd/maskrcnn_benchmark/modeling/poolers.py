"""Pooler: ROIAlign over a feature pyramid (reference modeling/poolers.py:11-121).

The reference assigns every ROI to a pyramid level on the host side of a `nonzero` per level, runs
one ROIAlign launch per level and scatters the results back (4 syncs + 4 launches + 4 index_puts
per call).  Here the whole thing is ONE kernel launch: `_C.roi_align_fpn_forward` evaluates the
LevelMapper formula per ROI on the device and reads from the right level's feature map
(include/detops.h: detops_roi_align_fpn_forward_f32).  The backward is likewise one launch that
produces every level's gradient map.
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from maskrcnn_benchmark import _C
from maskrcnn_benchmark.layers import ROIAlign

from .utils import cat


class LevelMapper(object):
    """FPN paper eq. 1: target level of each ROI from its scale (reference :11-42).  Kept as a
    host/torch-side helper (tests, tooling); the training path uses the in-kernel version."""

    def __init__(self, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
        self.k_min = k_min
        self.k_max = k_max
        self.s0 = canonical_scale
        self.lvl0 = canonical_level
        self.eps = eps

    def __call__(self, boxlists):
        s = torch.sqrt(cat([b.area() for b in boxlists]))
        lvl = torch.floor(self.lvl0 + torch.log2(s / self.s0 + self.eps))
        lvl = torch.clamp(lvl, min=self.k_min, max=self.k_max)
        return lvl.to(torch.int64) - self.k_min


PREPARE_BACKWARD_AT_FORWARD = False   # see _ROIAlignFPN.forward
_INDEX_COLUMNS = {}                   # Pooler.convert_to_roi_format: (lengths, dtype, device) -> [K, 1] image-index column


class _ROIAlignFPN(Function):
    """autograd glue of the fused multi-level ROIAlign."""

    @staticmethod
    def forward(ctx, rois, output_size, scales, sampling_ratio, k_min, k_max, out_channels_last, *features):
        # a channels-last pyramid is pooled in place by the NHWC kernels (and its gradient maps come back channels-last)
        ctx.nhwc = all(_C.is_channels_last(f) for f in features) and features[0].dtype == torch.float32
        out, levels = _C.roi_align_fpn_forward(features, rois, scales, output_size[0], output_size[1],
                                               sampling_ratio, k_min, k_max, out_channels_last=out_channels_last and ctx.nhwc)
        ctx.save_for_backward(rois, levels)
        ctx.cfg = (output_size, scales, sampling_ratio)
        ctx.shapes = [tuple(f.shape) for f in features]
        # The backward's pre-pass needs the ROIs and the map shapes only and CAN be issued now, on a side stream
        # (PREPARE_BACKWARD_AT_FORWARD).  Off by default — measured (profiles/r04_rejected_experiments.txt): the backward
        # entry drops from 106 to 92 us, but a second busy hardware queue next to the saturated compute queue costs the
        # STEP +0.5-0.9 ms in fp32 and +4 ms under bf16 autocast.
        ctx.prepared = None
        if PREPARE_BACKWARD_AT_FORWARD and any(ctx.needs_input_grad[7:]) and rois.is_cuda and not ctx.nhwc:
            ctx.prepared = _C.roi_align_fpn_backward_prepare(rois, levels, ctx.shapes, scales, output_size[0],
                                                             output_size[1], sampling_ratio)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        rois, levels = ctx.saved_tensors
        output_size, scales, sampling_ratio = ctx.cfg
        grads = _C.roi_align_fpn_backward(grad, rois, levels, ctx.shapes, scales, output_size[0],
                                          output_size[1], sampling_ratio, prepared=ctx.prepared, channels_last=ctx.nhwc)
        ctx.prepared = None
        return (None, None, None, None, None, None, None) + tuple(grads)


def roi_align_fpn(features, rois, output_size, scales, sampling_ratio, k_min, k_max, out_channels_last=False):
    """fp32 island (the reference marks ROIAlign.forward `@amp.float_function`, layers/roi_align.py:57)."""
    with torch.autocast(device_type=rois.device.type, enabled=False):
        return _ROIAlignFPN.apply(rois.float(), tuple(output_size), tuple(scales), sampling_ratio, k_min, k_max,
                                  bool(out_channels_last), *[f.float() for f in features])


class Pooler(nn.Module):
    """`Pooler(output_size, scales, sampling_ratio)`; `forward(x: list[Tensor], boxes: list[BoxList])
    -> Tensor [K, C, output_size...]` with K = total number of boxes, ordered image by image."""

    def __init__(self, output_size, scales, sampling_ratio):
        super(Pooler, self).__init__()
        self.poolers = nn.ModuleList(
            [ROIAlign(output_size, spatial_scale=s, sampling_ratio=sampling_ratio) for s in scales])
        self.output_size = output_size if isinstance(output_size, (tuple, list)) else (output_size, output_size)
        self.scales = tuple(float(s) for s in scales)
        self.sampling_ratio = sampling_ratio
        # scales are 1/2^k: recover k of the first / last level
        self.k_min = int(round(-torch.log2(torch.tensor(scales[0], dtype=torch.float32)).item()))
        self.k_max = int(round(-torch.log2(torch.tensor(scales[-1], dtype=torch.float32)).item()))
        self.map_levels = LevelMapper(self.k_min, self.k_max)
        # pooled tensor channels-last when the pyramid is (for a convolutional head running channels-last); the default
        # keeps [K, C, PH, PW] contiguous (the box head's FC layer flattens it)
        self.output_channels_last = False

    @staticmethod
    def convert_to_roi_format(boxes):
        """list[BoxList] -> [K,5] rows (image index, x1, y1, x2, y2); the index is stored as float
        like the reference (:78-89).  The index column depends on the list lengths only: it is kept per
        (lengths, dtype, device) — two concatenations per call instead of a fill + a concatenation per image."""
        first = boxes[0].bbox
        key = (tuple(b.bbox.shape[0] for b in boxes), first.dtype, str(first.device))
        col = _INDEX_COLUMNS.pop(key, None)
        if col is None:
            col = cat([first.new_full((n, 1), float(i)) for i, n in enumerate(key[0])], dim=0)   # on the device: no sync
            while len(_INDEX_COLUMNS) >= 32:
                _INDEX_COLUMNS.pop(next(iter(_INDEX_COLUMNS)))
        _INDEX_COLUMNS[key] = col            # most recently used last
        return torch.cat([col, cat([b.bbox for b in boxes], dim=0)], dim=1)

    def forward(self, x, boxes):
        rois = boxes if isinstance(boxes, torch.Tensor) else self.convert_to_roi_format(boxes)
        x = list(x)[:len(self.poolers)]  # extra pyramid levels (P6) are not pooled from (reference zip)
        if len(self.poolers) == 1:
            return self.poolers[0](x[0], rois)
        return roi_align_fpn(x, rois, self.output_size, self.scales, self.sampling_ratio,
                             self.k_min, self.k_max, out_channels_last=self.output_channels_last)


def make_pooler(cfg, head_name):
    head = cfg.MODEL[head_name]
    r = head.POOLER_RESOLUTION
    return Pooler(output_size=(r, r), scales=head.POOLER_SCALES, sampling_ratio=head.POOLER_SAMPLING_RATIO)
