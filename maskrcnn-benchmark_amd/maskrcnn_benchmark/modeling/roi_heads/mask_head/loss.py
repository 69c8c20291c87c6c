"""Mask loss and mask targets (reference roi_heads/mask_head/loss.py:11-143).

Target of a positive ROI = its matched ground-truth instance mask cropped to the (rounded) ROI
window and bilinearly resized to M x M (M = 28), cast to the mask's dtype
(structures/segmentation_mask.py:111-151: `crop` + `resize(..).type_as(masks)`).  The reference
builds these on the CPU one ROI at a time (device->host copy of the boxes, a Python loop, a
host->device copy of the result; "FIXME: CPU computation bottleneck").  `project_masks_on_boxes`
here evaluates the same crop window and the same bilinear formula (torch's align_corners=False
sampling: src = (dst + 0.5) * size/M - 0.5 clamped at 0, taps (i, min(i+1, size-1))) for all ROIs at
once on the device with four gathers, never leaving the GPU.
"""
import torch
from torch.nn import functional as F

from maskrcnn_benchmark import _C
from maskrcnn_benchmark.modeling.matcher import Matcher


def _axis_taps(lo, size, M):
    """crop start `lo` [P] and crop length `size` [P] (int64) -> (i0 [P,M], i1 [P,M] absolute pixel
    indices, frac [P,M] float32) of torch's bilinear resize to M samples."""
    scale = size.to(torch.float32) / M
    dst = torch.arange(M, device=lo.device, dtype=torch.float32)
    # ATen's CPU kernel evaluates scale*(dst+0.5)-0.5 as ONE fused multiply-add; the fp64 product of
    # two fp32 numbers is exact, so rounding the fp64 expression once to fp32 reproduces the FMA
    src = (scale.double()[:, None] * (dst.double()[None, :] + 0.5) - 0.5).to(torch.float32).clamp(min=0)
    i0 = src.floor().to(torch.int64)
    i0 = torch.minimum(i0, (size - 1)[:, None])
    i1 = torch.minimum(i0 + 1, (size - 1)[:, None])
    frac = src - i0.to(torch.float32)
    return i0 + lo[:, None], i1 + lo[:, None], frac


def project_masks_on_boxes(masks, mask_index, boxes, discretization_size):
    """masks [G,H,W] (any dtype; all instances of the batch stacked), mask_index [P] int64 (which
    instance each ROI crops), boxes [P,4] xyxy -> [P,M,M] float32 targets."""
    M = discretization_size
    G, H, W = masks.shape
    if _C.on_device(masks):
        # one workgroup per ROI, ATen's CPU operation order (csrc/targets.hip): bit-equal to the reference's path
        return _C.mask_targets(masks, mask_index, boxes, M)
    b = boxes.round().to(torch.int64)
    xmin = b[:, 0].clamp(min=0, max=W - 1)
    ymin = b[:, 1].clamp(min=0, max=H - 1)
    xmax = torch.maximum(b[:, 2].clamp(min=0, max=W), xmin + 1)
    ymax = torch.maximum(b[:, 3].clamp(min=0, max=H), ymin + 1)
    y0, y1, ly = _axis_taps(ymin, ymax - ymin, M)
    x0, x1, lx = _axis_taps(xmin, xmax - xmin, M)
    flat = masks.reshape(G, H * W)
    base = mask_index[:, None, None]

    def tap(yy, xx):
        return flat[base, yy[:, :, None] * W + xx[:, None, :]].to(torch.float32)

    hy, hx = 1.0 - ly, 1.0 - lx
    # operation order of ATen's CPU bilinear kernel (the reference resizes masks on the CPU):
    # ((h0*w0)*p00 + (h0*w1)*p01) + (h1*w0)*p10 + (h1*w1)*p11, every product rounded to fp32 —
    # matters because integer masks truncate a sum that may land one ulp below 1.0
    val = (hy[:, :, None] * hx[:, None, :]) * tap(y0, x0) + (hy[:, :, None] * lx[:, None, :]) * tap(y0, x1)
    val = val + (ly[:, :, None] * hx[:, None, :]) * tap(y1, x0)
    val = val + (ly[:, :, None] * lx[:, None, :]) * tap(y1, x1)
    if not masks.dtype.is_floating_point:
        val = val.to(masks.dtype).to(torch.float32)  # `.type_as(self.masks)`: integer masks truncate
    return val


class MaskRCNNLossComputation(object):
    def __init__(self, proposal_matcher, discretization_size):
        self.proposal_matcher = proposal_matcher
        self.discretization_size = discretization_size

    def __call__(self, proposals, mask_logits, targets):
        """proposals: the mask head's fixed-length positive slots (fields labels, matched_idxs);
        mask_logits [sum P, C, M, M]."""
        dev = mask_logits.device
        labels = torch.cat([p.get_field("labels") for p in proposals], dim=0)
        parts = []
        for p, t in zip(proposals, targets):  # per image: crop windows clamp to that image's size
            m = t.get_field("masks")
            m = (m.instances.masks if hasattr(m, "instances") else m.masks).to(dev)
            if m.shape[0] == 0 or len(p) == 0:
                parts.append(torch.zeros((len(p), self.discretization_size, self.discretization_size),
                                         dtype=torch.float32, device=dev))
                continue
            parts.append(project_masks_on_boxes(m, p.get_field("matched_idxs").clamp(min=0),
                                                p.convert("xyxy").bbox, self.discretization_size))
        mask_targets = torch.cat(parts, dim=0)
        pos = labels > 0
        if mask_targets.numel() == 0:
            return mask_logits.sum() * 0
        from ..box_head import loss as box_loss
        if box_loss._FUSED_LOSS and _C.on_device(mask_logits):
            # value + gradient in one pass (csrc/head_loss.hip; opt-in, see box_head/loss.py)
            return _C.mask_loss(mask_logits.float(), labels, mask_targets)
        # the class plane of every ROI by gather (reference: mask_logits[positive_inds, labels_pos], loss.py:137-139): its
        # backward is a scatter, where advanced indexing's is a sort-based index_put that leaves the device idle for
        # 0.1-0.2 ms per step (profiles/r04z_bench_f32_step_breakdown.txt: the gaps behind indexing_backward_kernel)
        M = mask_logits.shape[-1]
        cls = labels.clamp(min=0)[:, None, None, None].expand(-1, 1, mask_logits.shape[-2], M)
        logits = mask_logits.gather(1, cls).squeeze(1).float()
        bce = F.binary_cross_entropy_with_logits(logits, mask_targets, reduction="none")
        denom = (pos.sum() * bce[0].numel()).clamp(min=1).to(torch.float32)
        return torch.where(pos[:, None, None], bce, torch.zeros_like(bce)).sum() / denom


def make_roi_mask_loss_evaluator(cfg):
    H = cfg.MODEL.ROI_HEADS
    matcher = Matcher(H.FG_IOU_THRESHOLD, H.BG_IOU_THRESHOLD, allow_low_quality_matches=False)
    return MaskRCNNLossComputation(matcher, cfg.MODEL.ROI_MASK_HEAD.RESOLUTION)
