"""Fast R-CNN head loss + proposal sampling (reference roi_heads/box_head/loss.py:16-193).

`subsample`: label every proposal by IoU with the ground truth (fg >= 0.5, else bg; Matcher
without low-quality matches), draw 512 per image with <= 25 % foreground, attach class labels and
box-regression targets.  `__call__`: softmax cross-entropy over the sampled ROIs (mean) and
smooth-L1 (beta 1) on the positives' class-specific deltas, summed / #sampled.

The sampled set has a FIXED length (BATCH_SIZE_PER_IMAGE slots per image, positives first,
unfilled slots flagged invalid and label -1) so the head runs with static shapes and no `nonzero`.
"""
import os

import torch
from torch.nn import functional as F

from maskrcnn_benchmark import _C
from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from maskrcnn_benchmark.modeling.box_coder import BoxCoder
from maskrcnn_benchmark.modeling.matcher import Matcher
from maskrcnn_benchmark.modeling.rpn.loss import match_batched, pad_targets, smooth_l1_elementwise
from maskrcnn_benchmark.modeling.utils import cat
from maskrcnn_benchmark.structures.bounding_box import BoxList


_FUSED = os.environ.get("DETOPS_ROI_TARGETS", "fused") != "torch"   # A/B switch: the PyTorch composite on the GPU
# the fused box-head / mask-head loss kernels (csrc/head_loss.hip): value + gradient in one pass, ~65 launches fewer per step.
# Device parity (tests/test_targets_gpu.py) and the same-box A/B (profiles/r05a_head_loss_ab.txt: fp32 38.01 -> 37.80 ms,
# bf16 21.05 -> 20.95 ms) are in: the default.  DETOPS_HEAD_LOSS=torch keeps the ATen compositions for A/B runs.
_FUSED_LOSS = os.environ.get("DETOPS_HEAD_LOSS", "fused") != "torch"


def stack_proposals(proposals):
    """list[BoxList] (possibly different lengths, optional "valid" field) -> boxes [N,K,4],
    valid [N,K].  Lists that are the rows of one batch (the training RPN hands its proposals over like that,
    `BoxList.batch_rows`) are returned as that batch: no copy."""
    rows = [getattr(p, "batch_rows", None) for p in proposals]
    if rows[0] is not None and all(r is not None and r[0] is rows[0][0] and r[1] == i for i, r in enumerate(rows)):
        batch = rows[0][0]
        bb = batch["boxes"]
        # still the rows they were handed over as (a list whose boxes were replaced since falls through to the copy)
        if bb.shape[0] == len(proposals) and all(
                p.bbox.shape == bb[i].shape and p.bbox.data_ptr() == bb[i].data_ptr() and p.has_field("valid")
                and p.get_field("valid").data_ptr() == batch["valid"][i].data_ptr() for i, p in enumerate(proposals)):
            return bb, batch["valid"]
    K = max(len(p) for p in proposals)
    dev = proposals[0].bbox.device
    boxes = torch.zeros((len(proposals), K, 4), dtype=torch.float32, device=dev)
    valid = torch.zeros((len(proposals), K), dtype=torch.bool, device=dev)
    for i, p in enumerate(proposals):
        n = len(p)
        boxes[i, :n] = p.convert("xyxy").bbox
        valid[i, :n] = p.get_field("valid") if p.has_field("valid") else True
    return boxes, valid


class FastRCNNLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, box_coder, cls_agnostic_bbox_reg=False):
        self.proposal_matcher = proposal_matcher
        self.fg_bg_sampler = fg_bg_sampler
        self.box_coder = box_coder
        self.cls_agnostic_bbox_reg = cls_agnostic_bbox_reg

    def prepare_targets(self, boxes, valid, targets):
        """-> labels [N,K] int64 (class, 0 bg, -1 ignored/invalid), regression_targets [N,K,4],
        matched_idxs [N,K]."""
        gt, row_valid, extra = pad_targets(targets, boxes.device, ("labels",))
        matched = match_batched(self.proposal_matcher, gt, row_valid, boxes)
        labels = torch.gather(extra["labels"], 1, matched.clamp(min=0))
        labels = torch.where(matched == Matcher.BELOW_LOW_THRESHOLD, torch.zeros_like(labels), labels)
        labels = torch.where(matched == Matcher.BETWEEN_THRESHOLDS, torch.full_like(labels, -1), labels)
        labels = torch.where(valid, labels, torch.full_like(labels, -1))
        matched_gt = torch.gather(gt, 1, matched.clamp(min=0)[:, :, None].expand(-1, -1, 4))
        return labels, self.box_coder.encode(matched_gt, boxes), matched

    def _subsample_fused(self, proposals, boxes, valid, targets):
        """`subsample` with three launches behind the matcher (csrc/targets.hip: labels of all proposals, the sampler, the
        sampled slots' boxes / labels / encoded targets / matched indices / objectness) in place of the label chain, the
        encode of ALL proposals and the per-image indexing of every field."""
        gt, row_valid, extra = pad_targets(targets, boxes.device, ("labels",))
        matched = match_batched(self.proposal_matcher, gt, row_valid, boxes)
        labels = _C.match_labels(matched, extra["labels"], valid, torch.int64)
        idx, slot_valid = self.fg_bg_sampler.sample_fixed(labels)
        objectness = None
        if all(p.has_field("objectness") for p in proposals):
            rows = getattr(proposals[0], "batch_rows", None)
            if rows is not None and rows[0]["boxes"] is boxes and "objectness" in rows[0]:
                objectness = rows[0]["objectness"]
            else:
                objectness = boxes.new_zeros(boxes.shape[:2])
                for i, p in enumerate(proposals):
                    objectness[i, :len(p)] = p.get_field("objectness")
        b, lab, reg, mat, obj = _C.roi_head_targets(boxes, matched, gt, extra["labels"], valid, idx, slot_valid, objectness,
                                                    self.box_coder.weights)
        out = []
        for i, p in enumerate(proposals):
            bl = BoxList(b[i], p.size, mode="xyxy")
            bl.add_field("labels", lab[i])
            bl.add_field("regression_targets", reg[i])
            bl.add_field("matched_idxs", mat[i])
            bl.add_field("valid", slot_valid[i])
            if obj is not None:
                bl.add_field("objectness", obj[i])
            out.append(bl)
        return out

    def subsample(self, proposals, targets):
        boxes, valid = stack_proposals(proposals)
        if _C.on_device(boxes) and _FUSED and boxes.shape[1] > 0:
            self._proposals = self._subsample_fused(proposals, boxes, valid, targets)
            return self._proposals
        labels, regression_targets, matched = self.prepare_targets(boxes, valid, targets)
        idx, slot_valid = self.fg_bg_sampler.sample_fixed(labels)
        out = []
        for i, p in enumerate(proposals):
            sel = idx[i]
            bl = BoxList(boxes[i][sel], p.size, mode="xyxy")
            bl.add_field("labels", torch.where(slot_valid[i], labels[i][sel], torch.full_like(sel, -1)))
            bl.add_field("regression_targets", regression_targets[i][sel])
            bl.add_field("matched_idxs", matched[i][sel])
            bl.add_field("valid", slot_valid[i])
            if p.has_field("objectness"):
                obj = p.get_field("objectness")
                if obj.shape[0] < boxes.shape[1]:     # lists of different lengths were padded to the longest: so is the field
                    obj = torch.cat([obj, obj.new_zeros(boxes.shape[1] - obj.shape[0])])
                bl.add_field("objectness", obj[sel])
            out.append(bl)
        self._proposals = out
        return out

    def __call__(self, class_logits, box_regression):
        class_logits = cat(class_logits, dim=0).float()
        box_regression = cat(box_regression, dim=0).float()
        if not hasattr(self, "_proposals"):
            raise RuntimeError("subsample needs to be called before")
        proposals = self._proposals
        labels = cat([p.get_field("labels") for p in proposals], dim=0)
        regression_targets = cat([p.get_field("regression_targets") for p in proposals], dim=0)
        if _FUSED_LOSS and _C.on_device(class_logits) and labels.numel() > 0:
            # value + gradient of both losses in one pass (csrc/head_loss.hip) instead of ~35 launches with their autograd mirror
            return _C.fastrcnn_loss(class_logits, box_regression, labels, regression_targets, self.cls_agnostic_bbox_reg, 1.0)
        num_sampled = (labels >= 0).sum().clamp(min=1).to(torch.float32)
        classification_loss = F.cross_entropy(class_logits, labels, ignore_index=-1, reduction="sum") / num_sampled
        pos = labels > 0
        if self.cls_agnostic_bbox_reg:
            deltas = box_regression[:, 4:8]
        else:
            cols = 4 * labels.clamp(min=0)[:, None] + torch.arange(4, device=labels.device)
            deltas = torch.gather(box_regression, 1, cols)
        l1 = smooth_l1_elementwise(deltas, regression_targets, beta=1.0).sum(dim=1)
        box_loss = torch.where(pos, l1, torch.zeros_like(l1)).sum() / num_sampled
        return classification_loss, box_loss


def make_roi_box_loss_evaluator(cfg):
    H = cfg.MODEL.ROI_HEADS
    matcher = Matcher(H.FG_IOU_THRESHOLD, H.BG_IOU_THRESHOLD, allow_low_quality_matches=False)
    sampler = BalancedPositiveNegativeSampler(H.BATCH_SIZE_PER_IMAGE, H.POSITIVE_FRACTION)
    return FastRCNNLossComputation(matcher, sampler, BoxCoder(weights=H.BBOX_REG_WEIGHTS),
                                   cfg.MODEL.CLS_AGNOSTIC_BBOX_REG)
