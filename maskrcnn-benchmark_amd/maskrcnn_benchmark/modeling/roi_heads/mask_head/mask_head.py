"""ROIMaskHead (reference roi_heads/mask_head/mask_head.py:12-83).

Training consumes the box head's sampled proposals.  The reference keeps "only the positive
boxes" with a `nonzero` per image; here the sampler already orders every image's slots positives
first, so the mask head takes the first P = BATCH_SIZE_PER_IMAGE * POSITIVE_FRACTION slots of each
image (a static shape) and the loss ignores the slots whose label is not positive."""
import torch

from maskrcnn_benchmark.structures.bounding_box import BoxList

from .inference import make_roi_mask_post_processor
from .loss import make_roi_mask_loss_evaluator
from .roi_mask_feature_extractors import make_roi_mask_feature_extractor
from .roi_mask_predictors import make_roi_mask_predictor


_SLOT_INDICES = {}


def _first_slots(n, device):
    """arange(n) on `device`, built once per (n, device): the index tensor of "the first n slots" without a launch per step"""
    key = (int(n), str(device))
    t = _SLOT_INDICES.get(key)
    if t is None:
        if len(_SLOT_INDICES) >= 16:
            _SLOT_INDICES.clear()
        t = _SLOT_INDICES[key] = torch.arange(int(n), device=device)
    return t


def keep_only_positive_boxes(boxes, max_positives=None):
    """list[BoxList] with "labels" -> (positive-slot BoxLists, per-image slot INDEX TENSORS — the reference's return type,
    roi_heads/mask_head/mask_head.py:15-34, usable as `sel + offset` / `torch.cat(inds)`).  With `max_positives` (positives-
    first inputs) every image contributes exactly its first `max_positives` slots: the BoxLists are then VIEWS of the
    proposals' tensors (no indexing launches; do not write into them) and the indices a cached arange."""
    assert isinstance(boxes, (list, tuple)) and isinstance(boxes[0], BoxList)
    assert boxes[0].has_field("labels")
    out, inds = [], []
    for b in boxes:
        if max_positives is None:
            sel = (b.get_field("labels") > 0).nonzero().squeeze(1)
            out.append(b[sel])
        else:
            n = min(max_positives, len(b))
            sel = _first_slots(n, b.bbox.device)
            out.append(b[slice(0, n)])
        inds.append(sel)
    return out, inds


class ROIMaskHead(torch.nn.Module):
    def __init__(self, cfg, in_channels):
        super(ROIMaskHead, self).__init__()
        self.cfg = cfg.clone()
        self.feature_extractor = make_roi_mask_feature_extractor(cfg, in_channels)
        self.predictor = make_roi_mask_predictor(cfg, self.feature_extractor.out_channels)
        self.post_processor = make_roi_mask_post_processor(cfg)
        self.loss_evaluator = make_roi_mask_loss_evaluator(cfg)
        H = cfg.MODEL.ROI_HEADS
        self.max_positives = int(H.BATCH_SIZE_PER_IMAGE * H.POSITIVE_FRACTION)

    def forward(self, features, proposals, targets=None):
        if self.training:
            all_proposals = proposals
            fixed = all(p.has_field("valid") for p in proposals)
            proposals, positive_inds = keep_only_positive_boxes(
                proposals, self.max_positives if fixed else None)
        if self.training and self.cfg.MODEL.ROI_MASK_HEAD.SHARE_BOX_FEATURE_EXTRACTOR:
            # `features` are the box head's pooled features of ALL sampled proposals, image after image:
            # per-image slot indices become rows of that tensor by adding each image's offset (the
            # reference concatenates per-image boolean masks, mask_head.py:60-63)
            rows, base = [], 0
            for sel, b in zip(positive_inds, all_proposals):
                rows.append(sel + base)
                base += len(b)
            x = features[torch.cat(rows, dim=0)]
        else:
            x = self.feature_extractor(features, proposals)
        mask_logits = self.predictor(x)
        if not self.training:
            return x, self.post_processor(mask_logits, proposals), {}
        loss_mask = self.loss_evaluator(proposals, mask_logits, targets)
        return x, all_proposals, dict(loss_mask=loss_mask)


def build_roi_mask_head(cfg, in_channels):
    return ROIMaskHead(cfg, in_channels)
