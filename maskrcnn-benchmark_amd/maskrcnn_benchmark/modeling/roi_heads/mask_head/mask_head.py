"""ROIMaskHead (reference roi_heads/mask_head/mask_head.py:12-83).

Training consumes the box head's sampled proposals.  The reference keeps "only the positive
boxes" with a `nonzero` per image (a device->host synchronisation in the middle of the forward); here the
sampler already orders every image's slots positives first, so the mask head takes the FIRST n slots of each
image and the loss ignores the slots whose label is not positive (masked sums normalised by the number of
positives: the padding changes no value).

How many slots (round 6, late).  "fixed": n = P = BATCH_SIZE_PER_IMAGE * POSITIVE_FRACTION (128) for every image
— a static shape, no read-back, and the mask head's four 3 x 3 convolutions + deconvolution always run on N x 128
ROIs (6 ms of the 33 ms fp32 step) however few of them are positive.  "dynamic" (default in the eager step):
n = the image's positive count rounded up to a multiple of 16 (DETOPS_MASK_SLOT_GRANULE; at least one granule, at most P).  The counts leave the device
by an asynchronous copy issued right after the box head's sampler — a whole box-head forward before they are
needed — so the wait in front of the mask head is over when the host gets there and the device queue never
drains (`PositiveCounts`).  This is the reference's workload (it runs the mask head on the positives only),
with its shapes quantised (15 batch sizes at 2 images per GPU) so that MIOpen's find-db holds every key.  A captured HIP graph
cannot read anything back: engine/graph_step.py switches to "fixed"."""
import os

import torch

from maskrcnn_benchmark.structures.bounding_box import BoxList

from .inference import make_roi_mask_post_processor
from .loss import make_roi_mask_loss_evaluator
from .roi_mask_feature_extractors import make_roi_mask_feature_extractor
from .roi_mask_predictors import make_roi_mask_predictor


# "dynamic" | "fixed" | "<n>" or "<n0>,<n1>,..." (forced slot counts per image, cycled: tuning runs that must visit a batch size)
SLOT_MODE = os.environ.get("DETOPS_MASK_SLOTS", "dynamic")
SLOT_GRANULE = max(1, int(os.environ.get("DETOPS_MASK_SLOT_GRANULE", "16")))

_SLOT_INDICES = {}
_PINNED = {}


class PositiveCounts(object):
    """Per-image number of positive slots of the sampled proposals, on its way to the host: constructed right after the box
    head's sampler (a count + an asynchronous copy into pinned memory + an event), read by the mask head with `get()`."""

    def __init__(self, proposals):
        labels = [p.get_field("labels") for p in proposals]
        self.n = len(labels)
        self.event = None
        if self.n == 0:
            self.host = None
            return
        counts = torch.stack([(l > 0).sum() for l in labels]).to(torch.int32) if len({l.shape[0] for l in labels}) > 1 \
            else (torch.stack(labels) > 0).sum(dim=1, dtype=torch.int32)
        if counts.is_cuda:
            key = (self.n, counts.device.index)
            host = _PINNED.get(key)
            if host is None:
                host = _PINNED[key] = torch.empty((self.n,), dtype=torch.int32, pin_memory=True)
            host.copy_(counts, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
            self.host = host
        else:
            self.host = counts

    def get(self):
        if self.event is not None:
            self.event.synchronize()
        return [] if self.host is None else [int(v) for v in self.host.tolist()]


def slot_mode():
    return SLOT_MODE


def request_positive_counts(proposals):
    """called by the box head right after its sampler (training, positives-first fixed-length proposals): starts the counts'
    trip to the host when the mask head will want them"""
    if SLOT_MODE == "dynamic" and proposals and all(p.has_field("valid") for p in proposals):
        proposals[0]._positive_counts = PositiveCounts(proposals)


def slots_per_image(proposals, max_positives):
    """-> list of slot counts (one per image) for the fixed-length positives-first proposals"""
    if SLOT_MODE == "fixed":
        return [max_positives] * len(proposals)
    if SLOT_MODE != "dynamic":
        forced = [int(v) for v in SLOT_MODE.split(",")]
        return [min(max_positives, max(1, forced[i % len(forced)])) for i in range(len(proposals))]
    handle = getattr(proposals[0], "_positive_counts", None) if proposals else None
    if handle is None:
        handle = PositiveCounts(proposals)
    g = SLOT_GRANULE
    return [min(max_positives, max(g, -(-c // g) * g)) for c in handle.get()]


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
    first inputs; an int or one int per image) every image contributes exactly its first `max_positives` slots: the BoxLists
    are then VIEWS of the proposals' tensors (no indexing launches; do not write into them) and the indices a cached arange."""
    assert isinstance(boxes, (list, tuple)) and isinstance(boxes[0], BoxList)
    assert boxes[0].has_field("labels")
    out, inds = [], []
    for i, b in enumerate(boxes):
        if max_positives is None:
            sel = (b.get_field("labels") > 0).nonzero().squeeze(1)
            out.append(b[sel])
        else:
            n = min(max_positives[i] if isinstance(max_positives, (list, tuple)) else max_positives, len(b))
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
        self.last_slots = None      # slot counts of the last training forward (one per image): bench.py reports them

    def forward(self, features, proposals, targets=None):
        if self.training:
            all_proposals = proposals
            fixed = all(p.has_field("valid") for p in proposals)
            slots = slots_per_image(proposals, self.max_positives) if fixed else None
            self.last_slots = slots
            proposals, positive_inds = keep_only_positive_boxes(proposals, slots)
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
