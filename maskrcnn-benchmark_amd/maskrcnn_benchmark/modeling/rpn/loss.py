"""RPN loss (reference modeling/rpn/loss.py:21-158).

Per image: IoU(gt, anchors) -> Matcher (0.7 / 0.3, low-quality matches allowed) -> labels
{1 fg, 0 bg, -1 ignored: between thresholds or anchor outside the image} and box-regression
targets; sample 256 anchors per image (<= 50 % fg); objectness = BCE-with-logits (mean over the
sampled anchors), box = smooth-L1(beta 1/9) summed over sampled positives / #sampled.

Batched over images (ground truth padded to the largest count with a row mask) and free of
`nonzero`: sampled subsets are boolean masks and the losses are masked sums, so the values equal
the reference's indexed forms while nothing is synchronised with the host.
"""
import os

import torch
from torch.nn import functional as F

from maskrcnn_benchmark import _C
from maskrcnn_benchmark.modeling.matcher import Matcher
from maskrcnn_benchmark.modeling.utils import device_constant
from maskrcnn_benchmark.structures.boxlist_ops import box_iou_matrix

from ..balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from .utils import concat_box_prediction_layers


_FUSED = os.environ.get("DETOPS_RPN_LOSS", "fused") != "torch"   # A/B switch: the PyTorch composite on the GPU


# The RPN loss, the proposal selector (ground truth appended to the proposals) and the box head all pad the SAME targets of a
# step: the padded batch is kept for the duration of one detector forward (`begin_step` drops it — the detector calls it on
# entry, so a batch that is fed again, as a benchmark does, is padded again like a fresh one).  A hit needs the same BoxList
# objects with untouched tensors (identity + tensor version), so a stale entry can never be served.
_PADDED = {}


def begin_step():
    """Drop the shared padded-target batch.  GeneralizedRCNN.forward calls it on entry AND on exit (also when the forward
    raises).  A custom detector that drives RPNLossComputation / the box-head loss / RPNPostProcessor itself must do the
    same around each step: the cache is keyed by object identity + tensor versions, an in-place write through `.data`
    bumps no version and would otherwise be served stale."""
    _PADDED.clear()


def _targets_stamp(targets, device, fields):
    return (str(device), tuple(fields)) + tuple(
        (id(t), id(t.bbox), t.bbox._version) + tuple((id(t.get_field(f)), t.get_field(f)._version) for f in fields)
        for t in targets)


def pad_targets(targets, device, fields=()):
    """list[BoxList] -> (boxes [N,M,4] (padding rows = a far-away unit box), row_valid [N,M] bool,
    {field: [N,M]}) with M = max number of ground-truth boxes; host-side sizes only, no sync.
    The result is shared by the callers of one step (see `begin_step`): treat it as read-only."""
    fields = tuple(fields)
    stamp = _targets_stamp(targets, device, fields)
    hit = _PADDED.get(stamp)
    if hit is not None and all(a is b for a, b in zip(hit[0], targets)):
        return hit[1]
    out = _pad_targets(targets, device, fields)
    if len(_PADDED) >= 4:
        _PADDED.clear()
    _PADDED[stamp] = (list(targets), out)   # the references keep the ids of the stamp alive
    return out


_PADDING_ROWS = {}


def _padding_rows(n, kind, device):
    """n rows of padding — kind "box": the far-away unit box (-1e5, -1e5, -1e5 + 1, -1e5 + 1); kind "zero": int64 zeros;
    "true" / "false": bool rows of the validity mask — as
    a slice of a constant kept on the device (rows rounded up to a power of two >= 64: one constant serves every batch)."""
    cap = 64
    while cap < n:
        cap *= 2
    key = (cap, kind, str(device))
    rows = _PADDING_ROWS.get(key)
    if rows is None:
        if kind == "box":
            rows = device_constant([[-1e5, -1e5, -1e5 + 1, -1e5 + 1]] * cap, torch.float32, device)
        elif kind in ("true", "false"):
            rows = device_constant([kind == "true"] * cap, torch.bool, device)
        else:
            rows = device_constant([0] * cap, torch.int64, device)
        _PADDING_ROWS[key] = rows      # a handful of entries: (power of two, kind, device)
    return rows[:n]


def _pad_targets(targets, device, fields):
    """ONE concatenation per padded tensor (each image's rows followed by its padding rows, sliced from device-resident
    constants), the row-validity mask included: 2 + len(fields) launches, no host-to-device copy."""
    N = len(targets)
    lens = [len(t) for t in targets]
    M = max(max(lens), 1)
    parts = {f: [] for f in ("bbox",) + tuple(fields)}
    for t, m in zip(targets, lens):
        if m:
            parts["bbox"].append(t.convert("xyxy").bbox.to(device=device, dtype=torch.float32))
            for f in fields:
                parts[f].append(t.get_field(f).to(device=device, dtype=torch.int64))
        if m < M:
            parts["bbox"].append(_padding_rows(M - m, "box", device))
            for f in fields:
                parts[f].append(_padding_rows(M - m, "zero", device))
    boxes = torch.cat(parts["bbox"], dim=0).view(N, M, 4)
    # the validity mask from device-resident pieces as well (m true rows, M - m false rows per image): a constant keyed by
    # the per-image counts would be a fresh host build + upload for nearly every batch of a real data loader
    row_valid = torch.cat([_padding_rows(k, kind, device) for m in lens for k, kind in ((m, "true"), (M - m, "false")) if k],
                          dim=0).view(N, M)
    extra = {f: torch.cat(parts[f], dim=0).view(N, M) for f in fields}
    return boxes, row_valid, extra


def match_batched(matcher, gt_boxes, row_valid, boxes):
    """IoU + Matcher for a batch: gt_boxes [N,M,4], boxes [N,K,4] (or [K,4] shared) -> matched_idxs
    [N,K] int64 (>= 0 index into the image's gt rows, -1 / -2 as in Matcher)."""
    if _C.on_device(boxes):
        # fused IoU + Matcher kernel (csrc/targets.hip): the [N, M, K] quality matrix is never materialised
        return _C.match_boxes(gt_boxes, row_valid, boxes, matcher.high_threshold, matcher.low_threshold,
                              matcher.allow_low_quality_matches)
    if boxes.dim() == 2:
        boxes = boxes.unsqueeze(0).expand(gt_boxes.shape[0], -1, -1)
    iou = box_iou_matrix(gt_boxes, boxes)
    iou = torch.where(row_valid[:, :, None], iou, iou.new_full((), -1.0))
    return matcher(iou, row_valid)


def smooth_l1_elementwise(pred, target, beta):
    n = torch.abs(pred - target)
    return torch.where(n < beta, 0.5 * n * n / beta, n - 0.5 * beta)


class RPNLossComputation(object):
    def __init__(self, proposal_matcher, fg_bg_sampler, box_coder, generate_labels_func):
        self.proposal_matcher = proposal_matcher
        self.fg_bg_sampler = fg_bg_sampler
        self.box_coder = box_coder
        self.copied_fields = []
        self.generate_labels_func = generate_labels_func
        self.discard_cases = ["not_visibility", "between_thresholds"]

    # The per-level anchor tensors and their visibility masks are cached objects of the anchor generator (one per grid /
    # image size): their concatenations are cached here on the identity of the parts (bounded), instead of two cat
    # launches + a stack per step.
    def _cat_cached(self, parts, build):
        cache = self.__dict__.setdefault("_cat_cache", {})     # (subclasses have their own __init__)
        key = tuple(id(p) for p in parts)
        hit = cache.pop(key, None)
        if hit is None or any(a is not b for a, b in zip(hit[0], parts)):
            hit = (tuple(parts), build(parts))
            while len(cache) >= 64:
                cache.pop(next(iter(cache)))
        cache[key] = hit
        return hit[1]

    def _all_anchors(self, level_boxlists):
        return self._cat_cached([b.bbox for b in level_boxlists], lambda ps: torch.cat(ps, dim=0))

    def _visibility(self, anchors):
        parts = [b.get_field("visibility") for per_image in anchors for b in per_image]
        n = len(anchors)
        return self._cat_cached(parts, lambda ps: torch.stack([torch.cat(ps[i * len(ps) // n:(i + 1) * len(ps) // n], dim=0)
                                                               for i in range(n)], dim=0))

    def _match(self, anchors, targets):
        """-> (labels [N,A] float 1/0/-1, matched_idxs [N,A] int64, padded gt [N,M,4], all anchors [A,4])."""
        all_anchors = self._all_anchors(anchors[0])
        dev = all_anchors.device
        gt, row_valid, extra = pad_targets(targets, dev, self.copied_fields)
        matched = match_batched(self.proposal_matcher, gt, row_valid, all_anchors)
        if _C.on_device(matched) and _FUSED and self.generate_labels_func is generate_rpn_labels \
                and "between_thresholds" in self.discard_cases:
            # the masked assignments below in one launch
            vis = self._visibility(anchors) if "not_visibility" in self.discard_cases else None
            return _C.match_labels(matched, None, vis, torch.float32), matched, gt, all_anchors
        labels = self.generate_labels_func(matched, extra).to(torch.float32)
        labels = torch.where(matched == Matcher.BELOW_LOW_THRESHOLD, torch.zeros_like(labels), labels)
        if "not_visibility" in self.discard_cases:
            labels = torch.where(self._visibility(anchors), labels, labels.new_full((), -1.0))
        if "between_thresholds" in self.discard_cases:
            labels = torch.where(matched == Matcher.BETWEEN_THRESHOLDS, labels.new_full((), -1.0), labels)
        return labels, matched, gt, all_anchors

    def prepare_targets(self, anchors, targets):
        """anchors: list (image) of list (level) of BoxList.  -> labels [N,A] float (1/0/-1),
        regression_targets [N,A,4]."""
        labels, matched, gt, all_anchors = self._match(anchors, targets)
        matched_gt = torch.gather(gt, 1, matched.clamp(min=0)[:, :, None].expand(-1, -1, 4))
        regression_targets = self.box_coder.encode(matched_gt, all_anchors.unsqueeze(0))
        return labels, regression_targets

    def __call__(self, anchors, objectness, box_regression, targets):
        if _C.on_device(objectness[0]) and len(objectness) <= 8 and _FUSED:
            # one launch over the head outputs in their own layout (csrc/targets.hip::rpn_loss_kernel): no permute /
            # cat of the 5 levels, no [N, A, 4] regression targets, no masked reductions — and none of their
            # autograd mirrors
            labels, matched, gt, all_anchors = self._match(anchors, targets)
            pos, neg = self.fg_bg_sampler._masks(labels)
            # half-precision head outputs (autocast): the loss is evaluated in fp32 like the composite below
            objectness = [t.float() for t in objectness]
            box_regression = [t.float() for t in box_regression]
            return _C.rpn_loss(objectness, box_regression, all_anchors, matched, pos, neg, gt, 1.0 / 9,
                               self.box_coder.weights)
        labels, regression_targets = self.prepare_targets(anchors, targets)
        pos, neg = self.fg_bg_sampler._masks(labels)
        sampled = pos | neg
        num_sampled = sampled.sum().clamp(min=1).to(torch.float32)
        objectness, box_regression = concat_box_prediction_layers(objectness, box_regression, keep_batch=True)
        objectness = objectness.squeeze(-1).float()
        box_regression = box_regression.float()
        box_l = smooth_l1_elementwise(box_regression, regression_targets, beta=1.0 / 9).sum(dim=-1)
        box_loss = torch.where(pos, box_l, torch.zeros_like(box_l)).sum() / num_sampled
        bce = F.binary_cross_entropy_with_logits(objectness, labels.clamp(min=0), reduction="none")
        objectness_loss = torch.where(sampled, bce, torch.zeros_like(bce)).sum() / num_sampled
        return objectness_loss, box_loss


def generate_rpn_labels(matched_idxs, extra=None):
    return matched_idxs >= 0


def make_rpn_loss_evaluator(cfg, box_coder):
    R = cfg.MODEL.RPN
    matcher = Matcher(R.FG_IOU_THRESHOLD, R.BG_IOU_THRESHOLD, allow_low_quality_matches=True)
    sampler = BalancedPositiveNegativeSampler(R.BATCH_SIZE_PER_IMAGE, R.POSITIVE_FRACTION)
    return RPNLossComputation(matcher, sampler, box_coder, generate_rpn_labels)
