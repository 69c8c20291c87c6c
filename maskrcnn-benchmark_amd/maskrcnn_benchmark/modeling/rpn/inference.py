"""RPN proposal selection (reference modeling/rpn/inference.py:15-206).

Semantics kept from the reference: per feature level take the `pre_nms_top_n` highest-objectness
anchors of each image, decode, clip to the image, drop boxes smaller than `min_size`, NMS at
`nms_thresh`, keep at most `post_nms_top_n` per (image, level); then across levels keep the
`fpn_post_nms_top_n` best — over the whole batch in training when FPN_POST_NMS_PER_BATCH is set
(:161-172), per image otherwise; in training the ground-truth boxes are appended (:51-71).

Design (not the reference's per-image Python loop with ~24 host syncs):
every step works on fixed-shape `[N, K]` tensors with a validity mask; the 5 levels x N images
NMS problems run as ONE segmented launch of the HIP kernel that returns a dense keep mask
(`_C.nms_batched_mask`).  Nothing is read back to the host in training: proposals leave as padded
BoxLists with a boolean "valid" field.  In eval mode the lists are compacted (one sync) so callers
see exactly the reference's variable-length BoxLists.
"""
import torch

from maskrcnn_benchmark.modeling.utils import device_constant

from maskrcnn_benchmark import _C
from maskrcnn_benchmark.modeling.box_coder import BoxCoder
from maskrcnn_benchmark.structures.bounding_box import BoxList

from .utils import permute_and_flatten


class RPNPostProcessor(torch.nn.Module):
    def __init__(self, pre_nms_top_n, post_nms_top_n, nms_thresh, min_size, box_coder=None,
                 fpn_post_nms_top_n=None, fpn_post_nms_per_batch=True):
        super(RPNPostProcessor, self).__init__()
        self.pre_nms_top_n = pre_nms_top_n
        self.post_nms_top_n = post_nms_top_n
        self.nms_thresh = nms_thresh
        self.min_size = min_size
        self.box_coder = box_coder if box_coder is not None else BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.fpn_post_nms_top_n = post_nms_top_n if fpn_post_nms_top_n is None else fpn_post_nms_top_n
        self.fpn_post_nms_per_batch = fpn_post_nms_per_batch
        self._seg_cache = {}
        self.fused_decode = True  # device tensors: one _C.rpn_decode launch per level instead of the ATen composition

    # ------------------------------------------------------------------ per level (batched over images)
    def _level_candidates(self, level_anchors, objectness, box_regression, image_sizes):
        """-> boxes [N,k,4] (decoded, clipped), scores [N,k] (sigmoid), ok [N,k] (min_size test)."""
        N, A, H, W = objectness.shape
        scores = permute_and_flatten(objectness, N, A, 1, H, W).view(N, -1).sigmoid()
        deltas = permute_and_flatten(box_regression, N, A, 4, H, W)
        k = min(self.pre_nms_top_n, A * H * W)
        scores, idx = scores.topk(k, dim=1, sorted=True)
        deltas = torch.gather(deltas, 1, idx[:, :, None].expand(N, k, 4))
        anchors = level_anchors[idx.reshape(-1)]
        boxes = self.box_coder.decode(deltas.reshape(-1, 4), anchors).view(N, k, 4)
        # clip_to_image(remove_empty=False): per image (w-1, h-1) upper bounds
        hi = device_constant([[w - 1, h - 1, w - 1, h - 1] for (h, w) in image_sizes], boxes.dtype, boxes.device)[:, None, :]
        boxes = torch.minimum(boxes.clamp(min=0), hi)
        ws = boxes[..., 2] - boxes[..., 0] + 1
        hs = boxes[..., 3] - boxes[..., 1] + 1
        ok = (ws >= self.min_size) & (hs >= self.min_size)
        return boxes, scores, ok

    def _segments(self, ks, N, device):
        """int32 [S+1] offsets of the (level, image) segments in the level-major flat layout."""
        key = (tuple(ks), N, str(device))
        if key not in self._seg_cache:
            lens = [k for k in ks for _ in range(N)]
            offs = [0]
            for v in lens:
                offs.append(offs[-1] + v)
            self._seg_cache[key] = torch.tensor(offs, dtype=torch.int32).to(device)
        return self._seg_cache[key]

    def select(self, anchors_per_level, objectness, box_regression, image_sizes, training):
        """-> boxes [N,K,4], scores [N,K], valid [N,K] with K = sum_l k_l (level-major per image).
        Runs in fp32 with autocast off (the reference marks nms as an fp32 function, layers/nms.py:8)."""
        with torch.autocast(device_type=objectness[0].device.type, enabled=False):
            return self._select(anchors_per_level, [o.float() for o in objectness],
                                [r.float() for r in box_regression], image_sizes, training)

    def _candidates_fused(self, anchors_per_level, objectness, box_regression, image_sizes):
        """The device path of `_level_candidates` + the NMS-input assembly below: per level a sigmoid, a top-k and ONE
        `_C.rpn_decode` launch that reads the head output in its own layout and writes straight into the image-major
        result and the level-major NMS input (no permuted copy of the regression, no per-level concatenation)."""
        N = objectness[0].shape[0]
        dev = objectness[0].device
        ks = [min(self.pre_nms_top_n, o.shape[1] * o.shape[2] * o.shape[3]) for o in objectness]
        K = sum(ks)
        boxes = torch.empty((N, K, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((N, K), dtype=torch.float32, device=dev)
        flat_boxes = torch.empty((N * K, 4), dtype=torch.float32, device=dev)
        flat_scores = torch.empty((N * K,), dtype=torch.float32, device=dev)
        flat_ok = torch.empty((N * K,), dtype=torch.uint8, device=dev)
        image_hw = device_constant([[h, w] for (h, w) in image_sizes], torch.float32, dev)
        col = 0
        for a, o, r, k in zip(anchors_per_level, objectness, box_regression, ks):
            A, H, W = o.shape[1:]
            s, idx = permute_and_flatten(o, N, A, 1, H, W).view(N, -1).sigmoid().topk(k, dim=1, sorted=True)
            _C.rpn_decode(r, idx, s, a, image_hw, self.box_coder.weights, self.box_coder.bbox_xform_clip, self.min_size,
                          boxes, scores, col, flat_boxes, flat_scores, flat_ok, N * col)
            col += k
        return ks, boxes, scores, flat_boxes, flat_scores, flat_ok.view(torch.bool)

    def _select(self, anchors_per_level, objectness, box_regression, image_sizes, training):
        N = objectness[0].shape[0]
        dev = objectness[0].device
        if _C.on_device(objectness[0]) and self.fused_decode:
            ks, boxes, scores, flat_boxes, flat_scores, flat_ok = self._candidates_fused(
                anchors_per_level, objectness, box_regression, image_sizes)
        else:
            boxes, scores, oks = [], [], []
            for a, o, r in zip(anchors_per_level, objectness, box_regression):
                b, s, ok = self._level_candidates(a, o, r, image_sizes)
                boxes.append(b)
                scores.append(s)
                oks.append(ok)
            ks = [b.shape[1] for b in boxes]
            # one segmented NMS over all (level, image) problems
            flat_boxes = torch.cat([b.reshape(-1, 4) for b in boxes], dim=0)
            flat_scores = torch.cat([s.reshape(-1) for s in scores], dim=0)
            flat_ok = torch.cat([k.reshape(-1) for k in oks], dim=0)
            if self.min_size > 0:
                # removed boxes must not take part in NMS: move them far away with the lowest score
                far = device_constant([-1e6, -1e6, -1e6 + 1, -1e6 + 1], flat_boxes.dtype, flat_boxes.device)
                flat_boxes = torch.where(flat_ok[:, None], flat_boxes, far)
                flat_scores = torch.where(flat_ok, flat_scores, flat_scores.new_full((), -1.0))
            boxes = torch.cat(boxes, dim=1)
            scores = torch.cat(scores, dim=1)
        keep, _ = _C.nms_batched_mask(flat_boxes, flat_scores, self._segments(ks, N, dev), max(ks),
                                      self.nms_thresh)
        keep = keep & flat_ok
        valid, off = [], 0
        for k in ks:
            v = keep[off:off + N * k].view(N, k)
            off += N * k
            if self.post_nms_top_n < k:  # keep the first post_nms_top_n survivors (score order)
                v = v & (v.cumsum(dim=1) <= self.post_nms_top_n)
            valid.append(v)
        valid = torch.cat(valid, dim=1)
        if len(ks) > 1:
            valid = self._select_over_all_levels(scores, valid, training)
        return boxes, scores, valid

    def _select_over_all_levels(self, scores, valid, training):
        N, K = scores.shape
        masked = torch.where(valid, scores, scores.new_full((), -1.0))
        if training and self.fpn_post_nms_per_batch:
            k = min(self.fpn_post_nms_top_n, N * K)
            top, idx = masked.reshape(-1).topk(k, sorted=False)
            chosen = torch.zeros(N * K, dtype=torch.bool, device=scores.device)
            chosen[idx] = top >= 0
            return chosen.view(N, K)
        k = min(self.fpn_post_nms_top_n, K)
        top, idx = masked.topk(k, dim=1, sorted=False)
        chosen = torch.zeros_like(valid)
        chosen.scatter_(1, idx, top >= 0)
        return chosen

    # ------------------------------------------------------------------ public entry
    def forward(self, anchors, objectness, box_regression, targets=None):
        """anchors: list (image) of list (level) of BoxList, as AnchorGenerator.forward returns.
        Returns one BoxList per image with fields "objectness" and (training) "valid"."""
        image_sizes = [(a[0].size[1], a[0].size[0]) for a in anchors]  # (h, w)
        per_level = [b.bbox for b in anchors[0]]
        boxes, scores, valid = self.select(per_level, objectness, box_regression, image_sizes, self.training)
        out = []
        if self.training and targets is not None and _C.on_device(boxes):
            # add_gt_proposals for the whole batch: the padded ground truth (rows beyond an image's count are far-away
            # boxes flagged invalid) joins every image's candidates with three concatenations; the lists handed to the
            # box head are rows of these tensors and say so (`batch_rows`), so its sampler takes the batch as it is
            from .loss import pad_targets
            gt, row_valid, _ = pad_targets(targets, boxes.device)
            batch = {"boxes": torch.cat([boxes, gt.to(boxes.dtype)], dim=1),
                     "objectness": torch.cat([scores, scores.new_ones(row_valid.shape)], dim=1),
                     "valid": torch.cat([valid, row_valid], dim=1)}
            for i, (h, w) in enumerate(image_sizes):
                bl = BoxList(batch["boxes"][i], (w, h), mode="xyxy")
                bl.add_field("objectness", batch["objectness"][i])
                bl.add_field("valid", batch["valid"][i])
                bl.batch_rows = (batch, i)
                out.append(bl)
            return out
        for i, (h, w) in enumerate(image_sizes):
            if self.training:
                b, s, v = boxes[i], scores[i], valid[i]
                if targets is not None:  # add_gt_proposals
                    gt = targets[i].convert("xyxy").bbox.to(b.dtype)
                    b = torch.cat([b, gt], dim=0)
                    s = torch.cat([s, s.new_ones(len(gt))], dim=0)
                    v = torch.cat([v, v.new_ones(len(gt))], dim=0)
                bl = BoxList(b, (w, h), mode="xyxy")
                bl.add_field("objectness", s)
                bl.add_field("valid", v)
            else:
                sel = valid[i].nonzero().squeeze(1)
                order = scores[i][sel].argsort(descending=True)
                sel = sel[order]
                bl = BoxList(boxes[i][sel], (w, h), mode="xyxy")
                bl.add_field("objectness", scores[i][sel])
            out.append(bl)
        return out


def make_rpn_postprocessor(config, rpn_box_coder, is_train):
    R = config.MODEL.RPN
    return RPNPostProcessor(
        pre_nms_top_n=R.PRE_NMS_TOP_N_TRAIN if is_train else R.PRE_NMS_TOP_N_TEST,
        post_nms_top_n=R.POST_NMS_TOP_N_TRAIN if is_train else R.POST_NMS_TOP_N_TEST,
        nms_thresh=R.NMS_THRESH,
        min_size=R.MIN_SIZE,
        box_coder=rpn_box_coder,
        fpn_post_nms_top_n=R.FPN_POST_NMS_TOP_N_TRAIN if is_train else R.FPN_POST_NMS_TOP_N_TEST,
        fpn_post_nms_per_batch=R.FPN_POST_NMS_PER_BATCH)
