"""Box-head post-processing for inference (reference roi_heads/box_head/inference.py:13-172):
softmax scores, decode per-class boxes, clip, score threshold, per-class NMS, top detections."""
import torch

from maskrcnn_benchmark.modeling.utils import device_constant
import torch.nn.functional as F
from torch import nn

from maskrcnn_benchmark import _C
from maskrcnn_benchmark.modeling.box_coder import BoxCoder
from maskrcnn_benchmark.structures.bounding_box import BoxList


class PostProcessor(nn.Module):
    def __init__(self, score_thresh=0.05, nms=0.5, detections_per_img=100, box_coder=None,
                 cls_agnostic_bbox_reg=False, bbox_aug_enabled=False):
        super(PostProcessor, self).__init__()
        self.score_thresh = score_thresh
        self.nms = nms
        self.detections_per_img = detections_per_img
        self.box_coder = BoxCoder(weights=(10., 10., 5., 5.)) if box_coder is None else box_coder
        self.cls_agnostic_bbox_reg = cls_agnostic_bbox_reg
        self.bbox_aug_enabled = bbox_aug_enabled

    def forward(self, x, boxes):
        class_logits, box_regression = x
        class_prob = F.softmax(class_logits.float(), -1)
        image_shapes = [b.size for b in boxes]
        counts = [len(b) for b in boxes]
        concat_boxes = torch.cat([b.bbox for b in boxes], dim=0)
        if self.cls_agnostic_bbox_reg:
            box_regression = box_regression[:, -4:]
        decoded = self.box_coder.decode(box_regression.float().view(sum(counts), -1), concat_boxes)
        if self.cls_agnostic_bbox_reg:
            decoded = decoded.repeat(1, class_prob.shape[1])
        results = []
        for prob, b, shape in zip(class_prob.split(counts, dim=0), decoded.split(counts, dim=0), image_shapes):
            results.append(self.filter_results(b, prob, shape))
        return results

    def filter_results(self, boxes, scores, image_shape):
        """boxes [n, 4*C], scores [n, C] of one image -> BoxList(scores, labels).  The per-class
        NMS problems (classes 1..C-1) run as ONE segmented launch."""
        n, C = scores.shape
        W, H = image_shape
        if n == 0:  # no proposals: an empty result like the reference's (inference.py:108-149 on empty input)
            out = BoxList(boxes.new_zeros((0, 4)), image_shape, mode="xyxy")
            out.add_field("scores", scores.new_zeros((0,)))
            out.add_field("labels", torch.zeros((0,), dtype=torch.int64, device=scores.device))
            return out
        boxes = boxes.view(n, C, 4)
        hi = device_constant([W - 1, H - 1, W - 1, H - 1], boxes.dtype, boxes.device)
        boxes = torch.minimum(boxes.clamp(min=0), hi)
        cand = scores > self.score_thresh
        cand[:, 0] = False
        # class-major flat layout: segment c holds the n boxes of class c
        cb = boxes.permute(1, 0, 2).reshape(-1, 4)
        cs = scores.t().reshape(-1)
        cm = cand.t().reshape(-1)
        far = device_constant([-1e6, -1e6, -1e6 + 1, -1e6 + 1], cb.dtype, cb.device)
        cb_n = torch.where(cm[:, None], cb, far)
        cs_n = torch.where(cm, cs, cs.new_full((), -1.0))
        seg = torch.arange(0, (C + 1) * n, n, dtype=torch.int32, device=cb.device)
        keep, _ = _C.nms_batched_mask(cb_n, cs_n, seg, n, self.nms)
        keep = keep & cm
        sel = keep.nonzero().squeeze(1)
        det_scores = cs[sel]
        det_labels = sel // n
        det_boxes = cb[sel]
        if 0 < self.detections_per_img < sel.numel():
            # the reference thresholds at the k-th value (inference.py:139-146): ties at the threshold all stay
            thr = det_scores.topk(self.detections_per_img).values[-1]
            top = (det_scores >= thr).nonzero().squeeze(1)
            det_scores, det_labels, det_boxes = det_scores[top], det_labels[top], det_boxes[top]
        out = BoxList(det_boxes, image_shape, mode="xyxy")
        out.add_field("scores", det_scores)
        out.add_field("labels", det_labels)
        return out


def make_roi_box_post_processor(cfg):
    H = cfg.MODEL.ROI_HEADS
    return PostProcessor(H.SCORE_THRESH, H.NMS, H.DETECTIONS_PER_IMG, BoxCoder(weights=H.BBOX_REG_WEIGHTS),
                         cfg.MODEL.CLS_AGNOSTIC_BBOX_REG, False)
