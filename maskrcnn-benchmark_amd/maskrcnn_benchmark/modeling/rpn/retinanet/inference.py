"""RetinaNet inference post-processing (reference modeling/rpn/retinanet/inference.py:14-194):
per level keep (anchor, class) pairs with score > INFERENCE_TH (at most PRE_NMS_TOP_N), decode and
clip; over all levels run per-class NMS (NMS_TH) and keep the DETECTIONS_PER_IMG best.  The per-class
NMS problems run as one segmented launch of the HIP kernel; eval-only code, host syncs allowed."""
import torch

from maskrcnn_benchmark.modeling.utils import device_constant

from maskrcnn_benchmark import _C
from maskrcnn_benchmark.modeling.box_coder import BoxCoder
from maskrcnn_benchmark.structures.bounding_box import BoxList

from ..utils import permute_and_flatten


class RetinaNetPostProcessor(torch.nn.Module):
    def __init__(self, pre_nms_thresh, pre_nms_top_n, nms_thresh, fpn_post_nms_top_n, min_size, num_classes,
                 box_coder=None):
        super(RetinaNetPostProcessor, self).__init__()
        self.pre_nms_thresh = pre_nms_thresh
        self.pre_nms_top_n = pre_nms_top_n
        self.nms_thresh = nms_thresh
        self.fpn_post_nms_top_n = fpn_post_nms_top_n
        self.min_size = min_size
        self.num_classes = num_classes
        self.box_coder = BoxCoder(weights=(10., 10., 5., 5.)) if box_coder is None else box_coder

    def _level(self, level_anchors, box_cls, box_regression, image_sizes):
        N, _, H, W = box_cls.shape
        A = box_regression.size(1) // 4
        C = box_cls.size(1) // A
        scores = permute_and_flatten(box_cls, N, A, C, H, W).sigmoid().reshape(N, -1)
        deltas = permute_and_flatten(box_regression, N, A, 4, H, W)
        k = min(self.pre_nms_top_n, scores.shape[1])
        top, idx = scores.topk(k, dim=1)
        loc, cls = idx // C, idx % C + 1
        boxes = self.box_coder.decode(torch.gather(deltas, 1, loc[:, :, None].expand(N, k, 4)).reshape(-1, 4).float(),
                                      level_anchors[loc.reshape(-1)]).view(N, k, 4)
        hi = device_constant([[w - 1, h - 1, w - 1, h - 1] for (h, w) in image_sizes], boxes.dtype, boxes.device)[:, None, :]
        boxes = torch.minimum(boxes.clamp(min=0), hi)
        ok = (top > self.pre_nms_thresh)
        ok &= (boxes[..., 2] - boxes[..., 0] + 1 >= self.min_size) & (boxes[..., 3] - boxes[..., 1] + 1 >= self.min_size)
        return boxes, top, cls, ok

    def forward(self, anchors, box_cls, box_regression, targets=None):
        image_sizes = [(a[0].size[1], a[0].size[0]) for a in anchors]
        parts = [self._level(b.bbox, c, r, image_sizes) for b, c, r in zip(anchors[0], box_cls, box_regression)]
        boxes, scores, labels, ok = (torch.cat([p[j] for p in parts], dim=1) for j in range(4))
        results = []
        for i, (h, w) in enumerate(image_sizes):
            sel = ok[i].nonzero().squeeze(1)
            b, s, l = boxes[i][sel], scores[i][sel], labels[i][sel]
            # per-class NMS in one launch: sort by class, segments = class runs
            order = torch.argsort(l, stable=True)
            b, s, l = b[order], s[order], l[order]
            counts = torch.bincount(l, minlength=self.num_classes)
            seg = torch.zeros(self.num_classes + 1, dtype=torch.int32, device=b.device)
            seg[1:] = counts.cumsum(0).to(torch.int32)
            max_n = int(counts.max().item()) if l.numel() else 0
            if max_n > 0:
                keep, _ = _C.nms_batched_mask(b, s, seg, max_n, self.nms_thresh)
                b, s, l = b[keep], s[keep], l[keep]
            if 0 < self.fpn_post_nms_top_n < s.numel():
                # k-th value threshold like the reference (inference.py:160-168): ties at the threshold all stay
                thr = s.topk(self.fpn_post_nms_top_n).values[-1]
                top = (s >= thr).nonzero().squeeze(1)
                b, s, l = b[top], s[top], l[top]
            out = BoxList(b, (w, h), mode="xyxy")
            out.add_field("scores", s)
            out.add_field("labels", l)
            results.append(out)
        return results


def make_retinanet_postprocessor(config, rpn_box_coder, is_train):
    R = config.MODEL.RETINANET
    return RetinaNetPostProcessor(pre_nms_thresh=R.INFERENCE_TH, pre_nms_top_n=R.PRE_NMS_TOP_N, nms_thresh=R.NMS_TH,
                                  fpn_post_nms_top_n=config.TEST.DETECTIONS_PER_IMG, min_size=0,
                                  num_classes=R.NUM_CLASSES, box_coder=rpn_box_coder)
