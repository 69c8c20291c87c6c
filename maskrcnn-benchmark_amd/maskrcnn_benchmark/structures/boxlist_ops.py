"""BoxList operations (reference structures/boxlist_ops.py:9-128)."""
import torch

from maskrcnn_benchmark.layers import nms as _box_nms

from .bounding_box import BoxList

TO_REMOVE = 1


def boxlist_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores"):
    """Greedy NMS on a BoxList (reference :9-31): keeps the `max_proposals` lowest kept indices,
    which are the best-scoring ones when the input is sorted by score."""
    if nms_thresh <= 0:
        return boxlist
    mode = boxlist.mode
    boxlist = boxlist.convert("xyxy")
    keep = _box_nms(boxlist.bbox, boxlist.get_field(score_field), nms_thresh)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep.to(boxlist.bbox.device)].convert(mode)


def remove_small_boxes(boxlist, min_size):
    """Keep boxes with both sides >= min_size (reference :34-48)."""
    wh = boxlist.convert("xywh").bbox[:, 2:]
    keep = ((wh[:, 0] >= min_size) & (wh[:, 1] >= min_size)).nonzero().squeeze(1)
    return boxlist[keep]


def box_iou_matrix(a, b):
    """IoU of every box of `a [M,4]` with every box of `b [K,4]` (xyxy, +1 pixel convention),
    reference :53-89.  Also accepts leading batch dimensions ([B,M,4] x [B,K,4] -> [B,M,K])."""
    area_a = (a[..., 2] - a[..., 0] + TO_REMOVE) * (a[..., 3] - a[..., 1] + TO_REMOVE)
    area_b = (b[..., 2] - b[..., 0] + TO_REMOVE) * (b[..., 3] - b[..., 1] + TO_REMOVE)
    lt = torch.max(a[..., :, None, :2], b[..., None, :, :2])
    rb = torch.min(a[..., :, None, 2:], b[..., None, :, 2:])
    wh = (rb - lt + TO_REMOVE).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[..., :, None] + area_b[..., None, :] - inter)


def boxlist_iou(boxlist1, boxlist2):
    if boxlist1.size != boxlist2.size:
        raise RuntimeError("boxlists should have same image size, got {}, {}".format(boxlist1, boxlist2))
    return box_iou_matrix(boxlist1.convert("xyxy").bbox, boxlist2.convert("xyxy").bbox)


def _cat(tensors, dim=0):
    assert isinstance(tensors, (list, tuple))
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)


def cat_boxlist(bboxes):
    """Concatenate BoxLists of one image (same size, mode and fields), reference :104-128."""
    assert isinstance(bboxes, (list, tuple)) and all(isinstance(b, BoxList) for b in bboxes)
    size, mode, fields = bboxes[0].size, bboxes[0].mode, set(bboxes[0].fields())
    assert all(b.size == size for b in bboxes)
    assert all(b.mode == mode for b in bboxes)
    assert all(set(b.fields()) == fields for b in bboxes)
    out = BoxList(_cat([b.bbox for b in bboxes], dim=0), size, mode)
    for f in fields:
        out.add_field(f, _cat([b.get_field(f) for b in bboxes], dim=0))
    return out
