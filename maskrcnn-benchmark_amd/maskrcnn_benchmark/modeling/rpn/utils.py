"""Layout helpers for head outputs (reference modeling/rpn/utils.py:9-45): [N, A*C, H, W] ->
[N, H*W*A, C], i.e. anchors ordered (y, x, a) like the anchor grid."""
import torch

from ..utils import cat


def permute_and_flatten(layer, N, A, C, H, W):
    if layer.dim() == 4 and not layer.is_contiguous() and layer.is_contiguous(memory_format=torch.channels_last):
        # a channels-last head output IS [N, H, W, A*C] in memory: the reference's (N, H, W, A, C) order is a view of it
        return layer.permute(0, 2, 3, 1).reshape(N, -1, C)
    return layer.view(N, -1, C, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, C)


def concat_box_prediction_layers(box_cls, box_regression, keep_batch=False):
    """per-level lists -> ([N*sum(HWA), C], [N*sum(HWA), 4]) (or with the batch dim kept)."""
    cls_flat, reg_flat = [], []
    C = 1
    for c, r in zip(box_cls, box_regression):
        N, AxC, H, W = c.shape
        A = r.shape[1] // 4
        C = AxC // A
        cls_flat.append(permute_and_flatten(c, N, A, C, H, W))
        reg_flat.append(permute_and_flatten(r, N, A, 4, H, W))
    box_cls = cat(cls_flat, dim=1)
    box_regression = cat(reg_flat, dim=1)
    if keep_batch:
        return box_cls, box_regression
    return box_cls.reshape(-1, C), box_regression.reshape(-1, 4)

