"""Mask post-processing for inference (reference roi_heads/mask_head/inference.py:12-55):
sigmoid, pick each detection's class channel, attach as field "mask" [n,1,M,M]."""
import torch
from torch import nn


class MaskPostProcessor(nn.Module):
    def __init__(self, masker=None):
        super(MaskPostProcessor, self).__init__()
        self.masker = masker

    def forward(self, x, boxes):
        prob = x.sigmoid()
        labels = torch.cat([b.get_field("labels") for b in boxes])
        prob = prob[torch.arange(prob.shape[0], device=labels.device), labels][:, None]
        counts = [len(b) for b in boxes]
        results = []
        for p, b in zip(prob.split(counts, dim=0), boxes):
            out = b.copy_with_fields(b.fields())
            out.add_field("mask", p)
            results.append(out)
        return results


def make_roi_mask_post_processor(cfg):
    return MaskPostProcessor(None)
