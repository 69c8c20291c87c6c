"""GeneralizedRCNN (reference modeling/detector/generalized_rcnn.py:16-65): backbone -> RPN ->
ROI heads.  `model(images, targets)` returns a dict of losses in training and a list of BoxLists
(detections) in eval mode."""
from torch import nn

from maskrcnn_benchmark.structures.image_list import to_image_list

from ..backbone import build_backbone
from ..roi_heads.roi_heads import build_roi_heads
from ..rpn.loss import begin_step
from ..rpn.rpn import build_rpn


class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super(GeneralizedRCNN, self).__init__()
        self.backbone = build_backbone(cfg)
        self.rpn = build_rpn(cfg, self.backbone.out_channels)
        self.roi_heads = build_roi_heads(cfg, self.backbone.out_channels)

    def forward(self, images, targets=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        begin_step()   # the padded-target batch is shared by the callers of ONE forward, never across forwards
        try:
            images = to_image_list(images)
            features = self.backbone(images.tensors)
            proposals, proposal_losses = self.rpn(images, features, targets)
            if self.roi_heads:
                x, result, detector_losses = self.roi_heads(features, proposals, targets)
            else:  # RPN-only models (RetinaNet) have no ROI heads
                x, result, detector_losses = features, proposals, {}
        finally:
            begin_step()   # ... and never beyond it: nothing of this batch stays referenced after the forward (or its exception)
        if self.training:
            losses = {}
            losses.update(detector_losses)
            losses.update(proposal_losses)
            return losses
        return result
