"""ROIBoxHead (reference roi_heads/box_head/box_head.py:11-71)."""
import torch

from .inference import make_roi_box_post_processor
from .loss import make_roi_box_loss_evaluator
from .roi_box_feature_extractors import make_roi_box_feature_extractor
from .roi_box_predictors import make_roi_box_predictor


class ROIBoxHead(torch.nn.Module):
    def __init__(self, cfg, in_channels):
        super(ROIBoxHead, self).__init__()
        self.feature_extractor = make_roi_box_feature_extractor(cfg, in_channels)
        self.predictor = make_roi_box_predictor(cfg, self.feature_extractor.out_channels)
        self.post_processor = make_roi_box_post_processor(cfg)
        self.loss_evaluator = make_roi_box_loss_evaluator(cfg)
        self.mask_follows = bool(cfg.MODEL.MASK_ON)

    def forward(self, features, proposals, targets=None):
        """-> (x: pooled features, proposals: sampled (train) / detections (eval), losses)."""
        if self.training:
            with torch.no_grad():
                proposals = self.loss_evaluator.subsample(proposals, targets)
                if self.mask_follows:
                    # the mask head runs on the positives only: their per-image counts start their (asynchronous) trip to the
                    # host here, a whole box-head forward before the mask head asks for them (mask_head.PositiveCounts)
                    from ..mask_head.mask_head import request_positive_counts
                    request_positive_counts(proposals)
        x = self.feature_extractor(features, proposals)
        class_logits, box_regression = self.predictor(x)
        if not self.training:
            return x, self.post_processor((class_logits, box_regression), proposals), {}
        loss_classifier, loss_box_reg = self.loss_evaluator([class_logits], [box_regression])
        return x, proposals, dict(loss_classifier=loss_classifier, loss_box_reg=loss_box_reg)


def build_roi_box_head(cfg, in_channels):
    return ROIBoxHead(cfg, in_channels)
