"""RetinaNet loss (reference modeling/rpn/retinanet/loss.py:18-107).

Every anchor is labelled with its matched ground-truth class (IoU >= 0.5), 0 (background, < 0.4)
or -1 (ignored, in between); classification = SigmoidFocalLoss summed over all anchors x classes
/ (#positives + N); regression = smooth-L1 (beta 0.11) summed over positives / max(1, 4 * #pos).
The focal term runs in the hand-written HIP kernel (fused sum forward, scalar-gradient backward);
positives are handled with masks instead of `nonzero`."""
import torch

from maskrcnn_benchmark.layers import SigmoidFocalLoss
from maskrcnn_benchmark.modeling.matcher import Matcher
from maskrcnn_benchmark.modeling.rpn.loss import RPNLossComputation, smooth_l1_elementwise

from ..utils import concat_box_prediction_layers


class RetinaNetLossComputation(RPNLossComputation):
    def __init__(self, proposal_matcher, box_coder, generate_labels_func, sigmoid_focal_loss,
                 bbox_reg_beta=0.11, regress_norm=1.0):
        self.proposal_matcher = proposal_matcher
        self.box_coder = box_coder
        self.box_cls_loss_func = sigmoid_focal_loss
        self.bbox_reg_beta = bbox_reg_beta
        self.copied_fields = ["labels"]
        self.generate_labels_func = generate_labels_func
        self.discard_cases = ["between_thresholds"]
        self.regress_norm = regress_norm

    def __call__(self, anchors, box_cls, box_regression, targets):
        labels, regression_targets = self.prepare_targets(anchors, targets)  # [N,A], [N,A,4]
        N = labels.shape[0]
        box_cls, box_regression = concat_box_prediction_layers(box_cls, box_regression)
        labels = labels.reshape(-1)
        regression_targets = regression_targets.reshape(-1, 4)
        pos = labels > 0
        num_pos = pos.sum().to(torch.float32)
        l1 = smooth_l1_elementwise(box_regression.float(), regression_targets, self.bbox_reg_beta).sum(dim=1)
        reg_loss = torch.where(pos, l1, torch.zeros_like(l1)).sum() / (num_pos * self.regress_norm).clamp(min=1)
        cls_loss = self.box_cls_loss_func(box_cls, labels.to(torch.int32)) / (num_pos + N)
        return cls_loss, reg_loss


def generate_retinanet_labels(matched_idxs, extra):
    return torch.gather(extra["labels"], 1, matched_idxs.clamp(min=0))


def make_retinanet_loss_evaluator(cfg, box_coder):
    R = cfg.MODEL.RETINANET
    matcher = Matcher(R.FG_IOU_THRESHOLD, R.BG_IOU_THRESHOLD, allow_low_quality_matches=True)
    return RetinaNetLossComputation(matcher, box_coder, generate_retinanet_labels,
                                    SigmoidFocalLoss(R.LOSS_GAMMA, R.LOSS_ALPHA),
                                    bbox_reg_beta=R.BBOX_REG_BETA, regress_norm=R.BBOX_REG_WEIGHT)
