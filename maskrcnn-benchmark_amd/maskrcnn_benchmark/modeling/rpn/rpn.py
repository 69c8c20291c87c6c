"""RPN head + module (reference modeling/rpn/rpn.py:14-208)."""
import torch
import torch.nn.functional as F
from torch import nn

from maskrcnn_benchmark.layers.misc import conv_bias_act

from maskrcnn_benchmark.modeling import registry
from maskrcnn_benchmark.modeling.box_coder import BoxCoder

from .anchor_generator import make_anchor_generator
from .inference import make_rpn_postprocessor
from .loss import make_rpn_loss_evaluator


class RPNHeadConvRegressor(nn.Module):
    """classification + regression heads without the shared 3x3 conv (reference :14-45)."""

    def __init__(self, cfg, in_channels, num_anchors):
        super(RPNHeadConvRegressor, self).__init__()
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, kernel_size=1, stride=1)
        for m in (self.cls_logits, self.bbox_pred):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        assert isinstance(x, (list, tuple))
        return [self.cls_logits(y) for y in x], [self.bbox_pred(y) for y in x]


@registry.RPN_HEADS.register("SingleConvRPNHead")
class RPNHead(nn.Module):
    """3x3 conv + relu shared trunk, then 1x1 objectness (A) and 1x1 box deltas (4A) per level."""

    def __init__(self, cfg, in_channels, num_anchors):
        super(RPNHead, self).__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, kernel_size=1, stride=1)
        for m in (self.conv, self.cls_logits, self.bbox_pred):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        logits, bbox_reg = [], []
        for feature in x:
            t = conv_bias_act(self.conv, feature, relu=True)      # channels-last: conv, then ONE fused bias + ReLU pass
            # the proposal / loss kernels read the A and 4A-channel outputs in NCHW; under a channels-last pyramid these two
            # small tensors are the only ones converted (a no-op for NCHW features)
            logits.append(conv_bias_act(self.cls_logits, t).contiguous())
            bbox_reg.append(conv_bias_act(self.bbox_pred, t).contiguous())
        return logits, bbox_reg


class RPNModule(nn.Module):
    """features -> proposals (+ RPN losses in training)."""

    def __init__(self, cfg, in_channels):
        super(RPNModule, self).__init__()
        self.cfg = cfg.clone()
        anchor_generator = make_anchor_generator(cfg)
        head = registry.RPN_HEADS[cfg.MODEL.RPN.RPN_HEAD](cfg, in_channels,
                                                          anchor_generator.num_anchors_per_location()[0])
        rpn_box_coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.anchor_generator = anchor_generator
        self.head = head
        self.box_selector_train = make_rpn_postprocessor(cfg, rpn_box_coder, is_train=True)
        self.box_selector_test = make_rpn_postprocessor(cfg, rpn_box_coder, is_train=False)
        self.loss_evaluator = make_rpn_loss_evaluator(cfg, rpn_box_coder)

    def forward(self, images, features, targets=None):
        objectness, rpn_box_regression = self.head(features)
        anchors = self.anchor_generator(images, features)
        if self.training:
            return self._forward_train(anchors, objectness, rpn_box_regression, targets)
        return self._forward_test(anchors, objectness, rpn_box_regression)

    def _forward_train(self, anchors, objectness, rpn_box_regression, targets):
        if self.cfg.MODEL.RPN_ONLY:
            boxes = anchors  # proposals are not consumed; only the loss matters
        else:
            with torch.no_grad():  # end-to-end models do not backprop through the proposals
                boxes = self.box_selector_train(anchors, objectness, rpn_box_regression, targets)
        loss_objectness, loss_rpn_box_reg = self.loss_evaluator(anchors, objectness, rpn_box_regression, targets)
        return boxes, {"loss_objectness": loss_objectness, "loss_rpn_box_reg": loss_rpn_box_reg}

    def _forward_test(self, anchors, objectness, rpn_box_regression):
        boxes = self.box_selector_test(anchors, objectness, rpn_box_regression)
        if self.cfg.MODEL.RPN_ONLY:
            boxes = [b[b.get_field("objectness").sort(descending=True)[1]] for b in boxes]
        return boxes, {}


def build_rpn(cfg, in_channels):
    if cfg.MODEL.RETINANET_ON:
        from .retinanet.retinanet import build_retinanet
        return build_retinanet(cfg, in_channels)
    return RPNModule(cfg, in_channels)
