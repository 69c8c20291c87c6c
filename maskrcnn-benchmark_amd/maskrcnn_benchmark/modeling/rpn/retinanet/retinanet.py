"""RetinaNet head + module (reference modeling/rpn/retinanet/retinanet.py:13-153)."""
import math

import torch
from torch import nn

from maskrcnn_benchmark.layers.misc import conv_bias_act
from maskrcnn_benchmark.modeling.box_coder import BoxCoder

from ..anchor_generator import make_anchor_generator_retinanet
from .inference import make_retinanet_postprocessor
from .loss import make_retinanet_loss_evaluator


class RetinaNetHead(nn.Module):
    """Two 4-conv towers shared across P3..P7; class logits (A*(C-1)) start at the prior
    probability PRIOR_PROB (bias = -log((1-p)/p))."""

    def __init__(self, cfg, in_channels):
        super(RetinaNetHead, self).__init__()
        R = cfg.MODEL.RETINANET
        num_classes = R.NUM_CLASSES - 1
        num_anchors = len(R.ASPECT_RATIOS) * R.SCALES_PER_OCTAVE
        cls_tower, bbox_tower = [], []
        for _ in range(R.NUM_CONVS):
            cls_tower += [nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1), nn.ReLU()]
            bbox_tower += [nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1), nn.ReLU()]
        self.add_module("cls_tower", nn.Sequential(*cls_tower))
        self.add_module("bbox_tower", nn.Sequential(*bbox_tower))
        self.cls_logits = nn.Conv2d(in_channels, num_anchors * num_classes, kernel_size=3, stride=1, padding=1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, kernel_size=3, stride=1, padding=1)
        for modules in (self.cls_tower, self.bbox_tower, self.cls_logits, self.bbox_pred):
            for m in modules.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.normal_(m.weight, std=0.01)
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - R.PRIOR_PROB) / R.PRIOR_PROB))

    @staticmethod
    def _tower(tower, f):
        # (conv, ReLU) pairs: on a channels-last activation each pair is the convolution without its bias + ONE fused
        # bias + ReLU pass, with the bias gradient reduced inside its backward (layers/misc.py::conv_bias_act)
        mods = list(tower)
        i = 0
        while i < len(mods):
            if isinstance(mods[i], nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                f = conv_bias_act(mods[i], f, relu=True)
                i += 2
            else:
                f = mods[i](f)
                i += 1
        return f

    def forward(self, x):
        return ([conv_bias_act(self.cls_logits, self._tower(self.cls_tower, f)) for f in x],
                [conv_bias_act(self.bbox_pred, self._tower(self.bbox_tower, f)) for f in x])


class RetinaNetModule(nn.Module):
    def __init__(self, cfg, in_channels):
        super(RetinaNetModule, self).__init__()
        self.cfg = cfg.clone()
        self.anchor_generator = make_anchor_generator_retinanet(cfg)
        self.head = RetinaNetHead(cfg, in_channels)
        box_coder = BoxCoder(weights=(10., 10., 5., 5.))
        self.box_selector_test = make_retinanet_postprocessor(cfg, box_coder, is_train=False)
        self.loss_evaluator = make_retinanet_loss_evaluator(cfg, box_coder)

    def forward(self, images, features, targets=None):
        box_cls, box_regression = self.head(features)
        anchors = self.anchor_generator(images, features)
        if self.training:
            loss_cls, loss_reg = self.loss_evaluator(anchors, box_cls, box_regression, targets)
            return anchors, {"loss_retina_cls": loss_cls, "loss_retina_reg": loss_reg}
        with torch.no_grad():
            return self.box_selector_test(anchors, box_cls, box_regression), {}


def build_retinanet(cfg, in_channels):
    return RetinaNetModule(cfg, in_channels)
