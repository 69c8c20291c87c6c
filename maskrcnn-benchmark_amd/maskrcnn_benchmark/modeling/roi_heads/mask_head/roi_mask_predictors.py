"""Mask predictors (reference roi_heads/mask_head/roi_mask_predictors.py:10-57)."""
from torch import nn
from torch.nn import functional as F

from maskrcnn_benchmark.layers import Conv2d, ConvTranspose2d
from maskrcnn_benchmark.layers.misc import conv_bias_act
from maskrcnn_benchmark.modeling import registry


def _kaiming_out(module):
    for name, p in module.named_parameters():
        if "bias" in name:
            nn.init.constant_(p, 0)
        elif "weight" in name:
            nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")


@registry.ROI_MASK_PREDICTOR.register("MaskRCNNC4Predictor")
class MaskRCNNC4Predictor(nn.Module):
    """2x2 stride-2 transposed conv (14 -> 28) + relu, then a 1x1 conv to per-class mask logits."""

    def __init__(self, cfg, in_channels):
        super(MaskRCNNC4Predictor, self).__init__()
        num_classes = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES
        dim_reduced = cfg.MODEL.ROI_MASK_HEAD.CONV_LAYERS[-1]
        self.conv5_mask = ConvTranspose2d(in_channels, dim_reduced, 2, 2, 0)
        self.mask_fcn_logits = Conv2d(dim_reduced, num_classes, 1, 1, 0)
        _kaiming_out(self)

    def forward(self, x):
        return conv_bias_act(self.mask_fcn_logits, conv_bias_act(self.conv5_mask, x, relu=True))


@registry.ROI_MASK_PREDICTOR.register("MaskRCNNConv1x1Predictor")
class MaskRCNNConv1x1Predictor(nn.Module):
    def __init__(self, cfg, in_channels):
        super(MaskRCNNConv1x1Predictor, self).__init__()
        self.mask_fcn_logits = Conv2d(in_channels, cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES, 1, 1, 0)
        _kaiming_out(self)

    def forward(self, x):
        return self.mask_fcn_logits(x)


def make_roi_mask_predictor(cfg, in_channels):
    return registry.ROI_MASK_PREDICTOR[cfg.MODEL.ROI_MASK_HEAD.PREDICTOR](cfg, in_channels)
