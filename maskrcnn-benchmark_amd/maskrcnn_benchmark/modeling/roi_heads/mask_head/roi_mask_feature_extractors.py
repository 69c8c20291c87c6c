"""Mask-head feature extractor (reference roi_heads/mask_head/roi_mask_feature_extractors.py:17-70)."""
from torch import nn
from torch.nn import functional as F

from maskrcnn_benchmark.layers.misc import conv_bias_act
from maskrcnn_benchmark.modeling import registry
from maskrcnn_benchmark.modeling.make_layers import make_conv3x3
from maskrcnn_benchmark.modeling.poolers import make_pooler

from ..box_head.roi_box_feature_extractors import ResNet50Conv5ROIFeatureExtractor

registry.ROI_MASK_FEATURE_EXTRACTORS.register("ResNet50Conv5ROIFeatureExtractor",
                                              ResNet50Conv5ROIFeatureExtractor)


@registry.ROI_MASK_FEATURE_EXTRACTORS.register("MaskRCNNFPNFeatureExtractor")
class MaskRCNNFPNFeatureExtractor(nn.Module):
    """multi-level ROIAlign (14x14) followed by a stack of 3x3 conv + relu."""

    def __init__(self, cfg, in_channels):
        super(MaskRCNNFPNFeatureExtractor, self).__init__()
        H = cfg.MODEL.ROI_MASK_HEAD
        self.pooler = make_pooler(cfg, "ROI_MASK_HEAD")
        self.blocks = []
        nxt = in_channels
        for i, width in enumerate(H.CONV_LAYERS, 1):
            name = "mask_fcn{}".format(i)
            self.add_module(name, make_conv3x3(nxt, width, dilation=H.DILATION, stride=1, use_gn=H.USE_GN))
            nxt = width
            self.blocks.append(name)
        self.out_channels = nxt

    def forward(self, x, proposals):
        x = self.pooler(x, proposals)
        for name in self.blocks:
            x = conv_bias_act(getattr(self, name), x, relu=True)
        return x


def make_roi_mask_feature_extractor(cfg, in_channels):
    return registry.ROI_MASK_FEATURE_EXTRACTORS[cfg.MODEL.ROI_MASK_HEAD.FEATURE_EXTRACTOR](cfg, in_channels)
