"""build_backbone(cfg) (reference modeling/backbone/backbone.py:13-83).  FBNet backbones are a
separate architecture-search family outside the BASELINE configs and are not built."""
from collections import OrderedDict

from torch import nn

from maskrcnn_benchmark.modeling import registry
from maskrcnn_benchmark.modeling.make_layers import conv_with_kaiming_uniform

from . import fpn as fpn_module
from . import resnet


@registry.BACKBONES.register("R-50-C4")
@registry.BACKBONES.register("R-50-C5")
@registry.BACKBONES.register("R-101-C4")
@registry.BACKBONES.register("R-101-C5")
def build_resnet_backbone(cfg):
    model = nn.Sequential(OrderedDict([("body", resnet.ResNet(cfg))]))
    model.out_channels = cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS
    return model


def _resnet_fpn(cfg, first_in, top_blocks):
    c2 = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
    out_channels = cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS
    body = resnet.ResNet(cfg)
    neck = fpn_module.FPN(in_channels_list=[first_in, c2 * 2, c2 * 4, c2 * 8], out_channels=out_channels,
                          conv_block=conv_with_kaiming_uniform(cfg.MODEL.FPN.USE_GN, cfg.MODEL.FPN.USE_RELU),
                          top_blocks=top_blocks)
    model = nn.Sequential(OrderedDict([("body", body), ("fpn", neck)]))
    model.out_channels = out_channels
    return model


@registry.BACKBONES.register("R-50-FPN")
@registry.BACKBONES.register("R-101-FPN")
@registry.BACKBONES.register("R-152-FPN")
def build_resnet_fpn_backbone(cfg):
    return _resnet_fpn(cfg, cfg.MODEL.RESNETS.RES2_OUT_CHANNELS, fpn_module.LastLevelMaxPool())


@registry.BACKBONES.register("R-50-FPN-RETINANET")
@registry.BACKBONES.register("R-101-FPN-RETINANET")
def build_resnet_fpn_p3p7_backbone(cfg):
    c2 = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
    out_channels = cfg.MODEL.RESNETS.BACKBONE_OUT_CHANNELS
    p6p7_in = c2 * 8 if cfg.MODEL.RETINANET.USE_C5 else out_channels
    # C2 gets no lateral (in_channels 0): RetinaNet's pyramid starts at P3
    return _resnet_fpn(cfg, 0, fpn_module.LastLevelP6P7(p6p7_in, out_channels))


def build_backbone(cfg):
    name = cfg.MODEL.BACKBONE.CONV_BODY
    assert name in registry.BACKBONES, \
        "cfg.MODEL.BACKBONE.CONV_BODY: {} is not registered in registry".format(name)
    return registry.BACKBONES[name](cfg)
