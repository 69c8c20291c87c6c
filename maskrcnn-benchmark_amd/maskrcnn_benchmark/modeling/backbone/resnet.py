"""ResNet trunk (reference modeling/backbone/resnet.py:24-451), Caffe2-style: stride on the first
1x1 conv (`STRIDE_IN_1X1`), frozen batch-norm folded to a per-channel affine, optional deformable
3x3 (`STAGE_WITH_DCN`).  The convolutions themselves run on MIOpen (MFMA); this file is structure
only."""
from collections import namedtuple

import torch.nn.functional as F
from torch import nn

from maskrcnn_benchmark.layers import Conv2d, DFConv2d, FrozenBatchNorm2d
from maskrcnn_benchmark.modeling.make_layers import group_norm

StageSpec = namedtuple("StageSpec", ["index", "block_count", "return_features"])


def _stages(counts, returns):
    return tuple(StageSpec(index=i + 1, block_count=c, return_features=r)
                 for i, (c, r) in enumerate(zip(counts, returns)))


_STAGE_SPECS = {
    "R-50-C4": _stages((3, 4, 6), (False, False, True)),
    "R-50-C5": _stages((3, 4, 6, 3), (False, False, False, True)),
    "R-101-C4": _stages((3, 4, 23), (False, False, True)),
    "R-101-C5": _stages((3, 4, 23, 3), (False, False, False, True)),
    "R-50-FPN": _stages((3, 4, 6, 3), (True,) * 4),
    "R-50-FPN-RETINANET": _stages((3, 4, 6, 3), (True,) * 4),
    "R-101-FPN": _stages((3, 4, 23, 3), (True,) * 4),
    "R-101-FPN-RETINANET": _stages((3, 4, 23, 3), (True,) * 4),
    "R-152-FPN": _stages((3, 8, 36, 3), (True,) * 4),
}


def _kaiming(conv):
    nn.init.kaiming_uniform_(conv.weight, a=1)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 -> 1x1 residual block with a projection shortcut when the shape changes."""

    def __init__(self, in_channels, bottleneck_channels, out_channels, num_groups, stride_in_1x1,
                 stride, dilation, norm_func, dcn_config):
        super(Bottleneck, self).__init__()
        self.downsample = None
        if in_channels != out_channels:
            down_stride = stride if dilation == 1 else 1
            self.downsample = nn.Sequential(
                Conv2d(in_channels, out_channels, kernel_size=1, stride=down_stride, bias=False),
                norm_func(out_channels))
            _kaiming(self.downsample[0])
        if dilation > 1:
            stride = 1  # dilated res5 keeps the resolution
        stride_1x1, stride_3x3 = (stride, 1) if stride_in_1x1 else (1, stride)

        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=stride_1x1, bias=False)
        self.bn1 = norm_func(bottleneck_channels)
        if dcn_config.get("stage_with_dcn", False):
            self.conv2 = DFConv2d(bottleneck_channels, bottleneck_channels,
                                  with_modulated_dcn=dcn_config.get("with_modulated_dcn", False),
                                  kernel_size=3, stride=stride_3x3, groups=num_groups, dilation=dilation,
                                  deformable_groups=dcn_config.get("deformable_groups", 1), bias=False)
        else:
            self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=stride_3x3,
                                padding=dilation, bias=False, groups=num_groups, dilation=dilation)
            _kaiming(self.conv2)
        self.bn2 = norm_func(bottleneck_channels)
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False)
        self.bn3 = norm_func(out_channels)
        _kaiming(self.conv1)
        _kaiming(self.conv3)

    def forward(self, x):
        if isinstance(self.bn1, FrozenBatchNorm2d) and x.numel() > 0:
            # fused affine(+residual)+ReLU: one HIP pass per convolution output (csrc/frozen_bn.hip)
            if self.downsample is None:
                identity = x
            else:
                identity = self.downsample[1].fused(self.downsample[0](x))
            out = self.bn1.fused(self.conv1(x), relu=True)
            out = self.bn2.fused(self.conv2(out), relu=True)
            return self.bn3.fused(self.conv3(out), relu=True, residual=identity)
        identity = x if self.downsample is None else self.downsample(x)
        out = F.relu_(self.bn1(self.conv1(x)))
        out = F.relu_(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        out += identity
        return F.relu_(out)


class BottleneckWithFixedBatchNorm(Bottleneck):
    def __init__(self, in_channels, bottleneck_channels, out_channels, num_groups=1, stride_in_1x1=True,
                 stride=1, dilation=1, dcn_config=None):
        super(BottleneckWithFixedBatchNorm, self).__init__(
            in_channels, bottleneck_channels, out_channels, num_groups, stride_in_1x1, stride, dilation,
            FrozenBatchNorm2d, dcn_config or {})


class BottleneckWithGN(Bottleneck):
    def __init__(self, in_channels, bottleneck_channels, out_channels, num_groups=1, stride_in_1x1=True,
                 stride=1, dilation=1, dcn_config=None):
        super(BottleneckWithGN, self).__init__(
            in_channels, bottleneck_channels, out_channels, num_groups, stride_in_1x1, stride, dilation,
            group_norm, dcn_config or {})


class Stem(nn.Module):
    """7x7/2 conv + norm + relu + 3x3/2 max-pool."""

    def __init__(self, cfg, norm_func):
        super(Stem, self).__init__()
        out_channels = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
        self.conv1 = Conv2d(3, out_channels, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_func(out_channels)
        _kaiming(self.conv1)

    def forward(self, x):
        if isinstance(self.bn1, FrozenBatchNorm2d) and x.numel() > 0:
            x = self.bn1.fused(self.conv1(x), relu=True)
        else:
            x = F.relu_(self.bn1(self.conv1(x)))
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


class StemWithFixedBatchNorm(Stem):
    def __init__(self, cfg):
        super(StemWithFixedBatchNorm, self).__init__(cfg, norm_func=FrozenBatchNorm2d)


class StemWithGN(Stem):
    def __init__(self, cfg):
        super(StemWithGN, self).__init__(cfg, norm_func=group_norm)


_TRANSFORMATION_MODULES = {"BottleneckWithFixedBatchNorm": BottleneckWithFixedBatchNorm,
                           "BottleneckWithGN": BottleneckWithGN}
_STEM_MODULES = {"StemWithFixedBatchNorm": StemWithFixedBatchNorm, "StemWithGN": StemWithGN}


def _make_stage(block, in_channels, bottleneck_channels, out_channels, block_count, num_groups,
                stride_in_1x1, first_stride, dilation=1, dcn_config=None):
    blocks, stride = [], first_stride
    for _ in range(block_count):
        blocks.append(block(in_channels, bottleneck_channels, out_channels, num_groups, stride_in_1x1,
                            stride, dilation=dilation, dcn_config=dcn_config))
        stride, in_channels = 1, out_channels
    return nn.Sequential(*blocks)


class ResNet(nn.Module):
    def __init__(self, cfg):
        super(ResNet, self).__init__()
        R = cfg.MODEL.RESNETS
        block = _TRANSFORMATION_MODULES[R.TRANS_FUNC]
        self.stem = _STEM_MODULES[R.STEM_FUNC](cfg)
        in_channels = R.STEM_OUT_CHANNELS
        width = R.NUM_GROUPS * R.WIDTH_PER_GROUP
        self.stages, self.return_features = [], {}
        for spec in _STAGE_SPECS[cfg.MODEL.BACKBONE.CONV_BODY]:
            name = "layer" + str(spec.index)
            factor = 2 ** (spec.index - 1)
            out_channels = R.RES2_OUT_CHANNELS * factor
            stage = _make_stage(block, in_channels, width * factor, out_channels, spec.block_count,
                                R.NUM_GROUPS, R.STRIDE_IN_1X1, first_stride=int(spec.index > 1) + 1,
                                dcn_config={"stage_with_dcn": R.STAGE_WITH_DCN[spec.index - 1],
                                            "with_modulated_dcn": R.WITH_MODULATED_DCN,
                                            "deformable_groups": R.DEFORMABLE_GROUPS})
            in_channels = out_channels
            self.add_module(name, stage)
            self.stages.append(name)
            self.return_features[name] = spec.return_features
        self._freeze_backbone(cfg.MODEL.BACKBONE.FREEZE_CONV_BODY_AT)

    def _freeze_backbone(self, freeze_at):
        for i in range(max(freeze_at, 0)):
            m = self.stem if i == 0 else getattr(self, "layer" + str(i))
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, x):
        outputs = []
        x = self.stem(x)
        for name in self.stages:
            x = getattr(self, name)(x)
            if self.return_features[name]:
                outputs.append(x)
        return outputs


class ResNetHead(nn.Module):
    """res5 applied to pooled ROIs (C4 models, reference :146-200)."""

    def __init__(self, block_module, stages, num_groups=1, width_per_group=64, stride_in_1x1=True,
                 stride_init=None, res2_out_channels=256, dilation=1, dcn_config=None):
        super(ResNetHead, self).__init__()
        factor = 2 ** (stages[0].index - 1)
        out_channels = res2_out_channels * factor
        in_channels = out_channels // 2
        block = _TRANSFORMATION_MODULES[block_module]
        self.stages = []
        stride = stride_init
        for spec in stages:
            name = "layer" + str(spec.index)
            if not stride:
                stride = int(spec.index > 1) + 1
            self.add_module(name, _make_stage(block, in_channels, num_groups * width_per_group * factor,
                                              out_channels, spec.block_count, num_groups, stride_in_1x1,
                                              first_stride=stride, dilation=dilation, dcn_config=dcn_config))
            stride = None
            self.stages.append(name)
        self.out_channels = out_channels

    def forward(self, x):
        for name in self.stages:
            x = getattr(self, name)(x)
        return x
