"""Layer factories with the reference's initialisation (reference modeling/make_layers.py:13-123)."""
import torch
from torch import nn
from torch.nn import functional as F

from maskrcnn_benchmark.config import cfg
from maskrcnn_benchmark.layers import Conv2d


def get_group_gn(dim, dim_per_gp, num_groups):
    """number of GroupNorm groups from either a per-group width or a group count (one must be -1)."""
    assert dim_per_gp == -1 or num_groups == -1, "GroupNorm: can only specify G or C/G."
    if dim_per_gp > 0:
        assert dim % dim_per_gp == 0, "dim: {}, dim_per_gp: {}".format(dim, dim_per_gp)
        return dim // dim_per_gp
    assert dim % num_groups == 0, "dim: {}, num_groups: {}".format(dim, num_groups)
    return num_groups


def group_norm(out_channels, affine=True, divisor=1):
    out_channels = out_channels // divisor
    dim_per_gp = cfg.MODEL.GROUP_NORM.DIM_PER_GP // divisor
    num_groups = cfg.MODEL.GROUP_NORM.NUM_GROUPS // divisor
    return nn.GroupNorm(get_group_gn(out_channels, dim_per_gp, num_groups), out_channels,
                        cfg.MODEL.GROUP_NORM.EPSILON, affine)


def make_conv3x3(in_channels, out_channels, dilation=1, stride=1, use_gn=False, use_relu=False,
                 kaiming_init=True):
    conv = Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=dilation,
                  dilation=dilation, bias=not use_gn)
    if kaiming_init:
        nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
    else:
        nn.init.normal_(conv.weight, std=0.01)
    if not use_gn:
        nn.init.constant_(conv.bias, 0)
    layers = [conv]
    if use_gn:
        layers.append(group_norm(out_channels))
    if use_relu:
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers) if len(layers) > 1 else conv


def make_fc(dim_in, hidden_dim, use_gn=False):
    fc = nn.Linear(dim_in, hidden_dim, bias=not use_gn)
    nn.init.kaiming_uniform_(fc.weight, a=1)
    if use_gn:
        return nn.Sequential(fc, group_norm(hidden_dim))
    nn.init.constant_(fc.bias, 0)
    return fc


def conv_with_kaiming_uniform(use_gn=False, use_relu=False):
    def make_conv(in_channels, out_channels, kernel_size, stride=1, dilation=1):
        conv = Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                      padding=dilation * (kernel_size - 1) // 2, dilation=dilation, bias=not use_gn)
        nn.init.kaiming_uniform_(conv.weight, a=1)
        if not use_gn:
            nn.init.constant_(conv.bias, 0)
        layers = [conv]
        if use_gn:
            layers.append(group_norm(out_channels))
        if use_relu:
            layers.append(nn.ReLU(inplace=True))
        return nn.Sequential(*layers) if len(layers) > 1 else conv

    return make_conv
