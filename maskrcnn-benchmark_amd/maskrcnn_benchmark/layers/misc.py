"""Conv2d / ConvTranspose2d / BatchNorm2d / interpolate shims that accept empty batches, and
DFConv2d (reference layers/misc.py:18-203).

The reference needed `_NewEmptyTensorOp` because PyTorch 1.0 could not run these ops on a
zero-element batch.  The shape arithmetic is kept (an empty ROI batch still has to produce a
correctly shaped, differentiable empty tensor), everything else defers to torch.nn.
"""
import math
import os

import torch
from torch import nn
from torch.nn.modules.utils import _ntuple


class _EmptyLike(torch.autograd.Function):
    """x (0 elements) -> empty tensor of `shape`, differentiable (gradient is an empty tensor)."""

    @staticmethod
    def forward(ctx, x, shape):
        ctx.in_shape = x.shape
        return x.new_empty(shape)

    @staticmethod
    def backward(ctx, grad):
        return _EmptyLike.apply(grad, ctx.in_shape), None


def _conv_out(size, pad, dil, k, stride):
    return (size + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


# A/B switch: "0" keeps half-precision (autocast) convolutions on the module call (bias inside the convolution)
BIAS_ACT_HALF = os.environ.get("DETOPS_BIAS_ACT_HALF", "1") != "0"


def _half_compute(x):
    return x.dtype != torch.float32 or torch.is_autocast_enabled()


def conv_bias_act(conv, x, relu=False):
    """`conv(x)` (+ ReLU) for a convolution / transposed-convolution MODULE.  On a channels-last activation (fp32, or fp16 /
    bf16 under autocast) on the device the bias leaves the convolution call: the convolution runs without it and ONE fused
    pass adds it (and applies the ReLU); the backward's bias gradient is a column sum inside that pass' mirror (csrc/bias_act.hip) instead of PyTorch's
    strided `sum((0, 2, 3))`, which its generic reduce kernel serves at ~0.1 TB/s on channels-last gradients (2.3 ms of
    the fp32 step, profiles/r06d_*).  Same values as the module call (x + b is one rounding either way).  Everything else
    (NCHW, CPU, norm-wrapped blocks, widths the kernel does not serve) takes the module call."""
    from maskrcnn_benchmark import _C
    plain = type(conv) in (nn.Conv2d, Conv2d)
    trans = type(conv) in (nn.ConvTranspose2d, ConvTranspose2d)
    if (plain or trans) and conv.bias is not None and x.numel() > 0 and conv.padding_mode == "zeros" and _C.is_channels_last(x) \
            and x.dtype in _C._lib.DTYPE_CODE and _C.on_device(x) and (BIAS_ACT_HALF or not _half_compute(x)):
        if plain:
            y = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        else:
            y = torch.nn.functional.conv_transpose2d(x, conv.weight, None, conv.stride, conv.padding, conv.output_padding,
                                                     conv.groups, conv.dilation)
        if _C.bias_act_supported(y, conv.bias):
            return _C.bias_act(y, conv.bias, relu)
        y = y + conv.bias.to(y.dtype).view(1, -1, 1, 1)
        return torch.relu(y) if relu else y
    y = conv(x)
    return torch.relu(y) if relu else y


class Conv2d(nn.Conv2d):
    def forward(self, x):
        if x.numel() > 0:
            return super(Conv2d, self).forward(x)
        hw = [_conv_out(i, p, di, k, d) for i, p, di, k, d in
              zip(x.shape[-2:], self.padding, self.dilation, self.kernel_size, self.stride)]
        return _EmptyLike.apply(x, [x.shape[0], self.weight.shape[0]] + hw)


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x):
        if x.numel() > 0:
            return super(ConvTranspose2d, self).forward(x)
        hw = [(i - 1) * d - 2 * p + (di * (k - 1) + 1) + op for i, p, di, k, d, op in
              zip(x.shape[-2:], self.padding, self.dilation, self.kernel_size, self.stride,
                  self.output_padding)]
        return _EmptyLike.apply(x, [x.shape[0], self.weight.shape[1] * self.groups] + hw)


class BatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):
        if x.numel() > 0:
            return super(BatchNorm2d, self).forward(x)
        return _EmptyLike.apply(x, x.shape)


def interpolate(input, size=None, scale_factor=None, mode="nearest", align_corners=None):
    if input.numel() > 0:
        return torch.nn.functional.interpolate(input, size, scale_factor, mode, align_corners)
    if (size is None) == (scale_factor is None):
        raise ValueError("exactly one of size or scale_factor should be defined")
    if size is None:
        if isinstance(scale_factor, tuple) and len(scale_factor) != 2:
            raise ValueError("scale_factor shape must match input shape. Input is 2D, scale_factor "
                             "size is {}".format(len(scale_factor)))
        sf = _ntuple(2)(scale_factor)
        size = [int(math.floor(input.size(i + 2) * sf[i])) for i in range(2)]
    return _EmptyLike.apply(input, tuple(input.shape[:-2]) + tuple(_ntuple(2)(size)))


class DFConv2d(nn.Module):
    """Deformable convolutional layer: a plain conv predicts the offsets (and modulation mask),
    the deformable conv consumes them (reference layers/misc.py:114-203)."""

    def __init__(self, in_channels, out_channels, with_modulated_dcn=True, kernel_size=3, stride=1,
                 groups=1, dilation=1, deformable_groups=1, bias=False):
        super(DFConv2d, self).__init__()
        if isinstance(kernel_size, (list, tuple)):
            assert isinstance(stride, (list, tuple)) and isinstance(dilation, (list, tuple))
            assert len(kernel_size) == 2 and len(stride) == 2 and len(dilation) == 2
            padding = (dilation[0] * (kernel_size[0] - 1) // 2, dilation[1] * (kernel_size[1] - 1) // 2)
            taps = kernel_size[0] * kernel_size[1]
        else:
            padding = dilation * (kernel_size - 1) // 2
            taps = kernel_size * kernel_size
        if with_modulated_dcn:
            from maskrcnn_benchmark.layers import ModulatedDeformConv as conv_block
            offset_channels = taps * 3  # 2 offsets + 1 mask per tap (default 27)
        else:
            from maskrcnn_benchmark.layers import DeformConv as conv_block
            offset_channels = taps * 2  # default 18
        self.offset = Conv2d(in_channels, deformable_groups * offset_channels, kernel_size=kernel_size,
                             stride=stride, padding=padding, groups=1, dilation=dilation)
        nn.init.kaiming_uniform_(self.offset.weight, a=1)
        nn.init.constant_(self.offset.bias, 0.)
        self.conv = conv_block(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                               padding=padding, dilation=dilation, groups=groups,
                               deformable_groups=deformable_groups, bias=bias)
        self.with_modulated_dcn = with_modulated_dcn
        self.kernel_size = kernel_size
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.offset_split = taps * 2 * deformable_groups  # 18 for 3x3, dg=1 (reference hard-codes 18/-9)

    def forward(self, x):
        if x.numel() > 0:
            if not self.with_modulated_dcn:
                return self.conv(x, self.offset(x))
            offset_mask = self.offset(x)
            offset = offset_mask[:, :self.offset_split, :, :]
            mask = offset_mask[:, self.offset_split:, :, :].sigmoid()
            return self.conv(x, offset, mask)
        hw = [_conv_out(i, p, di, k, d) for i, p, di, k, d in
              zip(x.shape[-2:], _ntuple(2)(self.padding), _ntuple(2)(self.dilation),
                  _ntuple(2)(self.kernel_size), _ntuple(2)(self.stride))]
        return _EmptyLike.apply(x, [x.shape[0], self.conv.weight.shape[0]] + hw)
