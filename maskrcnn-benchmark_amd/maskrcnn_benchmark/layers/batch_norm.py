"""FrozenBatchNorm2d (reference layers/batch_norm.py:6-31): fixed statistics and affine.

`forward(x)` is the reference's module (x * scale + bias).  `fused(x, relu, residual)` is the form
the backbone uses here: affine (+ residual add) (+ ReLU) in ONE pass of the hand-written HIP
kernel (csrc/frozen_bn.hip) instead of three to four PyTorch elementwise launches per convolution,
with a one-pass backward; the folded scale / bias are cached until a buffer changes."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from maskrcnn_benchmark import _C


class _FrozenBNAct(Function):
    @staticmethod
    def forward(ctx, x, scale, bias, residual, relu):
        y = _C.frozen_bn_act_forward(x, scale, bias, residual, relu)
        ctx.relu = relu
        ctx.save_for_backward(y if relu else None, scale)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        y, scale = ctx.saved_tensors
        need_x, need_res = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
        gx, gres = _C.frozen_bn_act_backward(grad_y, y, scale, ctx.relu, need_res)
        return (gx if need_x else None), None, None, gres, None


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, n):
        super(FrozenBatchNorm2d, self).__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self._folded = None

    def folded(self):
        """(scale, bias) fp32 [C]: scale = weight * rsqrt(var), bias = bias - mean * scale."""
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((b._version, b.data_ptr()) for b in bufs)
        if self._folded is None or self._folded[0] != key:
            with torch.no_grad():
                scale = self.weight.float() * self.running_var.float().rsqrt()
                bias = self.bias.float() - self.running_mean.float() * scale
            self._folded = (key, scale.contiguous(), bias.contiguous())
        return self._folded[1], self._folded[2]

    def forward(self, x):
        # the folded scale/bias follow the activation dtype (reference casts the buffers to half)
        scale, bias = self.folded()
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + bias.reshape(1, -1, 1, 1).to(x.dtype)

    def fused(self, x, relu=False, residual=None):
        scale, bias = self.folded()
        if residual is not None and residual.dtype != x.dtype:
            residual = residual.to(x.dtype)
        return _FrozenBNAct.apply(x, scale, bias, residual, relu)
