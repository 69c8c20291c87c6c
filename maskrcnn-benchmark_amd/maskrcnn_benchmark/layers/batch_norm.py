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
        """(scale, bias) fp32 [C]: scale = weight * rsqrt(var), bias = bias - mean * scale.  Cached; the key checked on
        the hot path is the four buffers' version counters (in-place writes, `load_state_dict`) AND their storage addresses
        (`buf.data = x`, `module._buffers[name] = t`, which bypass `__setattr__` / `_apply`).  The one write no key can
        see is an in-place write through the `.data` alias (`buf.data.copy_(x)` bumps no counter and moves nothing):
        call `invalidate()` after it."""
        b = self._buffers      # (nn.Module.__getattr__ costs ~0.5 us per buffer access)
        w, bi, m, v = b["weight"], b["bias"], b["running_mean"], b["running_var"]
        version = (w._version + bi._version + m._version + v._version, w.data_ptr(), bi.data_ptr(), m.data_ptr(), v.data_ptr())
        f = self._folded
        if f is None or f[0] != version:
            with torch.no_grad():
                scale = self.weight.float() * self.running_var.float().rsqrt()
                bias = self.bias.float() - self.running_mean.float() * scale
            f = self._folded = (version, scale.contiguous(), bias.contiguous())
        return f[1], f[2]

    def invalidate(self):
        """drop the cached folded scale / bias (needed only after in-place writes through `.data`, see folded())"""
        self._folded = None

    def _apply(self, fn, *args, **kwargs):
        self._folded = None
        return super(FrozenBatchNorm2d, self)._apply(fn, *args, **kwargs)

    def __setattr__(self, name, value):
        if name in ("weight", "bias", "running_mean", "running_var"):
            object.__setattr__(self, "_folded", None)
        super(FrozenBatchNorm2d, self).__setattr__(name, value)

    def forward(self, x):
        # the folded scale/bias follow the activation dtype (reference casts the buffers to half)
        scale, bias = self.folded()
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + bias.reshape(1, -1, 1, 1).to(x.dtype)

    def fused(self, x, relu=False, residual=None):
        scale, bias = self.folded()
        if residual is not None and residual.dtype != x.dtype:
            residual = residual.to(x.dtype)
        return _FrozenBNAct.apply(x, scale, bias, residual, relu)
