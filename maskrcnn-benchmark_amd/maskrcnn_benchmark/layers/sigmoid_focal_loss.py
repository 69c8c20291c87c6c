"""SigmoidFocalLoss (reference layers/sigmoid_focal_loss.py:9-74).

`sigmoid_focal_loss_cuda(logits, targets, gamma, alpha) -> [R,C]` keeps the reference's autograd
function; the module's `losses.sum()` (reference :66-67) is served by a fused forward-sum /
scalar-gradient backward pair so that neither the [R,C] loss tensor nor a broadcast [R,C] upstream
gradient is materialised.  HIP only: CPU tensors raise (no Python composite fallback).
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from maskrcnn_benchmark import _C


class _SigmoidFocalLoss(Function):
    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        ctx.save_for_backward(logits, targets)
        num_classes = logits.shape[1]
        ctx.num_classes = num_classes
        ctx.gamma = gamma
        ctx.alpha = alpha
        return _C.sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loss):
        logits, targets = ctx.saved_tensors
        d_loss = d_loss.contiguous()
        d_logits = _C.sigmoid_focalloss_backward(logits, targets, d_loss, ctx.num_classes, ctx.gamma,
                                                 ctx.alpha)
        return d_logits, None, None, None, None


sigmoid_focal_loss_cuda = _SigmoidFocalLoss.apply


class _SigmoidFocalLossSum(Function):
    @staticmethod
    def forward(ctx, logits, targets, gamma, alpha):
        ctx.save_for_backward(logits, targets)
        ctx.num_classes = logits.shape[1]
        ctx.gamma = gamma
        ctx.alpha = alpha
        return _C.sigmoid_focalloss_forward_sum(logits, targets, ctx.num_classes, gamma, alpha)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_sum):
        logits, targets = ctx.saved_tensors
        d_logits = _C.sigmoid_focalloss_backward_scalar(logits, targets, d_sum, ctx.num_classes,
                                                        ctx.gamma, ctx.alpha)
        return d_logits, None, None, None


sigmoid_focal_loss_sum = _SigmoidFocalLossSum.apply


class SigmoidFocalLoss(nn.Module):
    def __init__(self, gamma, alpha):
        super(SigmoidFocalLoss, self).__init__()
        self.gamma = gamma
        self.alpha = alpha

    def forward(self, logits, targets):
        if not _C.on_device(logits):
            raise RuntimeError("SigmoidFocalLoss: Not implemented on the CPU (HIP-only build)")
        with torch.autocast(device_type=logits.device.type, enabled=False):
            return sigmoid_focal_loss_sum(logits.float(), targets, self.gamma, self.alpha)

    def __repr__(self):
        tmpstr = self.__class__.__name__ + "("
        tmpstr += "gamma=" + str(self.gamma)
        tmpstr += ", alpha=" + str(self.alpha)
        tmpstr += ")"
        return tmpstr
