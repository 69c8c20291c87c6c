"""FrozenBatchNorm2d (reference layers/batch_norm.py:6-31): fixed statistics and affine."""
import torch
from torch import nn


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, n):
        super(FrozenBatchNorm2d, self).__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def forward(self, x):
        # the folded scale/bias follow the activation dtype (reference casts the buffers to half)
        scale = self.weight * self.running_var.rsqrt()
        bias = self.bias - self.running_mean * scale
        scale = scale.reshape(1, -1, 1, 1).to(x.dtype)
        bias = bias.reshape(1, -1, 1, 1).to(x.dtype)
        return x * scale + bias
