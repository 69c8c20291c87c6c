"""Feature pyramid (reference modeling/backbone/fpn.py:7-99): 1x1 lateral convs, nearest-neighbour
top-down pathway, 3x3 output convs, plus an extra-level block (P6 by stride-2 subsampling for the
RPN; P6/P7 convs for RetinaNet)."""
import torch.nn.functional as F
from torch import nn

from maskrcnn_benchmark import _C
from maskrcnn_benchmark.layers.misc import conv_bias_act

_FUSED_TOPDOWN = __import__("os").environ.get("DETOPS_FPN_TOPDOWN", "fused") != "torch"   # A/B switch


class FPN(nn.Module):
    def __init__(self, in_channels_list, out_channels, conv_block, top_blocks=None):
        super(FPN, self).__init__()
        self.inner_blocks, self.layer_blocks = [], []
        for idx, in_channels in enumerate(in_channels_list, 1):
            inner, layer = "fpn_inner{}".format(idx), "fpn_layer{}".format(idx)
            if in_channels == 0:
                continue
            self.add_module(inner, conv_block(in_channels, out_channels, 1))
            self.add_module(layer, conv_block(out_channels, out_channels, 3, 1))
            self.inner_blocks.append(inner)
            self.layer_blocks.append(layer)
        self.top_blocks = top_blocks

    def forward(self, x):
        """x: C2..C5 (fine -> coarse).  Returns P2..P5 (+ extra levels), fine -> coarse."""
        last = conv_bias_act(getattr(self, self.inner_blocks[-1]), x[-1])
        results = [conv_bias_act(getattr(self, self.layer_blocks[-1]), last)]
        for feat, inner, layer in zip(x[:-1][::-1], self.inner_blocks[:-1][::-1], self.layer_blocks[:-1][::-1]):
            if not inner:
                continue
            lateral = conv_bias_act(getattr(self, inner), feat)
            if _FUSED_TOPDOWN and _C.on_device(lateral) and lateral.dtype == last.dtype and lateral.dtype in _C._lib.DTYPE_CODE:
                # one streaming pass instead of interpolate + add and their full-size temporary (csrc/fpn_topdown.hip)
                last = _C.fpn_topdown(lateral, last)
            else:
                top_down = F.interpolate(last, size=(int(lateral.shape[-2]), int(lateral.shape[-1])), mode="nearest")
                last = lateral + top_down
            results.insert(0, conv_bias_act(getattr(self, layer), last))
        if isinstance(self.top_blocks, LastLevelP6P7):
            results.extend(self.top_blocks(x[-1], results[-1]))
        elif isinstance(self.top_blocks, LastLevelMaxPool):
            results.extend(self.top_blocks(results[-1]))
        return tuple(results)


class LastLevelMaxPool(nn.Module):
    def forward(self, x):
        return [F.max_pool2d(x, 1, 2, 0)]


class LastLevelP6P7(nn.Module):
    """P6 = conv3x3/2(C5 or P5), P7 = conv3x3/2(relu(P6)) — RetinaNet's extra levels."""

    def __init__(self, in_channels, out_channels):
        super(LastLevelP6P7, self).__init__()
        self.p6 = nn.Conv2d(in_channels, out_channels, 3, 2, 1)
        self.p7 = nn.Conv2d(out_channels, out_channels, 3, 2, 1)
        for m in (self.p6, self.p7):
            nn.init.kaiming_uniform_(m.weight, a=1)
            nn.init.constant_(m.bias, 0)
        self.use_P5 = in_channels == out_channels

    def forward(self, c5, p5):
        p6 = conv_bias_act(self.p6, p5 if self.use_P5 else c5)
        return [p6, conv_bias_act(self.p7, F.relu(p6))]
