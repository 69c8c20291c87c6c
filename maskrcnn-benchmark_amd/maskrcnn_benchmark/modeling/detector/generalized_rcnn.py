"""GeneralizedRCNN (reference modeling/detector/generalized_rcnn.py:16-65): backbone -> RPN ->
ROI heads.  `model(images, targets)` returns a dict of losses in training and a list of BoxLists
(detections) in eval mode."""
import torch
from torch import nn

from maskrcnn_benchmark.structures.image_list import to_image_list

from maskrcnn_benchmark.layers.half_weights import HalfWeights

from ..backbone import build_backbone
from ..roi_heads.roi_heads import build_roi_heads
from ..rpn.loss import begin_step
from ..rpn.rpn import build_rpn


class GeneralizedRCNN(nn.Module):
    def __init__(self, cfg):
        super(GeneralizedRCNN, self).__init__()
        self.backbone = build_backbone(cfg)
        self.rpn = build_rpn(cfg, self.backbone.out_channels)
        self.roi_heads = build_roi_heads(cfg, self.backbone.out_channels)
        self.channels_last = False          # see set_channels_last
        self.channels_last_heads = False
        # mixed precision: every weight's half copy from one multi-tensor launch per forward instead of one cast per layer
        # (layers/half_weights.py); a plain attribute, not a sub-module: state_dict / parameters() do not see it
        object.__setattr__(self, "half_weights", HalfWeights(self))

    def set_channels_last(self, on=True, heads=False):
        """Run the backbone + FPN on channels-last (NHWC) activations: MIOpen's implicit-GEMM kernels then read and write
        their native layout (no `batched_transpose_*` launches around them) and the fused FrozenBN / top-down kernels
        follow the tensor's layout.  `heads=False`: the pyramid is handed to the RPN / ROI heads as NCHW (one conversion
        per level).  `heads=True`: the heads take the channels-last pyramid as it is — the RPN head's convolutions run
        channels-last (its two small outputs are handed to the proposal / loss kernels as NCHW), the poolers read the
        pyramid in place (csrc/roi_align_nhwc.hip), the box head's pooled tensor stays [K, C, 7, 7] for its FC layers and
        the mask head (pooler output, convolutions, deconvolution) runs channels-last.  Parameters, buffers and the
        state_dict are untouched (a memory format is a stride permutation, not a shape)."""
        fmt = torch.channels_last if on else torch.contiguous_format
        self.backbone.to(memory_format=fmt)
        heads = bool(on and heads and hasattr(self.rpn, "head") and type(self.rpn).__name__ in ("RPNModule", "RetinaNetModule"))
        hfmt = torch.channels_last if heads else torch.contiguous_format
        if type(self.rpn).__name__ in ("RPNModule", "RetinaNetModule"):
            # (RetinaNet: the two towers and the class / box convolutions; their channels-last outputs ARE the (N, H, W, A, C)
            #  order the loss and the post-processor flatten them to — modeling/rpn/utils.py::permute_and_flatten)
            self.rpn.head.to(memory_format=hfmt)
        mask = self.roi_heads["mask"] if (self.roi_heads and "mask" in self.roi_heads) else None
        if mask is not None and hasattr(mask.feature_extractor, "pooler") and hasattr(mask.feature_extractor, "blocks"):
            box = self.roi_heads["box"] if "box" in self.roi_heads else None
            if box is None or mask.feature_extractor is not box.feature_extractor:
                mask.feature_extractor.to(memory_format=hfmt)
                mask.predictor.to(memory_format=hfmt)
                mask.feature_extractor.pooler.output_channels_last = heads
        self.channels_last = bool(on)
        self.channels_last_heads = heads
        return self

    def forward(self, images, targets=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        begin_step()   # the padded-target batch is shared by the callers of ONE forward, never across forwards
        half = self.half_weights
        try:
            images = to_image_list(images)
            x = images.tensors
            if half.usable(x):
                half.install(torch.get_autocast_dtype(x.device.type))
            if self.channels_last:
                x = x.contiguous(memory_format=torch.channels_last)
            features = self.backbone(x)
            if self.channels_last and not self.channels_last_heads:
                features = tuple(f.contiguous() for f in features)
            proposals, proposal_losses = self.rpn(images, features, targets)
            if self.roi_heads:
                x, result, detector_losses = self.roi_heads(features, proposals, targets)
            else:  # RPN-only models (RetinaNet) have no ROI heads
                x, result, detector_losses = features, proposals, {}
        finally:
            half.remove()  # the modules' `weight` attributes resolve to the fp32 parameters again
            begin_step()   # ... and never beyond it: nothing of this batch stays referenced after the forward (or its exception)
        if self.training:
            losses = {}
            losses.update(detector_losses)
            losses.update(proposal_losses)
            return losses
        return result
