"""Deformable PS-ROI pooling autograd function (reference layers/dcn/deform_pool_func.py:8-95)
over `_C.deform_psroi_pooling_{forward,backward}` (csrc/deform_pool.hip).  HIP-only like the
reference's (`if not data.is_cuda: raise NotImplementedError`, :29-30)."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from maskrcnn_benchmark import _C


class DeformRoIPoolingFunction(Function):
    @staticmethod
    def forward(ctx, data, rois, offset, spatial_scale, out_size, out_channels, no_trans,
                group_size=1, part_size=None, sample_per_part=4, trans_std=.0):
        ctx.cfg = (no_trans, spatial_scale, out_channels, group_size, out_size,
                   out_size if part_size is None else part_size, sample_per_part, trans_std)
        assert 0.0 <= trans_std <= 1.0
        if not _C.on_device(data):
            raise NotImplementedError
        n = rois.shape[0]
        output = data.new_empty(n, out_channels, out_size, out_size)
        output_count = data.new_empty(n, out_channels, out_size, out_size)
        _C.deform_psroi_pooling_forward(data, rois, offset, output, output_count, *ctx.cfg)
        if data.requires_grad or rois.requires_grad or offset.requires_grad:
            ctx.save_for_backward(data, rois, offset)
        ctx.output_count = output_count
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not _C.on_device(grad_output):
            raise NotImplementedError
        data, rois, offset = ctx.saved_tensors
        grad_input = torch.zeros_like(data)
        grad_offset = torch.zeros_like(offset)
        # `_C` keeps the reference's "out_grad tensor has to be contiguous" check
        # (deform_pool_cuda.cu:71); an expanded upstream gradient (e.g. from `.sum()`) is
        # materialised here instead of failing like the reference's Function does.
        grad_output = grad_output.contiguous()
        _C.deform_psroi_pooling_backward(grad_output, data, rois, offset, ctx.output_count, grad_input,
                                         grad_offset, *ctx.cfg)
        return (grad_input, None, grad_offset, None, None, None, None, None, None, None, None)


deform_roi_pooling = DeformRoIPoolingFunction.apply
