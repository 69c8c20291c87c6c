"""Autograd functions of deformable convolution v1 / modulated v2
(reference layers/dcn/deform_conv_func.py:12-265) over the HIP kernels in `_C`.

Mixed precision: the reference registers nothing with apex for these functions, so under O1 they
ran in whatever dtype reached them.  Here fp32 / fp16 / bf16 tensors are accepted (one dtype per
call; the sampling arithmetic is fp32 inside the kernels, GEMMs accumulate in fp32).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from maskrcnn_benchmark import _C


def _same_dtype(ref, *ts):
    return [t if (t is None or t.dtype == ref.dtype) else t.to(ref.dtype) for t in ts]


class DeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1, im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = groups, deformable_groups, im2col_step
        offset, weight = _same_dtype(input, offset, weight)
        weight = _C._dcn_in(weight)       # contiguous, or channels-last as a model switched to channels-last carries its 4-d parameters
        ctx.save_for_backward(input, offset, weight)
        # a channels-last input gets a channels-last output (the pipeline is channel-fastest inside: no layout launches)
        output = torch.empty(DeformConvFunction._output_size(input, weight, ctx.padding, ctx.dilation, ctx.stride),
                             dtype=input.dtype, device=input.device,
                             memory_format=torch.channels_last if _C.is_channels_last(input) else torch.contiguous_format)
        ctx.bufs_ = [input.new_empty(0), input.new_empty(0)]  # columns, ones (API compatibility)
        if not _C.on_device(input):
            raise NotImplementedError
        step = min(ctx.im2col_step, input.shape[0])
        assert (input.shape[0] % step) == 0, "im2col step must divide batchsize"
        # a layer that will be differentiated keeps its channel-fastest input copy and column matrix for its backward
        # pass (one transpose + one im2col launch less there)
        keep = [] if any(ctx.needs_input_grad[:3]) else None
        _C.deform_conv_forward(input, weight, offset, output, ctx.bufs_[0], ctx.bufs_[1],
                               weight.size(3), weight.size(2), ctx.stride[1], ctx.stride[0],
                               ctx.padding[1], ctx.padding[0], ctx.dilation[1], ctx.dilation[0],
                               ctx.groups, ctx.deformable_groups, step, keep=keep)
        ctx.kept_ = keep[0] if keep else None
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        grad_input = grad_offset = grad_weight = None
        if not _C.on_device(grad_output):
            raise NotImplementedError
        (grad_output,) = _same_dtype(input, grad_output)
        step = min(ctx.im2col_step, input.shape[0])
        assert (input.shape[0] % step) == 0, "im2col step must divide batchsize"
        geom = (weight.size(3), weight.size(2), ctx.stride[1], ctx.stride[0], ctx.padding[1],
                ctx.padding[0], ctx.dilation[1], ctx.dilation[0], ctx.groups, ctx.deformable_groups)
        need_in = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        # one pass for all gradients where the channels-last pipeline serves the shape (_C.deform_conv_backward_all);
        # otherwise the reference's two calls (deform_conv_func.py:87-127)
        fused = _C.deform_conv_backward_all(input, offset, None, weight, grad_output, weight.size(2), weight.size(3),
                                            ctx.padding[0], ctx.padding[1], ctx.stride[0], ctx.stride[1],
                                            ctx.dilation[0], ctx.dilation[1], ctx.groups, ctx.deformable_groups,
                                            need_input=need_in, need_weight=ctx.needs_input_grad[2],
                                            saved=getattr(ctx, "kept_", None))
        ctx.kept_ = None
        if fused is not None:
            return (fused[0], fused[1], fused[3], None, None, None, None, None, None)
        if need_in:
            grad_input = torch.zeros_like(input, memory_format=torch.contiguous_format)
            grad_offset = torch.zeros_like(offset, memory_format=torch.contiguous_format)
            _C.deform_conv_backward_input(input, offset, grad_output, grad_input, grad_offset, weight,
                                          ctx.bufs_[0], *geom, step)
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight, memory_format=torch.contiguous_format)
            _C.deform_conv_backward_parameters(input, offset, grad_output, grad_weight, ctx.bufs_[0],
                                               ctx.bufs_[1], *geom, 1, step)
        return (grad_input, grad_offset, grad_weight, None, None, None, None, None, None)

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        output_size = (input.size(0), weight.size(0))
        for d in range(input.dim() - 2):
            kernel = dilation[d] * (weight.size(d + 2) - 1) + 1
            output_size += ((input.size(d + 2) + 2 * padding[d] - kernel) // stride[d] + 1,)
        if not all(map(lambda s: s > 0, output_size)):
            raise ValueError("convolution input is too small (output would be {})".format(
                "x".join(map(str, output_size))))
        return output_size


class ModulatedDeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                groups=1, deformable_groups=1):
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        ctx.with_bias = bias is not None
        if not ctx.with_bias:
            bias = input.new_empty(1)  # fake tensor
        if not _C.on_device(input):
            raise NotImplementedError
        # decided on the ORIGINAL arguments: a .to(dtype) copy made under no-grad reports requires_grad=False
        needs_grad = weight.requires_grad or mask.requires_grad or offset.requires_grad or input.requires_grad
        offset, mask, weight, bias = _same_dtype(input, offset, mask, weight, bias)
        weight = _C._dcn_in(weight)       # contiguous, or channels-last as a model switched to channels-last carries its 4-d parameters
        if needs_grad:
            ctx.save_for_backward(input, offset, mask, weight, bias)
        output = torch.empty(ModulatedDeformConvFunction._infer_shape(ctx, input, weight), dtype=input.dtype, device=input.device,
                             memory_format=torch.channels_last if _C.is_channels_last(input) else torch.contiguous_format)
        ctx._bufs = [input.new_empty(0), input.new_empty(0)]
        keep = [] if needs_grad else None
        _C.modulated_deform_conv_forward(input if _C.is_channels_last(input) else input.contiguous(), weight, bias, ctx._bufs[0], offset, mask,
                                         output, ctx._bufs[1], weight.shape[2], weight.shape[3],
                                         ctx.stride, ctx.stride, ctx.padding, ctx.padding,
                                         ctx.dilation, ctx.dilation, ctx.groups,
                                         ctx.deformable_groups, ctx.with_bias, keep=keep)
        ctx.kept_ = keep[0] if keep else None
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not _C.on_device(grad_output):
            raise NotImplementedError
        input, offset, mask, weight, bias = ctx.saved_tensors
        (grad_output,) = _same_dtype(input, grad_output)
        fused = _C.deform_conv_backward_all(input, offset, mask, weight, grad_output, weight.shape[2], weight.shape[3],
                                            ctx.padding, ctx.padding, ctx.stride, ctx.stride, ctx.dilation, ctx.dilation,
                                            ctx.groups, ctx.deformable_groups, need_bias=ctx.with_bias,
                                            saved=getattr(ctx, "kept_", None))
        ctx.kept_ = None
        if fused is not None:
            return fused + (None, None, None, None, None)
        grad_input = torch.zeros_like(input, memory_format=torch.contiguous_format)
        grad_offset = torch.zeros_like(offset, memory_format=torch.contiguous_format)
        grad_mask = torch.zeros_like(mask, memory_format=torch.contiguous_format)
        grad_weight = torch.zeros_like(weight, memory_format=torch.contiguous_format)
        grad_bias = torch.zeros_like(bias)
        _C.modulated_deform_conv_backward(input.contiguous(), weight.contiguous(), bias, ctx._bufs[0], offset, mask,
                                          ctx._bufs[1], grad_input, grad_weight, grad_bias,
                                          grad_offset, grad_mask, grad_output, weight.shape[2],
                                          weight.shape[3], ctx.stride, ctx.stride, ctx.padding,
                                          ctx.padding, ctx.dilation, ctx.dilation, ctx.groups,
                                          ctx.deformable_groups, ctx.with_bias)
        if not ctx.with_bias:
            grad_bias = None
        return (grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None, None,
                None)

    @staticmethod
    def _infer_shape(ctx, input, weight):
        n, channels_out = input.size(0), weight.size(0)
        height, width = input.shape[2:4]
        kernel_h, kernel_w = weight.shape[2:4]
        height_out = (height + 2 * ctx.padding - (ctx.dilation * (kernel_h - 1) + 1)) // ctx.stride + 1
        width_out = (width + 2 * ctx.padding - (ctx.dilation * (kernel_w - 1) + 1)) // ctx.stride + 1
        return n, channels_out, height_out, width_out


deform_conv = DeformConvFunction.apply
modulated_deform_conv = ModulatedDeformConvFunction.apply
