"""`maskrcnn_benchmark._C` — the reference's native-operator module, re-exported over the C ABI of
libdetops_gfx950.so (include/detops.h).

Same names, argument order and ownership rules as the reference's pybind module
(reference: maskrcnn_benchmark/csrc/vision.cpp:9-25 and the dispatch headers csrc/*.h):

    nms, roi_align_forward, roi_align_backward, roi_pool_forward, roi_pool_backward,
    sigmoid_focalloss_forward, sigmoid_focalloss_backward,
    deform_conv_forward, deform_conv_backward_input, deform_conv_backward_parameters,
    modulated_deform_conv_forward, modulated_deform_conv_backward

    deform_psroi_pooling_forward, deform_psroi_pooling_backward

PyTorch is plumbing here: it owns device memory and the current HIP
stream; every computation happens in the hand-written gfx950 kernels (GEMMs of the deformable
convolution go through torch.addmm -> hipBLASLt/rocBLAS, as the reference's go through cuBLAS).

There is no CPU path: CPU tensors raise "Not implemented on the CPU" (the reference's own message
for its CUDA-only operators, e.g. csrc/ROIAlign.h:44).
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_of

# ------------------------------------------------------------------------------------------ helpers


def _need_cuda(name, *tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("%s: Not implemented on the CPU (HIP-only build, no fallback)" % name)


def _f32c(name, t):
    if t.dtype != torch.float32:
        raise RuntimeError("%s: expected a float32 tensor, got %s" % (name, t.dtype))
    return t.contiguous()


def on_device(t):
    """True when `t` takes the HIP operators of this module (the model code asks this instead of `t.is_cuda`, so the
    host-emulation tests can drive the same branches with CPU tensors)."""
    return t.is_cuda


def _u8(name, t):
    """bool / uint8 mask -> contiguous uint8 (a bool tensor is reinterpreted, not converted: no launch)."""
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)
    if t.dtype != torch.uint8:
        raise RuntimeError("%s: expected a bool or uint8 mask, got %s" % (name, t.dtype))
    return t.contiguous()


def _on_device(t):
    """Context: `t`'s device is current for the launch (a shared no-op object in the usual case — the tensor lives
    on the current device — so the launch path allocates nothing)."""
    idx = t.device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NOSPAN
    return torch.cuda.device(idx)


class KernelTimer(object):
    """Optional per-entry-point timing with HIP events recorded on the launch stream (the current
    torch stream of the tensor's device).  bench.py installs one to measure the hand-written
    kernels live inside the timed region; `results()` must be called after a device sync.
    High-frequency entry points (FrozenBN: ~100 launches per step) are SAMPLED — every `every`-th call of a name
    gets an event pair, all calls are counted — so that the timer's own host cost stays out of the step."""

    def __init__(self, every_cap=None, strides=None, count_only=False):
        self.pairs = {}
        self.calls = {}
        self.every_cap = every_cap     # upper bound on the sampling stride (short runs: enough samples per name)
        self.strides = strides or {}   # per-name sampling stride (bench.py: from the call counts of a warm-up step, so
        self.count_only = count_only   # that every name gets ~12 event pairs over the timed steps and no more: a timed
                                       # launch costs ~30-100 us of device time — the event records drain the queue)

    def span(self, name, t, every=1):
        """`name`: a string or (format, args) — kept as the key and formatted in results() (no string work per launch)"""
        n = self.calls.get(name, 0)
        self.calls[name] = n + 1
        if self.count_only:
            return _NOSPAN
        stride = self.strides.get(name)
        if stride is not None:
            every = stride
        elif self.every_cap is not None and every > self.every_cap:
            every = self.every_cap
        return _Span(self, name, t) if n % every == 0 else _NOSPAN

    def results(self):
        """name -> (calls, timed launches, total_ms of the timed launches)"""
        out = {}
        for name, pairs in self.pairs.items():
            text = name[0] % name[1] if type(name) is tuple else name
            out[text] = (self.calls.get(name, len(pairs)), len(pairs), sum(a.elapsed_time(b) for a, b in pairs))
        return out


class _Span(object):
    def __init__(self, timer, name, t):
        self.timer, self.name, self.t = timer, name, t

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record(torch.cuda.current_stream(self.t.device))

    def __exit__(self, *exc):
        self.b.record(torch.cuda.current_stream(self.t.device))
        self.timer.pairs.setdefault(self.name, []).append((self.a, self.b))


class _NoSpan(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOSPAN = _NoSpan()
KERNEL_TIMER = None  # set to a KernelTimer to collect timings


def _timed(name, t, every=1):
    """`name`: a string, or (format, args) — formatted only when a timer is installed (the launch path of an
    untimed run does no string work)"""
    if KERNEL_TIMER is None:
        return _NOSPAN
    return KERNEL_TIMER.span(name, t, every)


_ESIZE = {torch.float32: 4, torch.float16: 2, torch.bfloat16: 2}


# ------------------------------------------------------------------------------------------ double precision
# The reference dispatches its operators over float AND double (AT_DISPATCH_FLOATING_TYPES).  float64 tensors take the plain
# kernels of csrc/f64_ops.hip (device) / csrc/cpu_branch.hip (CPU tensors: nms, roi_align_forward); everything tuned in this
# library is fp32.
def _f64(t):
    return t is not None and t.dtype == torch.float64


def _f64c(name, *ts):
    for t in ts:
        if t.dtype != torch.float64:
            raise RuntimeError("%s: all floating-point tensors must be float64 (got %s)" % (name, t.dtype))
    return [t.contiguous() for t in ts]


def _nms_f64(dets, scores, threshold):
    dets, scores = _f64c("nms", dets, scores)
    n = dets.size(0)
    if dets.dim() != 2 or dets.size(1) != 4 or scores.numel() != n:
        raise RuntimeError("nms: expected dets [n,4] and scores [n]")
    if not on_device(dets):
        keep = torch.empty((n,), dtype=torch.long)
        num = torch.zeros((1,), dtype=torch.int32)
        check(lib.detops_nms_cpu_f64(ptr(dets), ptr(scores), n, float(threshold), ptr(keep), ptr(num)), "nms(cpu, f64)")
        return keep[: int(num.item())]
    _need_cuda("nms", dets, scores)
    order = torch.sort(scores, descending=True, stable=True).indices       # ties: ascending index, like the CPU path
    sorted_boxes = dets[order].contiguous()
    kept = torch.empty((n,), dtype=torch.uint8, device=dets.device)
    nbytes = int(lib.detops_nms_sorted_f64_workspace_bytes(n))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dets.device)
    with _on_device(dets):
        check(lib.detops_nms_sorted_f64(ptr(sorted_boxes), n, float(threshold), ptr(kept), ptr(ws), nbytes, stream_of(dets)), "nms(f64)")
    return torch.sort(order[kept.bool()]).values       # ascending original indices = nonzero(suppressed == 0)


def _roi_align_forward_f64(input, rois, spatial_scale, ph, pw, sr):
    input, rois = _f64c("roi_align_forward", input, rois)
    N, C, H, W = input.shape
    K = rois.size(0)
    out = torch.empty((K, C, ph, pw), dtype=torch.float64, device=input.device)
    if out.numel() == 0:
        return out
    if not on_device(input):
        check(lib.detops_roi_align_forward_cpu_f64(ptr(input), ptr(rois), ptr(out), N, C, H, W, K, ph, pw, float(spatial_scale), int(sr)),
              "roi_align_forward(cpu, f64)")
        return out
    _need_cuda("roi_align_forward", input, rois)
    with _on_device(input):
        check(lib.detops_roi_align_forward_f64(ptr(input), ptr(rois), ptr(out), N, C, H, W, K, ph, pw, float(spatial_scale), int(sr),
                                               stream_of(input)), "roi_align_forward(f64)")
    return out


def _roi_align_backward_f64(grad, rois, spatial_scale, ph, pw, N, C, H, W, sr):
    _need_cuda("roi_align_backward", grad, rois)
    grad, rois = _f64c("roi_align_backward", grad, rois)
    gin = torch.empty((N, C, H, W), dtype=torch.float64, device=grad.device)
    with _on_device(grad):
        check(lib.detops_roi_align_backward_f64(ptr(grad), ptr(rois), ptr(gin), N, C, H, W, rois.size(0), ph, pw, float(spatial_scale),
                                                int(sr), 1, stream_of(grad)), "roi_align_backward(f64)")
    return gin


def _roi_pool_forward_f64(input, rois, spatial_scale, ph, pw):
    _need_cuda("roi_pool_forward", input, rois)
    input, rois = _f64c("roi_pool_forward", input, rois)
    N, C, H, W = input.shape
    K = rois.size(0)
    out = torch.empty((K, C, ph, pw), dtype=torch.float64, device=input.device)
    argmax = torch.zeros((K, C, ph, pw), dtype=torch.int32, device=input.device)
    if out.numel():
        with _on_device(input):
            check(lib.detops_roi_pool_forward_f64(ptr(input), ptr(rois), ptr(out), ptr(argmax), N, C, H, W, K, ph, pw, float(spatial_scale),
                                                  stream_of(input)), "roi_pool_forward(f64)")
    return out, argmax


def _roi_pool_backward_f64(grad, rois, argmax, ph, pw, N, C, H, W):
    _need_cuda("roi_pool_backward", grad, rois, argmax)
    grad, rois = _f64c("roi_pool_backward", grad, rois)
    argmax = argmax.contiguous()
    gin = torch.empty((N, C, H, W), dtype=torch.float64, device=grad.device)
    with _on_device(grad):
        check(lib.detops_roi_pool_backward_f64(ptr(grad), ptr(rois), ptr(argmax), ptr(gin), N, C, H, W, rois.size(0), ph, pw, 1,
                                               stream_of(grad)), "roi_pool_backward(f64)")
    return gin


def _focal_f64(logits, targets, d_losses, num_classes, gamma, alpha):
    _need_cuda("sigmoid_focalloss", logits, targets)
    (logits,) = _f64c("sigmoid_focalloss", logits)
    if logits.dim() != 2:
        raise RuntimeError("logits should be NxClass")
    if targets.dim() != 1:
        raise RuntimeError("targets should be Nx1")
    R, Cc = logits.shape
    if Cc != num_classes or targets.numel() != R:
        raise RuntimeError("sigmoid_focalloss: inconsistent shapes")
    t32 = targets.to(torch.int32).contiguous()
    out = torch.empty_like(logits)
    with _on_device(logits):
        if d_losses is None:
            check(lib.detops_sigmoid_focal_loss_forward_f64(ptr(logits), ptr(t32), ptr(out), R, Cc, float(gamma), float(alpha),
                                                            stream_of(logits)), "sigmoid_focalloss_forward(f64)")
        else:
            (d_losses,) = _f64c("sigmoid_focalloss_backward", d_losses)
            check(lib.detops_sigmoid_focal_loss_backward_f64(ptr(logits), ptr(t32), ptr(d_losses), ptr(out), R, Cc, float(gamma),
                                                             float(alpha), stream_of(logits)), "sigmoid_focalloss_backward(f64)")
    return out


# ------------------------------------------------------------------------------------------ NMS
def nms(dets, scores, threshold):
    """reference csrc/nms.h:10-28: dets [n,4] xyxy, scores [n] -> int64 kept indices, ascending.
    CPU semantics (IoU >= threshold suppresses; csrc/cpu/nms_cpu.cpp:60)."""
    if dets.numel() == 0:  # nms.h:17-18 returns an empty CPU long tensor
        return torch.empty((0,), dtype=torch.long, device="cpu")
    if _f64(dets):
        return _nms_f64(dets, scores, threshold)
    on_cpu = not on_device(dets) and not on_device(scores)
    if not on_cpu:
        _need_cuda("nms", dets, scores)
    dets = _f32c("nms", dets)
    scores = _f32c("nms", scores)
    n = dets.size(0)
    if dets.dim() != 2 or dets.size(1) != 4 or scores.numel() != n:
        raise RuntimeError("nms: expected dets [n,4] and scores [n]")
    if on_cpu:  # the reference dispatches CPU tensors to nms_cpu (csrc/nms.h:26-27); host code of the same library
        keep = torch.empty((n,), dtype=torch.long)
        num = torch.zeros((1,), dtype=torch.int32)
        check(lib.detops_nms_cpu_f32(ptr(dets), ptr(scores), n, float(threshold), ptr(keep), ptr(num)), "nms(cpu)")
        return keep[: int(num.item())]
    keep = torch.empty((n,), dtype=torch.long, device=dets.device)
    num = torch.empty((1,), dtype=torch.int32, device=dets.device)
    ws_bytes = lib.detops_nms_workspace_bytes(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dets.device)
    with _on_device(dets):
        check(lib.detops_nms_f32(ptr(dets), ptr(scores), n, float(threshold), ptr(keep), ptr(num),
                                 ptr(ws), ws_bytes, stream_of(dets)), "nms")
    k = int(num.item())  # the variable-length return value forces one readback
    if k < 0:
        raise RuntimeError("nms: a device-side wait of the single-launch kernel timed out (num_keep = -1): the GPU's "
                           "compute units were held by other work for seconds; no result was produced")
    return keep[:k]


_NMS_STATUS = {}


def nms_status(device):
    """The sticky device word (int32 [1]) the segmented NMS entry points pass to the library: incremented once for every
    segment the single-launch kernel gave up on and the repair launch redid (include/detops.h,
    detops_nms_batched_status_f32).  One per device, never cleared by the library, never read by the launch path."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else 0))
    t = _NMS_STATUS.get(key)
    if t is None:
        t = _NMS_STATUS[key] = torch.zeros((1,), dtype=torch.int32, device=device)
    return t


def nms_repaired_segments(device=None, reset=False):
    """Host-side read of the status words (a device->host copy: call it where a read-back happens anyway, e.g. with the
    logged losses).  Returns the number of NMS segments redone by the repair launch since the last reset."""
    total = 0
    for (typ, idx), t in list(_NMS_STATUS.items()):
        if device is not None and torch.device(device).type != typ:
            continue
        total += int(t.item())
        if reset:
            t.zero_()
    return total


def nms_batched(boxes, scores, seg_offsets, max_n, threshold):
    """Sync-free segmented NMS (extension; not in the reference `_C`).
    boxes [T,4], scores [T], seg_offsets int32 [S+1] (device).  Returns (keep [T] int64 with each
    segment's LOCAL kept indices packed at its offset, num_keep [S] int32) — all on device.
    A segment the single launch gives up on (starved producer workgroup) is redone by the library's repair launch on the
    same stream and counted in `nms_status(device)`: callers may ignore num_keep without ever losing a segment."""
    _need_cuda("nms_batched", boxes, scores, seg_offsets)
    boxes = _f32c("nms_batched", boxes)
    scores = _f32c("nms_batched", scores)
    S = seg_offsets.numel() - 1
    keep = torch.empty((boxes.size(0),), dtype=torch.long, device=boxes.device)
    num = torch.zeros((max(S, 0),), dtype=torch.int32, device=boxes.device)
    if S <= 0 or boxes.size(0) == 0:
        return keep, num
    seg_offsets = seg_offsets.to(torch.int32).contiguous()
    ws_bytes = lib.detops_nms_batched_workspace_bytes(S, int(max_n))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=boxes.device)
    with _on_device(boxes):
        check(lib.detops_nms_batched_status_f32(ptr(boxes), ptr(scores), ptr(seg_offsets), S, int(max_n),
                                                float(threshold), ptr(keep), None, ptr(num),
                                                ptr(nms_status(boxes.device)), ptr(ws), ws_bytes,
                                                stream_of(boxes)), "nms_batched")
    return keep, num


def nms_batched_mask(boxes, scores, seg_offsets, max_n, threshold):
    """Sync-free segmented NMS, dense form (extension): returns (keep_mask [T] bool,
    num_keep [S] int32), keep_mask in the ORIGINAL row order of `boxes`."""
    _need_cuda("nms_batched_mask", boxes, scores, seg_offsets)
    boxes = _f32c("nms_batched_mask", boxes)
    scores = _f32c("nms_batched_mask", scores)
    S = seg_offsets.numel() - 1
    mask = torch.zeros((boxes.size(0),), dtype=torch.uint8, device=boxes.device)
    num = torch.zeros((max(S, 0),), dtype=torch.int32, device=boxes.device)
    if S <= 0 or boxes.size(0) == 0:
        return mask.bool(), num
    if seg_offsets.dtype != torch.int32:
        seg_offsets = seg_offsets.to(torch.int32)
    seg_offsets = seg_offsets.contiguous()
    ws_bytes = lib.detops_nms_batched_workspace_bytes(S, int(max_n))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=boxes.device)
    with _on_device(boxes), _timed(("nms_batched[S=%d,max_n=%d]", (S, int(max_n))), boxes):
        check(lib.detops_nms_batched_status_f32(ptr(boxes), ptr(scores), ptr(seg_offsets), S, int(max_n),
                                                float(threshold), None, ptr(mask), ptr(num),
                                                ptr(nms_status(boxes.device)), ptr(ws), ws_bytes,
                                                stream_of(boxes)), "nms_batched_mask")
    return mask.view(torch.bool), num


# ------------------------------------------------------------------------------------------ FPN top-down step
class _UpsampleAdd(torch.autograd.Function):
    """lateral + nearest_upsample(top) in one pass (csrc/fpn_topdown.hip; reference modeling/backbone/fpn.py:59-64).
    A channels-last lateral selects the NHWC kernels and returns channels-last tensors in both directions."""

    @staticmethod
    def forward(ctx, lateral, top):
        nhwc = is_channels_last(lateral)
        fmt = torch.channels_last if nhwc else torch.contiguous_format
        lateral, top = lateral.contiguous(memory_format=fmt), top.contiguous(memory_format=fmt)
        N, C, H, W = lateral.shape
        h, w = int(top.shape[2]), int(top.shape[3])
        out = torch.empty_like(lateral)
        with _on_device(lateral), _timed(("fpn_topdown_fwd[n=%d,e=%d]", (lateral.numel(), _ESIZE[lateral.dtype])), lateral):
            if nhwc:
                check(lib.detops_fpn_topdown_forward_nhwc(ptr(lateral), ptr(top), ptr(out), _lib.DTYPE_CODE[lateral.dtype], N, C, H, W,
                                                          h, w, stream_of(lateral)), "fpn_topdown_forward_nhwc")
            else:
                check(lib.detops_fpn_topdown_forward(ptr(lateral), ptr(top), ptr(out), _lib.DTYPE_CODE[lateral.dtype], N * C, H, W, h, w,
                                                     stream_of(lateral)), "fpn_topdown_forward")
        ctx.shape = (N, C, H, W, h, w)
        ctx.nhwc = nhwc
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, H, W, h, w = ctx.shape
        fmt = torch.channels_last if ctx.nhwc else torch.contiguous_format
        g = g.contiguous(memory_format=fmt)
        gtop = None
        if ctx.needs_input_grad[1]:
            gtop = torch.empty((N, C, h, w), dtype=g.dtype, device=g.device, memory_format=fmt)
            with _on_device(g), _timed(("fpn_topdown_bwd[n=%d,e=%d]", (g.numel(), _ESIZE[g.dtype])), g):
                if ctx.nhwc:
                    check(lib.detops_fpn_topdown_backward_nhwc(ptr(g), ptr(gtop), _lib.DTYPE_CODE[g.dtype], N, C, H, W, h, w,
                                                               stream_of(g)), "fpn_topdown_backward_nhwc")
                else:
                    check(lib.detops_fpn_topdown_backward(ptr(g), ptr(gtop), _lib.DTYPE_CODE[g.dtype], N * C, H, W, h, w, stream_of(g)),
                          "fpn_topdown_backward")
        return (g if ctx.needs_input_grad[0] else None), gtop


def fpn_topdown(lateral, top):
    """out = lateral + F.interpolate(top, size=lateral.shape[-2:], mode="nearest") (extension; the reference composes it from
    two ATen ops, modeling/backbone/fpn.py:59-64).  [N, C, H, W] + [N, C, h, w], one dtype (fp32 / fp16 / bf16)."""
    _need_cuda("fpn_topdown", lateral, top)
    if lateral.dtype != top.dtype or lateral.dtype not in _lib.DTYPE_CODE:
        raise RuntimeError("fpn_topdown: lateral and top must share one dtype (fp32 / fp16 / bf16)")
    if lateral.dim() != 4 or top.dim() != 4 or lateral.shape[:2] != top.shape[:2]:
        raise RuntimeError("fpn_topdown: expected [N, C, H, W] and [N, C, h, w]")
    return _UpsampleAdd.apply(lateral, top)


# ------------------------------------------------------------------------------------------ data-parallel bucket kernels
_PACK_MAX = [0]


def pack_into(dst, tensors, offsets, stream=None):
    """dst (flat fp32) [offsets[i] : offsets[i] + tensors[i].numel()] = tensors[i] — the gradients of a bucket in ONE launch
    per 48 tensors (csrc/optim.hip; reference: DistributedDataParallel's bucket copies, tools/train_net.py:45-54).
    `stream`: a torch.cuda stream (default: the current one)."""
    _need_cuda("pack_into", dst, *tensors)
    if not _PACK_MAX[0]:
        _PACK_MAX[0] = int(lib.detops_pack_max_tensors())
    if dst.dtype != torch.float32 or not dst.is_contiguous():
        raise RuntimeError("pack_into: a contiguous fp32 destination is required")
    st = stream_of(dst) if stream is None else stream.cuda_stream
    n_all = len(tensors)
    with _on_device(dst):
        for i0 in range(0, n_all, _PACK_MAX[0]):
            part = list(tensors[i0:i0 + _PACK_MAX[0]])
            n = len(part)
            srcs = (ctypes.c_void_p * n)()
            cnts = (ctypes.c_int64 * n)()
            offs = (ctypes.c_int64 * n)()
            for j, t in enumerate(part):
                if t.dtype != torch.float32 or t.device != dst.device:
                    raise RuntimeError("pack_into: fp32 tensors on the destination's device are required")
                if not t.is_contiguous():
                    t = t.contiguous()
                    part[j] = t          # keep the copy alive until the launch is enqueued
                srcs[j], cnts[j], offs[j] = t.data_ptr(), t.numel(), int(offsets[i0 + j])
                if offs[j] < 0 or offs[j] + cnts[j] > dst.numel():
                    raise RuntimeError("pack_into: slice %d leaves the destination" % (i0 + j))
            check(lib.detops_pack_f32(srcs, cnts, offs, n, ptr(dst), st), "pack_f32")
    return dst


def sgd_momentum_flat_(params, grads, momentum_buf, split, lr_weights, wd_weights, lr_biases, wd_biases, momentum, stream=None):
    """torch.optim.SGD(momentum, dampening 0) over a whole flat bucket in one pass (csrc/optim.hip; reference
    solver/build.py:7-20 + engine/trainer.py:98 `optimizer.step()`): elements [0, split) take the weights' (lr, wd), the
    rest the biases'.  In place on `params` and `momentum_buf`."""
    _need_cuda("sgd_momentum_flat_", params, grads, momentum_buf)
    for t in (params, grads, momentum_buf):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != params.numel():
            raise RuntimeError("sgd_momentum_flat_: three contiguous fp32 arrays of one length are required")
    st = stream_of(params) if stream is None else stream.cuda_stream
    with _on_device(params):
        check(lib.detops_sgd_momentum_flat_f32(ptr(params), ptr(grads), ptr(momentum_buf), params.numel(), int(split),
                                               float(lr_weights), float(wd_weights), float(lr_biases), float(wd_biases),
                                               float(momentum), st), "sgd_momentum_flat")
    return params


# ------------------------------------------------------------------------------------------ RPN loss
class _RpnLoss(torch.autograd.Function):
    """Fused RPN loss (extension; reference modeling/rpn/loss.py:92-127): one launch evaluates both losses from the
    per-level head outputs and stores d(loss sum)/d(logit) in the same layouts; backward = one scaling launch."""

    @staticmethod
    def forward(ctx, anchors, matched, pos, neg, gt, beta, weights, A, L, *heads):
        obj, box = list(heads[:L]), list(heads[L:])
        N = obj[0].size(0)
        T, M = anchors.size(0), gt.size(1)
        gobj = [torch.empty_like(t) for t in obj]
        gbox = [torch.empty_like(t) for t in box]
        out3 = torch.empty((3,), dtype=torch.float32, device=anchors.device)
        nbytes = int(lib.detops_rpn_loss_workspace_bytes())
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=anchors.device)
        Hs = (ctypes.c_int * L)(*[t.size(2) for t in obj])
        Ws = (ctypes.c_int * L)(*[t.size(3) for t in obj])
        arr = lambda ts: (ctypes.c_void_p * L)(*[t.data_ptr() for t in ts])
        w4 = (ctypes.c_float * 4)(*[float(w) for w in weights])
        with _on_device(anchors), _timed(("rpn_loss[N=%d,T=%d]", (N, T)), anchors):
            check(lib.detops_rpn_loss_f32(arr(obj), arr(box), Hs, Ws, L, int(A), ptr(anchors), ptr(matched), ptr(pos),
                                          ptr(neg), ptr(gt), N, M, T, float(beta), w4, arr(gobj), arr(gbox), ptr(out3),
                                          ptr(ws), nbytes, stream_of(anchors)), "rpn_loss")
        ctx.grads, ctx.out3, ctx.meta = (gobj, gbox), out3, (L, int(A), N, T, Hs, Ws)
        ctx.mark_non_differentiable(out3)
        return out3[0], out3[1], out3

    @staticmethod
    def backward(ctx, g_obj, g_box, _unused):
        gobj, gbox = ctx.grads
        L, A, N, T, Hs, Ws = ctx.meta
        if gobj is None:
            raise RuntimeError("rpn_loss: backward called twice (the stored gradients are scaled in place)")
        up_o = (g_obj if g_obj is not None else torch.zeros((), device=ctx.out3.device)).reshape(1).float().contiguous()
        up_b = (g_box if g_box is not None else torch.zeros((), device=ctx.out3.device)).reshape(1).float().contiguous()
        arr = lambda ts: (ctypes.c_void_p * L)(*[t.data_ptr() for t in ts])
        with _on_device(ctx.out3):
            check(lib.detops_rpn_loss_backward_f32(arr(gobj), arr(gbox), Hs, Ws, L, A, N, T, ptr(up_o), ptr(up_b),
                                                   ptr(ctx.out3[2:]), stream_of(ctx.out3)), "rpn_loss_backward")
        ctx.grads = (None, None)
        return (None,) * 9 + tuple(gobj) + tuple(gbox)


def rpn_loss(objectness, box_regression, anchors, matched_idxs, pos_mask, neg_mask, gt_boxes, beta, weights):
    """objectness / box_regression: per-level lists [N, A, H, W] / [N, 4A, H, W] (fp32, contiguous); anchors [T, 4];
    matched_idxs [N, T] int64; pos_mask / neg_mask [N, T] bool; gt_boxes [N, M, 4] -> (objectness_loss, box_loss)."""
    _need_cuda("rpn_loss", anchors, matched_idxs, pos_mask, neg_mask, gt_boxes, *objectness, *box_regression)
    L = len(objectness)
    A = box_regression[0].size(1) // 4
    heads = [_f32c("rpn_loss", t) for t in list(objectness) + list(box_regression)]
    as_u8 = lambda m: (m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)).contiguous()
    lo, lb, _ = _RpnLoss.apply(_f32c("rpn_loss", anchors), matched_idxs.to(torch.int64).contiguous(), as_u8(pos_mask),
                               as_u8(neg_mask), _f32c("rpn_loss", gt_boxes), float(beta), tuple(weights), A, L, *heads)
    return lo, lb


# ------------------------------------------------------------------------------------------ ROI-head losses
class _HeadLoss(torch.autograd.Function):
    """Shared autograd glue of the fused ROI-head losses (csrc/head_loss.hip): forward stores d loss / d input next to the
    values, backward scales the stored gradients by the upstream scalars in one launch."""

    @staticmethod
    def forward(ctx, kind, aux, *inputs):
        grads, out = _HEAD_LOSS[kind](aux, *inputs)
        ctx.grads, ctx.n_in = grads, len(inputs)
        return tuple(out[i] for i in range(out.numel()))

    @staticmethod
    def backward(ctx, *ups):
        grads = ctx.grads
        if grads is None:
            raise RuntimeError("head loss: backward called twice (the stored gradients are scaled in place)")
        dev = grads[0].device
        up = [(u if u is not None else torch.zeros((), device=dev)).reshape(1).float().contiguous() for u in ups]
        a, b = grads[0], (grads[1] if len(grads) > 1 else None)
        with _on_device(a):
            check(lib.detops_head_loss_backward_f32(ptr(a), a.numel(), ptr(up[0]), ptr(b), b.numel() if b is not None else 0,
                                                    ptr(up[1]) if b is not None else None, stream_of(a)), "head_loss_backward")
        ctx.grads = None
        return (None, None) + tuple(grads) + (None,) * (ctx.n_in - len(grads))


def _fastrcnn_loss_forward(aux, class_logits, box_regression, labels, regression_targets):
    cls_agnostic, beta = aux
    R, C = class_logits.shape
    D = box_regression.size(1)
    dev = class_logits.device
    gl, gb = torch.empty_like(class_logits), torch.empty_like(box_regression)
    out = torch.empty((2,), dtype=torch.float32, device=dev)
    nbytes = int(lib.detops_fastrcnn_loss_workspace_bytes(R))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    with _on_device(class_logits), _timed(("fastrcnn_loss[R=%d,C=%d]", (R, C)), class_logits):
        check(lib.detops_fastrcnn_loss_f32(ptr(class_logits), ptr(box_regression), ptr(labels), ptr(regression_targets), R, C, D,
                                           int(bool(cls_agnostic)), float(beta), ptr(gl), ptr(gb), ptr(out), ptr(ws), nbytes,
                                           stream_of(class_logits)), "fastrcnn_loss")
    return (gl, gb), out


def _mask_loss_forward(aux, mask_logits, labels, mask_targets):
    P, C, M, _ = mask_logits.shape
    dev = mask_logits.device
    g = torch.empty_like(mask_logits)
    out = torch.empty((1,), dtype=torch.float32, device=dev)
    nbytes = int(lib.detops_mask_loss_workspace_bytes(P))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    with _on_device(mask_logits), _timed(("mask_loss[P=%d,C=%d,M=%d]", (P, C, M)), mask_logits):
        check(lib.detops_mask_loss_f32(ptr(mask_logits), ptr(labels), ptr(mask_targets), P, C, M, ptr(g), ptr(out), ptr(ws), nbytes,
                                       stream_of(mask_logits)), "mask_loss")
    return (g,), out


_HEAD_LOSS = {"fastrcnn": _fastrcnn_loss_forward, "mask": _mask_loss_forward}


def fastrcnn_loss(class_logits, box_regression, labels, regression_targets, cls_agnostic=False, beta=1.0):
    """Box-head loss, value + gradient in one pass (extension; reference roi_heads/box_head/loss.py:140-193):
    class_logits [R,C], box_regression [R,4C] (or [R,>=8] class-agnostic), labels [R] int64 (-1 = not sampled),
    regression_targets [R,4] -> (classification_loss, box_loss), both / max(#sampled, 1)."""
    _need_cuda("fastrcnn_loss", class_logits, box_regression, labels, regression_targets)
    class_logits, box_regression = _f32c("fastrcnn_loss", class_logits), _f32c("fastrcnn_loss", box_regression)
    regression_targets = _f32c("fastrcnn_loss", regression_targets)
    labels = labels.to(torch.int64).contiguous()
    R, C = class_logits.shape
    D = box_regression.size(1)
    if R == 0 or labels.shape != (R,) or regression_targets.shape != (R, 4) or box_regression.size(0) != R \
            or (D < 8 if cls_agnostic else D != 4 * C):
        raise ValueError("fastrcnn_loss: inconsistent arguments")
    return _HeadLoss.apply("fastrcnn", (bool(cls_agnostic), float(beta)), class_logits, box_regression, labels,
                           regression_targets)


def mask_loss(mask_logits, labels, mask_targets):
    """Mask-head loss, value + gradient in one pass (extension; reference roi_heads/mask_head/loss.py:113-143):
    mask_logits [P,C,M,M], labels [P] int64 (> 0 = positive), mask_targets [P,M,M] -> mean BCE-with-logits of the
    positives' class planes (0 without positives)."""
    _need_cuda("mask_loss", mask_logits, labels, mask_targets)
    mask_logits, mask_targets = _f32c("mask_loss", mask_logits), _f32c("mask_loss", mask_targets)
    labels = labels.to(torch.int64).contiguous()
    P, C, M, M2 = mask_logits.shape
    if P == 0 or M != M2 or labels.shape != (P,) or mask_targets.shape != (P, M, M):
        raise ValueError("mask_loss: inconsistent arguments")
    return _HeadLoss.apply("mask", None, mask_logits, labels, mask_targets)[0]


# ------------------------------------------------------------------------------------------ target assignment
def match_boxes(gt_boxes, gt_valid, boxes, high_threshold, low_threshold, allow_low_quality_matches):
    """Fused IoU + Matcher (extension; reference structures/boxlist_ops.py:53-89 + modeling/matcher.py:42-112):
    gt_boxes [N,M,4], gt_valid [N,M] bool, boxes [K,4] (shared) or [N,K,4] -> matched_idxs [N,K] int64."""
    _need_cuda("match_boxes", gt_boxes, gt_valid, boxes)
    gt_boxes = _f32c("match_boxes", gt_boxes)
    boxes = _f32c("match_boxes", boxes)
    valid = _u8("match_boxes", gt_valid)
    N, M = gt_boxes.shape[:2]
    batched = boxes.dim() == 3
    K = boxes.shape[-2]
    out = torch.empty((N, K), dtype=torch.int64, device=boxes.device)
    if N == 0 or K == 0:
        return out
    nbytes = int(lib.detops_match_boxes_workspace_bytes(N, M))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=boxes.device)
    with _on_device(boxes), _timed(("match_boxes[N=%d,M=%d,K=%d]", (N, M, K)), boxes):
        check(lib.detops_match_boxes_f32(ptr(gt_boxes), ptr(valid), ptr(boxes), int(batched), N, M, K,
                                         float(high_threshold), float(low_threshold), int(bool(allow_low_quality_matches)),
                                         ptr(out), ptr(ws), nbytes, stream_of(boxes)), "match_boxes")
    return out


_SAMPLER_CALLS = [0]


def _next_sampler_seed(device):
    """64-bit seed of the next sampler call, drawn from the DEVICE generator's (seed, Philox offset) — host-side state,
    no device op: the stream follows torch.cuda.manual_seed and is saved / restored with the generator state
    (torch.cuda.get_rng_state in a checkpoint), like the `torch.randperm` the reference's sampler draws from the same
    generator.  The offset is advanced by 4 per call (one Philox block)."""
    try:
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        base, off = gen.initial_seed(), gen.get_offset()
        gen.set_offset(off + 4)
    except (AttributeError, RuntimeError, IndexError):   # a torch without generator offsets: per-process counter
        _SAMPLER_CALLS[0] += 1
        base, off = torch.initial_seed(), _SAMPLER_CALLS[0]
    return (base * 0x9E3779B97F4A7C15 + off * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


# Set by engine/graph_step.py while a training step is captured into / replayed from a HIP graph: an int64 [1] device tensor that
# the captured step increments; the samplers mix it into their (frozen-at-capture) seed when they RUN.
GRAPH_SEED = None


def sample_labels(labels, batch_size_per_image, max_positives, with_list=False, seed=None):
    """BalancedPositiveNegativeSampler on device (extension; reference
    modeling/balanced_positive_negative_sampler.py:19-68): labels [N,n] float32 or int64 ->
    (pos_mask, neg_mask [N,n] bool[, idx [N,B] int64, valid [N,B] bool]).  The random subset is a pure function
    of (labels, seed); the default seed is drawn from the device generator's seed and Philox offset (no device op)."""
    _need_cuda("sample_labels", labels)
    if labels.dtype not in (torch.float32, torch.int64):
        labels = labels.to(torch.int64)
    labels = labels.contiguous()
    N, n = labels.shape
    B = int(batch_size_per_image)
    if seed is None:
        seed = _next_sampler_seed(labels.device)
    pos = torch.empty((N, n), dtype=torch.uint8, device=labels.device)
    neg = torch.empty((N, n), dtype=torch.uint8, device=labels.device)
    idx = torch.empty((N, B), dtype=torch.int64, device=labels.device) if with_list else None
    val = torch.empty((N, B), dtype=torch.uint8, device=labels.device) if with_list else None
    if N and n:
        nbytes = int(lib.detops_sample_labels_workspace_bytes(N, B))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=labels.device)
        with _on_device(labels), _timed(("sample_labels[N=%d,n=%d,B=%d]", (N, n, B)), labels):
            check(lib.detops_sample_labels_dseed(ptr(labels), 0 if labels.dtype == torch.float32 else 1, N, n, B,
                                                 int(max_positives), ctypes.c_uint64(seed), ptr(GRAPH_SEED), ptr(pos), ptr(neg),
                                                 ptr(idx), ptr(val), ptr(ws), nbytes, stream_of(labels)), "sample_labels")
    elif with_list:
        idx.zero_()
        val.zero_()
    if with_list:
        return pos.view(torch.bool), neg.view(torch.bool), idx, val.view(torch.bool)
    return pos.view(torch.bool), neg.view(torch.bool)


def mask_targets(masks, mask_index, boxes, discretization_size):
    """Mask-head targets (extension; reference roi_heads/mask_head/loss.py:11-42): masks [G,H,W] uint8 / bool /
    float32, mask_index [P] int64, boxes [P,4] xyxy -> [P,M,M] float32, bit-equal to the reference's CPU path."""
    _need_cuda("mask_targets", masks, mask_index, boxes)
    boxes = _f32c("mask_targets", boxes)
    if masks.dtype == torch.bool:
        code, m = 2, masks.contiguous().view(torch.uint8)
    elif masks.dtype == torch.uint8:
        code, m = 0, masks.contiguous()
    elif not masks.dtype.is_floating_point:
        # any other integer dtype: the reference's `.type_as(self.masks)` truncates the interpolated value for every
        # non-floating mask (segmentation_mask.py:111-158); binary masks are 0 / 1, so uint8 is the same computation
        code, m = 0, masks.to(torch.uint8).contiguous()
    else:
        code, m = 1, masks.to(torch.float32).contiguous()
    mask_index = mask_index.to(torch.int64).contiguous()
    G, H, W = m.shape
    P, M = boxes.size(0), int(discretization_size)
    out = torch.empty((P, M, M), dtype=torch.float32, device=boxes.device)
    if P:
        with _on_device(boxes), _timed(("mask_targets[P=%d,M=%d]", (P, M)), boxes):
            check(lib.detops_mask_targets(ptr(m), code, ptr(mask_index), ptr(boxes), G, H, W, P, M, ptr(out),
                                          stream_of(boxes)), "mask_targets")
    return out


def match_labels(matched, gt_labels=None, valid=None, dtype=torch.int64):
    """Matcher output [N,K] -> labels in one launch (extension; reference modeling/rpn/loss.py:70-88,
    roi_heads/box_head/loss.py:56-72): matched >= 0 -> gt_labels[n, matched] (1 when gt_labels is None), -1 -> 0,
    -2 -> -1, and -1 where `valid` [N,K] is false.  dtype float32 or int64."""
    _need_cuda("match_labels", matched, gt_labels, valid)
    if matched.dtype != torch.int64 or matched.dim() != 2 or dtype not in (torch.float32, torch.int64):
        raise ValueError("match_labels: matched must be [N, K] int64 and dtype float32 / int64")
    matched = matched.contiguous()
    N, K = matched.shape
    M = 0
    if gt_labels is not None:
        gt_labels = gt_labels.to(torch.int64).contiguous()
        M = gt_labels.size(1)
        if gt_labels.size(0) != N:
            raise ValueError("match_labels: gt_labels must be [N, M]")
    if valid is not None:
        valid = _u8("match_labels", valid)
        if valid.shape != matched.shape:
            raise ValueError("match_labels: valid must be [N, K]")
    out = torch.empty((N, K), dtype=dtype, device=matched.device)
    if N * K:
        with _on_device(matched), _timed(("match_labels[N=%d,K=%d]", (N, K)), matched):
            check(lib.detops_match_labels(ptr(matched), ptr(gt_labels) if gt_labels is not None else None,
                                          ptr(valid) if valid is not None else None, N, K, M,
                                          0 if dtype == torch.float32 else 1, ptr(out), stream_of(matched)), "match_labels")
    return out


def roi_head_targets(boxes, matched, gt_boxes, gt_labels, valid, idx, slot_valid, objectness, weights):
    """The box head's sampled slots in one launch (extension; reference roi_heads/box_head/loss.py:56-110):
    boxes [N,K,4], matched [N,K], gt_boxes [N,M,4], gt_labels [N,M], valid [N,K] | None, idx / slot_valid [N,B],
    objectness [N,K] | None -> (boxes [N,B,4], labels [N,B] int64, regression_targets [N,B,4], matched [N,B],
    objectness [N,B] | None)."""
    _need_cuda("roi_head_targets", boxes, matched, gt_boxes, gt_labels, valid, idx, slot_valid, objectness)
    boxes = _f32c("roi_head_targets", boxes)
    gt_boxes = _f32c("roi_head_targets", gt_boxes)
    matched, idx = matched.contiguous(), idx.contiguous()
    gt_labels = gt_labels.to(torch.int64).contiguous()
    slot_valid = _u8("roi_head_targets", slot_valid)
    N, K = matched.shape
    M, B = gt_boxes.size(1), idx.size(1)
    if matched.dtype != torch.int64 or idx.dtype != torch.int64 or boxes.shape != (N, K, 4) or gt_boxes.shape != (N, M, 4) \
            or gt_labels.shape != (N, M) or idx.shape != (N, B) or slot_valid.shape != (N, B):
        raise ValueError("roi_head_targets: inconsistent arguments")
    if valid is not None:
        valid = _u8("roi_head_targets", valid)
        if valid.shape != (N, K):
            raise ValueError("roi_head_targets: valid must be [N, K]")
    if objectness is not None:
        objectness = _f32c("roi_head_targets", objectness)
        if objectness.shape != (N, K):
            raise ValueError("roi_head_targets: objectness must be [N, K]")
    dev = boxes.device
    ob = torch.empty((N, B, 4), dtype=torch.float32, device=dev)
    oreg = torch.empty((N, B, 4), dtype=torch.float32, device=dev)
    ol = torch.empty((N, B), dtype=torch.int64, device=dev)
    om = torch.empty((N, B), dtype=torch.int64, device=dev)
    oo = torch.empty((N, B), dtype=torch.float32, device=dev) if objectness is not None else None
    if N * B:
        if K == 0 or M == 0:
            raise ValueError("roi_head_targets: no proposals / ground truth rows to take the slots from")
        wx, wy, ww, wh = (float(w) for w in weights)
        with _on_device(boxes), _timed(("roi_head_targets[N=%d,B=%d]", (N, B)), boxes):
            check(lib.detops_roi_head_targets_f32(
                ptr(boxes), ptr(matched), ptr(gt_boxes), ptr(gt_labels), ptr(valid) if valid is not None else None, ptr(idx),
                ptr(slot_valid), ptr(objectness) if objectness is not None else None, N, K, M, B, wx, wy, ww, wh, ptr(ob),
                ptr(ol), ptr(oreg), ptr(om), ptr(oo) if oo is not None else None, stream_of(boxes)), "roi_head_targets")
    return ob, ol, oreg, om, oo


def rpn_decode(box_regression, topk_idx, topk_scores, anchors, image_hw, weights, bbox_xform_clip, min_size,
               boxes, scores, col, nms_boxes, nms_scores, ok, off):
    """Box path of RPN proposal selection for one level in ONE launch (extension; reference
    modeling/rpn/inference.py:75-110 + box_coder.py:61-95): decodes the top-k anchors straight from the head output
    [N,4A,H,W] and writes boxes[:, col:col+k] / scores[:, col:col+k] of the image-major result and rows
    [off, off+N*k) of the level-major NMS input (nms_boxes, nms_scores, ok uint8)."""
    _need_cuda("rpn_decode", box_regression, topk_idx, topk_scores, anchors, image_hw, boxes, scores)
    reg = _f32c("rpn_decode", box_regression)
    topk_scores = _f32c("rpn_decode", topk_scores)
    anchors = _f32c("rpn_decode", anchors)
    topk_idx = topk_idx.contiguous()
    N, A4, H, W = reg.shape
    k = topk_idx.size(1)
    if topk_idx.dtype != torch.int64 or A4 % 4 or anchors.size(0) != (A4 // 4) * H * W or image_hw.shape != (N, 2):
        raise ValueError("rpn_decode: inconsistent arguments")
    for t in (boxes, scores, nms_boxes, nms_scores, ok):
        if not t.is_contiguous():
            raise ValueError("rpn_decode: outputs must be contiguous")
    if boxes.dtype != torch.float32 or scores.dtype != torch.float32 or ok.dtype != torch.uint8 \
            or col + k > boxes.size(1) or off + N * k > nms_scores.numel():
        raise ValueError("rpn_decode: outputs do not fit")
    if N * k == 0:
        return
    wx, wy, ww, wh = (float(w) for w in weights)
    with _on_device(reg), _timed(("rpn_decode[N=%d,k=%d]", (N, k)), reg):
        check(lib.detops_rpn_decode_f32(
            ptr(reg), ptr(topk_idx), ptr(topk_scores), ptr(anchors), ptr(image_hw), N, A4 // 4, H, W, k, wx, wy, ww, wh,
            float(bbox_xform_clip), float(min_size), boxes.data_ptr() + 16 * col, 4 * boxes.size(1),
            scores.data_ptr() + 4 * col, scores.size(1), nms_boxes.data_ptr() + 16 * off, nms_scores.data_ptr() + 4 * off,
            ok.data_ptr() + off, stream_of(reg)), "rpn_decode")


# ------------------------------------------------------------------------------------------ ROIAlign
def roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    """reference csrc/ROIAlign.h:11-25 -> [K,C,PH,PW]."""
    if _f64(input):
        return _roi_align_forward_f64(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio)
    if not on_device(input) and not on_device(rois):
        return _roi_align_forward_cpu(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio)
    _need_cuda("roi_align_forward", input, rois)
    input = _f32c("roi_align_forward", input)
    rois = _f32c("roi_align_forward", rois)
    N, C, H, W = input.shape
    K = rois.size(0)
    out = torch.empty((K, C, pooled_height, pooled_width), dtype=input.dtype, device=input.device)
    if out.numel() == 0:
        return out
    with _on_device(input):
        ws, nbytes = _fwd_workspace(input.device, K, pooled_height, pooled_width, sampling_ratio)
        check(lib.detops_roi_align_forward_ws_f32(ptr(input), ptr(rois), ptr(out), N, C, H, W, K,
                                                  pooled_height, pooled_width, float(spatial_scale),
                                                  int(sampling_ratio), ptr(ws), nbytes, stream_of(input)),
              "roi_align_forward")
    return out


def _roi_align_forward_cpu(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
    """CPU tensors: the reference dispatches to ROIAlign_forward_cpu (csrc/ROIAlign.h:19-24); host code of the
    same library (csrc/cpu_branch.hip).  Not a fallback of the device path: CUDA tensors never come here."""
    input = _f32c("roi_align_forward", input)
    rois = _f32c("roi_align_forward", rois)
    N, C, H, W = input.shape
    K = rois.size(0)
    out = torch.empty((K, C, pooled_height, pooled_width), dtype=torch.float32)
    if out.numel() == 0:
        return out
    check(lib.detops_roi_align_forward_cpu_f32(ptr(input), ptr(rois), ptr(out), N, C, H, W, K, pooled_height,
                                               pooled_width, float(spatial_scale), int(sampling_ratio)),
          "roi_align_forward(cpu)")
    return out


def _fwd_workspace(device, K, ph, pw, sr):
    """Scratch for the forward's pre-pass (per-ROI sample records; the ROI visiting order for K >= 384), from
    torch's stream-ordered caching allocator; (None, 0) when this shape is served without a pre-pass."""
    nbytes = int(lib.detops_roi_align_forward_workspace_bytes(int(K), int(ph), int(pw), int(sr)))
    if nbytes <= 0:
        return None, 0
    return torch.empty((nbytes,), dtype=torch.uint8, device=device), nbytes


def roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels,
                       height, width, sampling_ratio):
    """reference csrc/ROIAlign.h:27-45 -> zero-initialised [bs,ch,h,w] with the scattered grads."""
    if _f64(grad):
        return _roi_align_backward_f64(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width,
                                       sampling_ratio)
    _need_cuda("roi_align_backward", grad, rois)
    grad = _f32c("roi_align_backward", grad)
    rois = _f32c("roi_align_backward", rois)
    K = rois.size(0)
    gin = torch.empty((batch_size, channels, height, width), dtype=grad.dtype, device=grad.device)
    with _on_device(grad):
        Hs, Ws = (ctypes.c_int * 1)(height), (ctypes.c_int * 1)(width)
        ws, nbytes = _bwd_workspace(grad.device, Hs, Ws, 1, batch_size, channels, K, pooled_height, pooled_width)
        check(lib.detops_roi_align_backward_ws_f32(ptr(grad), ptr(rois), ptr(gin), batch_size, channels,
                                                   height, width, K, pooled_height, pooled_width,
                                                   float(spatial_scale), int(sampling_ratio), 1, ptr(ws), nbytes,
                                                   stream_of(grad)), "roi_align_backward")
    return gin


def _bwd_workspace(device, Hs, Ws, L, N, C, K, ph, pw):
    """Scratch for the binned ROIAlign backward (per-ROI adjoint rows + per-tile hit lists), from
    torch's stream-ordered caching allocator; (None, 0) when the shape uses the workspace-free kernels."""
    nbytes = int(lib.detops_roi_align_backward_workspace_bytes(Hs, Ws, L, int(N), int(C), int(K), int(ph), int(pw)))
    if nbytes <= 0:
        return None, 0
    return torch.empty((nbytes,), dtype=torch.uint8, device=device), nbytes


def _host_arrays(tensors, scales):
    L = len(tensors)
    ptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in tensors])
    Hs = (ctypes.c_int * L)(*[t.size(2) for t in tensors])
    Ws = (ctypes.c_int * L)(*[t.size(3) for t in tensors])
    sc = (ctypes.c_float * L)(*[float(s) for s in scales])
    return ptrs, Hs, Ws, sc


def _nhwc_host_arrays(tensors, scales):
    """(pointer array, H array, W array, scale array) of channels-last [N, C, H, W] tensors (stored [N, H, W, C])"""
    return _host_arrays(tensors, scales)


def roi_align_fpn_forward(inputs, rois, scales, pooled_height, pooled_width, sampling_ratio, k_min,
                          k_max, canonical_scale=224.0, canonical_level=4.0, eps=1e-6, out_channels_last=False):
    """Multi-level ROIAlign in one launch (extension): the device-side form of
    modeling/poolers.py:91-121.  Returns (out [K,C,PH,PW], levels int32 [K]).
    A channels-last pyramid (every level stored NHWC) is read in place by the NHWC kernels (csrc/roi_align_nhwc.hip: same
    values, bit for bit); `out_channels_last` then returns the pooled tensor channels-last as well (for a convolutional
    head) instead of contiguous (for the box head's FC layer)."""
    _need_cuda("roi_align_fpn_forward", rois, *inputs)
    if all(is_channels_last(t) for t in inputs) and inputs[0].dtype == torch.float32:
        return _roi_align_fpn_forward_nhwc(inputs, rois, scales, pooled_height, pooled_width, sampling_ratio, k_min, k_max,
                                           canonical_scale, canonical_level, eps, out_channels_last)
    inputs = [_f32c("roi_align_fpn_forward", t) for t in inputs]
    rois = _f32c("roi_align_fpn_forward", rois)
    N, C = inputs[0].shape[:2]
    K = rois.size(0)
    out = torch.empty((K, C, pooled_height, pooled_width), dtype=torch.float32, device=rois.device)
    levels = torch.empty((K,), dtype=torch.int32, device=rois.device)
    if K == 0:
        return out, levels
    ptrs, Hs, Ws, sc = _host_arrays(inputs, scales)
    with _on_device(rois), _timed(("roi_align_fpn_fwd[K=%d,C=%d,%dx%d]", (K, C, pooled_height, pooled_width)), rois):
        ws, nbytes = _fwd_workspace(rois.device, K, pooled_height, pooled_width, sampling_ratio)
        check(lib.detops_roi_align_fpn_forward_ws_f32(
            ptrs, Hs, Ws, sc, len(inputs), ptr(rois), ptr(out), ptr(levels), N, C, K, pooled_height,
            pooled_width, int(sampling_ratio), int(k_min), int(k_max), float(canonical_scale),
            float(canonical_level), float(eps), ptr(ws), nbytes, stream_of(rois)), "roi_align_fpn_forward")
    return out, levels


def _roi_align_fpn_forward_nhwc(inputs, rois, scales, pooled_height, pooled_width, sampling_ratio, k_min, k_max,
                                canonical_scale, canonical_level, eps, out_channels_last):
    rois = _f32c("roi_align_fpn_forward", rois)
    N, C = inputs[0].shape[:2]
    K = rois.size(0)
    fmt = torch.channels_last if out_channels_last else torch.contiguous_format
    out = torch.empty((K, C, pooled_height, pooled_width), dtype=torch.float32, device=rois.device, memory_format=fmt)
    levels = torch.empty((K,), dtype=torch.int32, device=rois.device)
    if K == 0:
        return out, levels
    ptrs, Hs, Ws, sc = _nhwc_host_arrays(inputs, scales)
    with _on_device(rois), _timed(("roi_align_fpn_fwd[K=%d,C=%d,%dx%d]", (K, C, pooled_height, pooled_width)), rois):
        nbytes = int(lib.detops_roi_align_fpn_forward_nhwc_workspace_bytes(int(K)))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=rois.device) if nbytes > 0 else None
        check(lib.detops_roi_align_fpn_forward_nhwc_f32(
            ptrs, Hs, Ws, sc, len(inputs), ptr(rois), ptr(out), int(bool(out_channels_last)), ptr(levels), N, C, K,
            pooled_height, pooled_width, int(sampling_ratio), int(k_min), int(k_max), float(canonical_scale),
            float(canonical_level), float(eps), ptr(ws), nbytes, stream_of(rois)), "roi_align_fpn_forward_nhwc")
    return out, levels


_SIDE_STREAMS = {}


def _side_stream(device):
    """one auxiliary stream per device for work that only has to be finished by the backward pass"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _SIDE_STREAMS.get(idx)
    if s is None:
        s = _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return s


def roi_align_fpn_backward_prepare(rois, levels, shapes, scales, pooled_height, pooled_width, sampling_ratio):
    """The backward's pre-pass (per-ROI adjoint rows + per-tile hit lists: a function of the ROIs and the map shapes only),
    issued at FORWARD time on a side stream so that it is off the backward pass's critical path.  Returns an opaque
    handle for roi_align_fpn_backward(prepared=...), or None when the shape is served by the one-call kernels."""
    K = rois.size(0)
    if K == 0 or not on_device(rois):
        return None
    N, C = shapes[0][:2]
    L = len(shapes)
    Hs = (ctypes.c_int * L)(*[s[2] for s in shapes])
    Ws = (ctypes.c_int * L)(*[s[3] for s in shapes])
    sc = (ctypes.c_float * L)(*[float(s) for s in scales])
    nbytes = int(lib.detops_roi_align_backward_workspace_bytes(Hs, Ws, L, int(N), int(C), int(K), int(pooled_height), int(pooled_width)))
    if nbytes <= 0:
        return None
    main = torch.cuda.current_stream(rois.device)
    side = _side_stream(rois.device)
    side.wait_stream(main)                       # the ROIs and their levels are produced on the main stream
    with torch.cuda.stream(side):
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=rois.device)
        with _timed(("roi_align_fpn_bwd_prepare[K=%d,C=%d,%dx%d]", (K, C, pooled_height, pooled_width)), rois):
            rc = lib.detops_roi_align_fpn_backward_prepare_f32(
                ptr(rois), ptr(levels), Hs, Ws, sc, L, int(N), int(C), int(K), int(pooled_height), int(pooled_width),
                int(sampling_ratio), ptr(ws), nbytes, side.cuda_stream)
        if rc == -3:                             # DETOPS_EUNSUPPORTED: the other kernels serve this shape
            return None
        check(rc, "roi_align_fpn_backward_prepare")
        done = torch.cuda.Event()
        done.record(side)
    ws.record_stream(main)                       # used by the backward pass on the main stream
    rois.record_stream(side)
    levels.record_stream(side)
    return ws, nbytes, done


def roi_align_fpn_backward(grad, rois, levels, shapes, scales, pooled_height, pooled_width,
                           sampling_ratio, prepared=None, channels_last=False):
    """Backward of roi_align_fpn_forward: returns one zero-initialised gradient map per level.  `prepared`: the handle of
    roi_align_fpn_backward_prepare (the pre-pass already ran): only the main kernel is launched.  `channels_last`: the
    forward read a channels-last pyramid — the gradient maps are produced channels-last by the NHWC kernel
    (csrc/roi_align_nhwc.hip), from a contiguous or a channels-last pooled gradient alike."""
    _need_cuda("roi_align_fpn_backward", grad, rois, levels)
    if channels_last:
        return _roi_align_fpn_backward_nhwc(grad, rois, levels, shapes, scales, pooled_height, pooled_width, sampling_ratio)
    grad = _f32c("roi_align_fpn_backward", grad)
    K = rois.size(0)
    gins = [torch.empty(tuple(s), dtype=torch.float32, device=grad.device) for s in shapes]
    N, C = shapes[0][:2]
    ptrs, Hs, Ws, sc = _host_arrays(gins, scales)
    if prepared is not None:
        ws, nbytes, done = prepared
        with _on_device(grad):
            torch.cuda.current_stream(grad.device).wait_event(done)
            with _timed(("roi_align_fpn_bwd[K=%d,C=%d,%dx%d]", (K, C, pooled_height, pooled_width)), grad):
                check(lib.detops_roi_align_fpn_backward_prepared_f32(
                    ptr(grad), ptrs, Hs, Ws, sc, len(gins), N, C, K, pooled_height, pooled_width, 1, ptr(ws), nbytes,
                    stream_of(grad)), "roi_align_fpn_backward(prepared)")
        return gins
    with _on_device(grad):
        ws, nbytes = _bwd_workspace(grad.device, Hs, Ws, len(gins), N, C, K, pooled_height, pooled_width)
        with _timed(("roi_align_fpn_bwd[K=%d,C=%d,%dx%d]", (K, C, pooled_height, pooled_width)), grad):
            check(lib.detops_roi_align_fpn_backward_ws_f32(
                ptr(grad), ptr(rois), ptr(levels), ptrs, Hs, Ws, sc, len(gins), N, C, K, pooled_height,
                pooled_width, int(sampling_ratio), 1, ptr(ws), nbytes, stream_of(grad)), "roi_align_fpn_backward")
    return gins


NHWC_BACKWARD = os.environ.get("DETOPS_ROIALIGN_NHWC_BWD", "auto")   # auto | ring | native  (A/B switch)


def _roi_align_fpn_backward_nhwc(grad, rois, levels, shapes, scales, pooled_height, pooled_width, sampling_ratio):
    """Channels-last gradient maps.  Two kernels serve this: the pixel-owner ring kernel of the NCHW path with a channels-last
    store epilogue (csrc/roi_align_bwd.hip: 7 x 7 / 14 x 14 bins on launches that fill the chip — the model's shapes; the
    pooled gradient is read as [K, C, PH, PW]), and the NHWC-native hit-parallel kernel (csrc/roi_align_nhwc.hip: any shape,
    contiguous or channels-last pooled gradient).  "auto": the ring kernel where its plan applies (measured 3x faster at the
    model's shapes, profiles/r06*), the native kernel otherwise."""
    if grad.dtype != torch.float32:
        raise RuntimeError("roi_align_fpn_backward: expected a float32 gradient, got %s" % grad.dtype)
    rois = _f32c("roi_align_fpn_backward", rois)
    K = rois.size(0)
    gins = [torch.empty(tuple(s), dtype=torch.float32, device=grad.device, memory_format=torch.channels_last) for s in shapes]
    N, C = shapes[0][:2]
    ptrs, Hs, Ws, sc = _nhwc_host_arrays(gins, scales)
    with _on_device(grad):
        if NHWC_BACKWARD != "native" and K > 0:
            g = grad.contiguous()          # [K, C, PH, PW] (a channels-last pooled gradient is converted: 51 MB for the mask head)
            ws, nbytes = _bwd_workspace(grad.device, Hs, Ws, len(gins), N, C, K, pooled_height, pooled_width)
            if ws is not None:
                with _timed(("roi_align_fpn_bwd[K=%d,C=%d,%dx%d]", (K, C, pooled_height, pooled_width)), g):
                    rc = lib.detops_roi_align_fpn_backward_ring_nhwc_f32(
                        ptr(g), ptr(rois), ptr(levels), ptrs, Hs, Ws, sc, len(gins), N, C, K, pooled_height, pooled_width,
                        int(sampling_ratio), 1, ptr(ws), nbytes, stream_of(g))
                if rc == 0:
                    return gins
                if rc != -3:      # DETOPS_EUNSUPPORTED: the ring plan does not serve this shape
                    check(rc, "roi_align_fpn_backward_ring_nhwc")
            if NHWC_BACKWARD == "ring":
                raise RuntimeError("roi_align_fpn_backward: the ring plan does not serve this shape (DETOPS_ROIALIGN_NHWC_BWD=ring)")
        g_nhwc = is_channels_last(grad)
        if not g_nhwc:
            grad = grad.contiguous()
        nbytes = int(lib.detops_roi_align_fpn_backward_nhwc_workspace_bytes(Hs, Ws, len(gins), N, C, K, pooled_height, pooled_width))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=grad.device) if nbytes > 0 else None
        with _timed(("roi_align_fpn_bwd[K=%d,C=%d,%dx%d]", (K, C, pooled_height, pooled_width)), grad):
            check(lib.detops_roi_align_fpn_backward_nhwc_f32(
                ptr(grad), int(g_nhwc), ptr(rois), ptr(levels), ptrs, Hs, Ws, sc, len(gins), N, C, K, pooled_height,
                pooled_width, int(sampling_ratio), 1, ptr(ws), nbytes, stream_of(grad)), "roi_align_fpn_backward_nhwc")
    return gins


# ------------------------------------------------------------------------------------------ ROIPool
def roi_pool_forward(input, rois, spatial_scale, pooled_height, pooled_width):
    """reference csrc/ROIPool.h:11-24 -> (output, argmax int32)."""
    if _f64(input):
        return _roi_pool_forward_f64(input, rois, spatial_scale, pooled_height, pooled_width)
    _need_cuda("roi_pool_forward", input, rois)
    input = _f32c("roi_pool_forward", input)
    rois = _f32c("roi_pool_forward", rois)
    N, C, H, W = input.shape
    K = rois.size(0)
    out = torch.empty((K, C, pooled_height, pooled_width), dtype=input.dtype, device=input.device)
    argmax = torch.zeros((K, C, pooled_height, pooled_width), dtype=torch.int32, device=input.device)
    if out.numel() == 0:
        return out, argmax
    with _on_device(input):
        check(lib.detops_roi_pool_forward_f32(ptr(input), ptr(rois), ptr(out), ptr(argmax), N, C, H,
                                              W, K, pooled_height, pooled_width,
                                              float(spatial_scale), stream_of(input)),
              "roi_pool_forward")
    return out, argmax


def roi_pool_backward(grad, input, rois, argmax, spatial_scale, pooled_height, pooled_width,
                      batch_size, channels, height, width):
    """reference csrc/ROIPool.h:26-45 (`input` is unused there too)."""
    if _f64(grad):
        return _roi_pool_backward_f64(grad, rois, argmax, pooled_height, pooled_width, batch_size, channels, height, width)
    _need_cuda("roi_pool_backward", grad, rois, argmax)
    grad = _f32c("roi_pool_backward", grad)
    rois = _f32c("roi_pool_backward", rois)
    argmax = argmax.contiguous()
    K = rois.size(0)
    gin = torch.empty((batch_size, channels, height, width), dtype=grad.dtype, device=grad.device)
    with _on_device(grad):
        check(lib.detops_roi_pool_backward_f32(ptr(grad), ptr(rois), ptr(argmax), ptr(gin),
                                               batch_size, channels, height, width, K, pooled_height,
                                               pooled_width, 1, stream_of(grad)), "roi_pool_backward")
    return gin


# ------------------------------------------------------------------------------------------ focal
def _focal_args(name, logits, targets, num_classes):
    _need_cuda(name, logits, targets)
    if logits.dim() != 2:
        raise RuntimeError("logits should be NxClass")  # SigmoidFocalLoss_cuda.cu:112
    logits = _f32c(name, logits)
    if targets.dtype != torch.int32:
        raise RuntimeError("%s: targets must be int32 (reference kernel reads `const int*`)" % name)
    targets = targets.contiguous()
    if targets.numel() != logits.size(0) or logits.size(1) != num_classes:
        raise RuntimeError("%s: shape mismatch" % name)
    return logits, targets


def sigmoid_focalloss_forward(logits, targets, num_classes, gamma, alpha):
    """reference csrc/SigmoidFocalLoss.h:10-24 -> losses [R,C]."""
    if _f64(logits):
        return _focal_f64(logits, targets, None, num_classes, gamma, alpha)
    logits, targets = _focal_args("sigmoid_focalloss_forward", logits, targets, num_classes)
    losses = torch.empty_like(logits)
    with _on_device(logits):
        check(lib.detops_sigmoid_focal_loss_forward_f32(ptr(logits), ptr(targets), ptr(losses),
                                                        logits.size(0), num_classes, float(gamma),
                                                        float(alpha), stream_of(logits)),
              "sigmoid_focalloss_forward")
    return losses


def sigmoid_focalloss_backward(logits, targets, d_losses, num_classes, gamma, alpha):
    """reference csrc/SigmoidFocalLoss.h:26-41 -> d_logits [R,C]."""
    if _f64(logits):
        return _focal_f64(logits, targets, d_losses, num_classes, gamma, alpha)
    logits, targets = _focal_args("sigmoid_focalloss_backward", logits, targets, num_classes)
    _need_cuda("sigmoid_focalloss_backward", d_losses)
    d_losses = _f32c("sigmoid_focalloss_backward", d_losses)
    if d_losses.shape != logits.shape:
        raise RuntimeError("sigmoid_focalloss_backward: d_losses must be [R,C]")
    d_logits = torch.empty_like(logits)
    with _on_device(logits):
        check(lib.detops_sigmoid_focal_loss_backward_f32(ptr(logits), ptr(targets), ptr(d_losses),
                                                         ptr(d_logits), logits.size(0), num_classes,
                                                         float(gamma), float(alpha),
                                                         stream_of(logits)),
              "sigmoid_focalloss_backward")
    return d_logits


def sigmoid_focalloss_forward_sum(logits, targets, num_classes, gamma, alpha):
    """Extension: sum(losses) without materialising [R,C] (what SigmoidFocalLoss.forward needs).  Two launches
    (per-workgroup sums, then one fixed-order reduction) and nothing else: no zero-fill, no torch reduction."""
    logits, targets = _focal_args("sigmoid_focalloss_forward_sum", logits, targets, num_classes)
    nbytes = int(lib.detops_sigmoid_focal_loss_sum_workspace_bytes())
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=logits.device)
    total = torch.empty((), dtype=torch.float32, device=logits.device)
    with _on_device(logits), _timed(("focal_fwd_sum[R=%d,C=%d]", (logits.size(0), num_classes)), logits):
        check(lib.detops_sigmoid_focal_loss_forward_sum_ws_f32(
            ptr(logits), ptr(targets), None, ptr(total), logits.size(0), num_classes, float(gamma), float(alpha),
            ptr(ws), nbytes, stream_of(logits)), "sigmoid_focalloss_forward_sum")
    return total


def sigmoid_focalloss_backward_scalar(logits, targets, d_loss, num_classes, gamma, alpha):
    """Extension: backward of the summed loss; d_loss is a 0-dim / 1-element device tensor."""
    logits, targets = _focal_args("sigmoid_focalloss_backward_scalar", logits, targets, num_classes)
    _need_cuda("sigmoid_focalloss_backward_scalar", d_loss)
    d_loss = d_loss.reshape(1).to(torch.float32).contiguous()
    d_logits = torch.empty_like(logits)
    with _on_device(logits), _timed(("focal_bwd_scalar[R=%d,C=%d]", (logits.size(0), num_classes)), logits):
        check(lib.detops_sigmoid_focal_loss_backward_scalar_f32(
            ptr(logits), ptr(targets), ptr(d_loss), ptr(d_logits), logits.size(0), num_classes,
            float(gamma), float(alpha), stream_of(logits)), "sigmoid_focalloss_backward_scalar")
    return d_logits


# ------------------------------------------------------------------------------------------ frozen BN
def is_channels_last(t):
    """4-d tensor stored NHWC (and not at the same time plain-contiguous, e.g. C == 1 or H == W == 1)"""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def frozen_bn_act_forward(x, scale, bias, residual, relu):
    """Extension: y = [relu](x * scale[c] + bias[c] [+ residual]) in one pass.  NCHW-contiguous input -> NCHW output; a
    channels-last (NHWC) input is processed in place of its layout and returns a channels-last output (no transpose)."""
    _need_cuda("frozen_bn_act_forward", x, scale, bias, residual)
    code = _lib.DTYPE_CODE[x.dtype]
    nhwc = is_channels_last(x)
    if not nhwc:
        x = x.contiguous()
    if residual is not None:
        if residual.dtype != x.dtype or residual.shape != x.shape:
            raise RuntimeError("frozen_bn_act_forward: residual must match x")
        residual = residual.contiguous(memory_format=torch.channels_last) if nhwc else residual.contiguous()
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(N * C, 1)
    y = torch.empty_like(x)      # preserves the memory format
    with _on_device(x), _timed(("frozen_bn_fwd[n=%d,nc=%d,e=%d,res=%d]", (x.numel(), N * C, _ESIZE[x.dtype], residual is not None)), x, every=8):
        if nhwc:
            check(lib.detops_frozen_bn_act_forward_nhwc(ptr(x), ptr(scale), ptr(bias), ptr(residual), ptr(y), code, N * HW, C,
                                                        int(bool(relu)), stream_of(x)), "frozen_bn_act_forward_nhwc")
        else:
            check(lib.detops_frozen_bn_act_forward(ptr(x), ptr(scale), ptr(bias), ptr(residual), ptr(y), code, N, C, HW,
                                                   int(bool(relu)), stream_of(x)), "frozen_bn_act_forward")
    return y


def frozen_bn_act_backward(grad_y, y, scale, relu, need_residual):
    """Extension: (grad_x, grad_residual or None) of frozen_bn_act_forward.  The layout follows the saved output `y` when
    there is one (ReLU), otherwise the incoming gradient's."""
    _need_cuda("frozen_bn_act_backward", grad_y, scale)
    code = _lib.DTYPE_CODE[grad_y.dtype]
    nhwc = is_channels_last(y) if (relu and y is not None) else is_channels_last(grad_y)
    grad_y = grad_y.contiguous(memory_format=torch.channels_last) if nhwc else grad_y.contiguous()
    N, C = grad_y.shape[0], grad_y.shape[1]
    HW = grad_y.numel() // max(N * C, 1)
    gx = torch.empty_like(grad_y)
    gres = torch.empty_like(grad_y) if need_residual else None
    with _on_device(grad_y), _timed(("frozen_bn_bwd[n=%d,nc=%d,e=%d,res=%d,relu=%d]", (grad_y.numel(), N * C, _ESIZE[grad_y.dtype], bool(need_residual), bool(relu))),
                                     grad_y, every=8):
        if nhwc:
            check(lib.detops_frozen_bn_act_backward_nhwc(ptr(grad_y), ptr(y) if relu else None, ptr(scale), ptr(gx), ptr(gres),
                                                         code, N * HW, C, int(bool(relu)), stream_of(grad_y)),
                  "frozen_bn_act_backward_nhwc")
        else:
            if relu and y is not None and not y.is_contiguous():
                y = y.contiguous()
            check(lib.detops_frozen_bn_act_backward(ptr(grad_y), ptr(y) if relu else None, ptr(scale), ptr(gx), ptr(gres),
                                                    code, N, C, HW, int(bool(relu)), stream_of(grad_y)),
                  "frozen_bn_act_backward")
    return gx, gres


_ONES = {}


def _ones(C, device):
    key = (int(C), str(device))
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones((C,), dtype=torch.float32, device=device)
    return t


def bias_act_supported(x, bias):
    """bias_act serves fp32 channels-last activations on the device (any channel count: widths dividing 1024 through the
    fused kernels of csrc/bias_act.hip, the others — the RPN's 3 / 12 and the mask logits' 81 channels — through an in-place
    add and a matrix-vector product for the bias gradient)"""
    return (bias is not None and on_device(x) and x.dtype in _lib.DTYPE_CODE and bias.dtype == torch.float32 and x.dim() == 4
            and is_channels_last(x))


class _BiasAct(torch.autograd.Function):
    """y = [relu](x + bias[c]) on a channels-last activation (csrc/bias_act.hip).  The point is the BACKWARD: the bias
    gradient of a channels-last gradient is a column sum over [N*H*W, C]; PyTorch's `sum((0, 2, 3))` runs it through a generic
    strided reduce kernel at ~0.1 TB/s."""

    @staticmethod
    def forward(ctx, x, bias, relu):
        ctx.fused = bool(lib.detops_bias_act_supported(int(x.shape[1])))
        if ctx.fused:
            y = frozen_bn_act_forward(x, _ones(x.shape[1], x.device), bias.contiguous(), None, relu)
        else:
            bb = bias.to(x.dtype).view(1, -1, 1, 1)
            y = x.add_(bb) if not x.requires_grad else x + bb
            if relu:
                y = y.relu_()
        ctx.relu = relu
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        N, C, H, W = gy.shape
        rows = N * H * W
        if not ctx.fused:
            gx = gy * (y > 0) if ctx.relu else gy
            return gx, column_sum(gx.permute(0, 2, 3, 1).reshape(rows, C)), None    # (a view of the channels-last storage)
        gb = torch.empty((C,), dtype=torch.float32, device=gy.device)
        gx = torch.empty_like(gy) if ctx.relu else gy
        nbytes = int(lib.detops_bias_act_backward_workspace_bytes(rows, C))
        ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=gy.device)
        with _on_device(gy), _timed(("bias_act_bwd[n=%d,C=%d,e=%d,relu=%d]", (gy.numel(), C, _ESIZE[gy.dtype], bool(ctx.relu))), gy, every=4):
            check(lib.detops_bias_act_backward_nhwc(ptr(gy), ptr(y) if ctx.relu else None, ptr(gx), ptr(gb), _lib.DTYPE_CODE[gy.dtype],
                                                    rows, C, int(bool(ctx.relu)), ptr(ws), nbytes, stream_of(gy)), "bias_act_backward")
        return gx, gb, None


def column_sum(x2d):
    """[rows, C] row-major fp32 / fp16 / bf16 -> fp32 [C] column sums (extension, csrc/bias_act.hip; C <= 256), deterministic"""
    _need_cuda("column_sum", x2d)
    if x2d.dtype not in _lib.DTYPE_CODE:
        raise RuntimeError("column_sum: unsupported dtype %s" % x2d.dtype)
    x2d = x2d.contiguous()
    rows, C = x2d.shape
    out = torch.empty((C,), dtype=torch.float32, device=x2d.device)
    if C > 256:
        return x2d.float().sum(0)
    nbytes = int(lib.detops_column_sum_workspace_bytes(rows, C))
    ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=x2d.device)
    with _on_device(x2d):
        check(lib.detops_column_sum(ptr(x2d), ptr(out), _lib.DTYPE_CODE[x2d.dtype], rows, C, ptr(ws), nbytes, stream_of(x2d)), "column_sum")
    return out


def bias_act(x, bias, relu=False):
    """[relu](x + bias[c]) for a channels-last fp32 activation (extension; see bias_act_supported).  `x` may be overwritten
    (it is the caller's fresh convolution output)."""
    _need_cuda("bias_act", x, bias)
    return _BiasAct.apply(x, bias, bool(relu))


# ------------------------------------------------------------------------------------------ deformable conv
def _dcn_check(name, *tensors):
    _need_cuda(name, *tensors)
    dt = tensors[0].dtype
    if dt not in _lib.DTYPE_CODE:
        raise RuntimeError("%s: unsupported dtype %s" % (name, dt))
    for t in tensors:
        if t is not None and t.dtype != dt:
            raise RuntimeError("%s: all tensors must share one dtype" % name)
    return _lib.DTYPE_CODE[dt]


def _out_hw(H, W, kH, kW, padH, padW, dH, dW, dilH, dilW):
    return ((H + 2 * padH - (dilH * (kH - 1) + 1)) // dH + 1,
            (W + 2 * padW - (dilW * (kW - 1) + 1)) // dW + 1)


def _shape_check(input, offset, grad_output, weight, kH, kW, dH, dW, padH, padW, dilH, dilW, group,
                 deformable_group):
    """reference csrc/cuda/deform_conv_cuda.cu:67-156 (same conditions, RuntimeError)."""
    if weight.dim() != 4:
        raise RuntimeError("4D weight tensor (nOutputPlane,nInputPlane,kH,kW) expected, but got: %d" % weight.dim())
    if not weight.is_contiguous() and not is_channels_last(weight):
        raise RuntimeError("weight tensor has to be contiguous")
    if kW <= 0 or kH <= 0:
        raise RuntimeError("kernel size should be greater than zero, but got kH: %d kW: %d" % (kH, kW))
    if weight.size(2) != kH or weight.size(3) != kW:
        raise RuntimeError("kernel size should be consistent with weight")
    if dW <= 0 or dH <= 0:
        raise RuntimeError("stride should be greater than zero, but got dH: %d dW: %d" % (dH, dW))
    if dilW <= 0 or dilH <= 0:
        raise RuntimeError("dilation should be greater than 0")
    if input.dim() != 4:
        raise RuntimeError("3D or 4D input tensor expected but got: %d" % input.dim())
    nIn = weight.size(1) * group
    H, W = input.size(2), input.size(3)
    Ho, Wo = _out_hw(H, W, kH, kW, padH, padW, dH, dW, dilH, dilW)
    if nIn % deformable_group != 0:
        raise RuntimeError("input channels must divide deformable group size")
    if Wo < 1 or Ho < 1:
        raise RuntimeError("Given input size: (%d x %d x %d). Calculated output size: (%d x %d x %d). "
                           "Output size is too small" % (nIn, H, W, weight.size(0), Ho, Wo))
    if input.size(1) != nIn:
        raise RuntimeError("invalid number of input planes, expected: %d, but got: %d" % (nIn, input.size(1)))
    if H < kH or W < kW:
        raise RuntimeError("input image is smaller than kernel")
    if offset.size(2) != Ho or offset.size(3) != Wo:
        raise RuntimeError("invalid spatial size of offset, expected height: %d width: %d, but got "
                           "height: %d width: %d" % (Ho, Wo, offset.size(2), offset.size(3)))
    if offset.size(1) != deformable_group * 2 * kH * kW:
        raise RuntimeError("invalid number of channels of offset")
    if offset.size(0) != input.size(0):
        raise RuntimeError("invalid batch size of offset")
    if grad_output is not None:
        if grad_output.size(1) != weight.size(0):
            raise RuntimeError("invalid number of gradOutput planes, expected: %d, but got: %d"
                               % (weight.size(0), grad_output.size(1)))
        if grad_output.size(2) != Ho or grad_output.size(3) != Wo:
            raise RuntimeError("invalid size of gradOutput")
    return Ho, Wo


def _geom_args(B, C, H, W, kH, kW, padH, padW, dH, dW, dilH, dilW, dg):
    return (B, C, H, W, kH, kW, padH, padW, dH, dW, dilH, dilW, dg)


def deformable_im2col(im, offset, mask, kH, kW, padH, padW, dH, dW, dilH, dilW, dg):
    """col [C*kH*kW, B*Ho*Wo] (deform_conv_kernel_cuda.cu:197-250 / :577-640)."""
    code = _dcn_check("deformable_im2col", im, offset, mask)
    B, C, H, W = im.shape
    Ho, Wo = _out_hw(H, W, kH, kW, padH, padW, dH, dW, dilH, dilW)
    col = torch.empty((C * kH * kW, B * Ho * Wo), dtype=im.dtype, device=im.device)
    with _on_device(im), _timed(("dcn_im2col[B=%d,C=%d,%dx%d,k=%d,e=%d,m=%d]", (B, C, H, W, kH, _ESIZE[im.dtype], mask is not None)), im, every=8):
        check(lib.detops_deformable_im2col(ptr(im), ptr(offset), ptr(mask), ptr(col), code,
                                           *_geom_args(B, C, H, W, kH, kW, padH, padW, dH, dW, dilH,
                                                       dilW, dg), stream_of(im)), "deformable_im2col")
    return col


def deformable_col2im(col, offset, mask, grad_im, kH, kW, padH, padW, dH, dW, dilH, dilW, dg):
    """accumulates into grad_im [B,C,H,W] (deform_conv_kernel_cuda.cu:286-342 / :642-700)."""
    code = _dcn_check("deformable_col2im", col, offset, mask, grad_im)
    B, C, H, W = grad_im.shape
    geom = _geom_args(B, C, H, W, kH, kW, padH, padW, dH, dW, dilH, dilW, dg)
    with _on_device(col):
        # gather lists for the atomic-free path live in a scratch tensor (torch's caching allocator:
        # stream-ordered reuse, no hipMalloc per call); 0 bytes = shape outside the index plan
        nbytes = int(lib.detops_deformable_col2im_workspace_bytes(*geom))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=col.device) if nbytes > 0 else None
        with _timed(("dcn_col2im[B=%d,C=%d,%dx%d,k=%d,e=%d,m=%d]", (B, C, H, W, kH, _ESIZE[col.dtype], mask is not None)), col, every=8):
            check(lib.detops_deformable_col2im_ws(ptr(col), ptr(offset), ptr(mask), ptr(grad_im), code, *geom,
                                                  ptr(ws), nbytes, stream_of(col)), "deformable_col2im")


def deformable_col2im_coord(col, im, offset, mask, grad_offset, grad_mask, kH, kW, padH, padW, dH,
                            dW, dilH, dilW, dg):
    """overwrites grad_offset (and grad_mask) (deform_conv_kernel_cuda.cu:380-443 / :702-774)."""
    code = _dcn_check("deformable_col2im_coord", col, im, offset, mask, grad_offset, grad_mask)
    B, C, H, W = im.shape
    with _on_device(col), _timed(("dcn_col2im_coord[B=%d,C=%d,%dx%d,k=%d,e=%d,m=%d]", (B, C, H, W, kH, _ESIZE[col.dtype], mask is not None)), col, every=8):
        check(lib.detops_deformable_col2im_coord(ptr(col), ptr(im), ptr(offset), ptr(mask),
                                                 ptr(grad_offset), ptr(grad_mask), code,
                                                 *_geom_args(B, C, H, W, kH, kW, padH, padW, dH, dW,
                                                             dilH, dilW, dg), stream_of(col)),
              "deformable_col2im_coord")


def _fused_dcn_forward(input, weight, offset, mask, bias, out, kH, kW, padH, padW, dH, dW, dilH, dilW, group, dg):
    """Fused implicit-GEMM forward on the matrix cores (csrc/deform_conv.hip: the deformed operand tile is built
    in LDS, `columns` is never written).  Returns False when the shape is outside the fused plan (fp32 storage,
    grouped convolution, channels per deformable group not a multiple of 32) or tuning dcn_fused = 2."""
    if group != 1 or input.dtype not in (torch.float16, torch.bfloat16) or _lib.tuning_get("dcn_fused") == 2:
        return False
    B, C, H, W = input.shape
    Cout = weight.size(0)
    code = 1 if input.dtype == torch.float16 else 2
    geo = (code, B, C, H, W, Cout, kH, kW, padH, padW, dH, dW, dilH, dilW, dg)
    nbytes = int(lib.detops_deform_conv_forward_fused_workspace_bytes(*geo))
    if nbytes == 0:
        return False
    with _on_device(input):
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=input.device)
        with _timed(("dcn_fused_fwd[B=%d,C=%d,%dx%d,Cout=%d,k=%d,e=2,m=%d]", (B, C, H, W, Cout, kH, mask is not None)), input, every=8):
            check(lib.detops_deform_conv_forward_fused(ptr(input), ptr(weight), ptr(offset), ptr(mask), ptr(bias),
                                                       ptr(out), *geo, ptr(ws), nbytes, stream_of(input)),
                  "deform_conv_forward_fused")
    return True


# ---- channels-last pipeline (csrc/deform_conv.hip, "channels-last pipeline"): every per-sampling-point operand is
# channel-fastest, the GEMMs are plain library GEMMs on those layouts, the input gradient uses the transposed
# sampling operator + a GEMM instead of the col2im scatter.  Served shapes: conv groups == 1, deformable_group == 1,
# channel counts that are power-of-two multiples of a 16-byte vector (all model shapes).
def _nhwc_ok(input, weight, group, dg, geom=None):
    """channels-last pipeline applicable?  With `geom` (kH, kW, padH, padW, dH, dW, dilH, dilW, dg) the backward's index
    plan is checked too (empty batches and shapes beyond its 32-bit limits return 0 bytes): callers then take the
    reference-layout kernels BEFORE anything is launched instead of failing half way."""
    if group != 1 or dg != 1 or input.dtype not in _lib.DTYPE_CODE or _lib.tuning_get("dcn_nhwc") == 2:
        return False
    if input.numel() == 0:
        return False
    if not lib.detops_deformable_nhwc_supported(_lib.DTYPE_CODE[input.dtype], input.size(1), weight.size(0), dg):
        return False
    if geom is not None:
        B, C, H, W = input.shape
        if int(lib.detops_deformable_transposed_sample_workspace_bytes(B, C, H, W, *geom)) == 0:
            return False
    return True


def _dcn_in(x):
    """a channels-last tensor stays as it is (the channels-last pipeline reads it in place); anything else is made contiguous"""
    return x if is_channels_last(x) else x.contiguous()


def _to_nhwc(x):
    """[B, C, H, W] -> [B, H*W, C] (contiguous).  A channels-last tensor IS that layout: a view, no launch."""
    B, C, H, W = x.shape
    if is_channels_last(x):
        return x.permute(0, 2, 3, 1).reshape(B, H * W, C)
    x = x.contiguous()
    out = torch.empty((B, H * W, C), dtype=x.dtype, device=x.device)
    if out.numel():
        with _on_device(x), _timed(("dcn_to_nhwc[n=%d,e=%d]", (x.numel(), _ESIZE[x.dtype])), x, every=8):
            check(lib.detops_nchw_to_nhwc(ptr(x), ptr(out), _lib.DTYPE_CODE[x.dtype], B, C, H * W, stream_of(x)), "nchw_to_nhwc")
    return out


def _im2col_nhwc(xT, offset, mask, B, C, H, W, geom):
    kH, kW = geom[0], geom[1]
    Ho, Wo = _out_hw(H, W, *geom[:8])
    colT = torch.empty((B * Ho * Wo, kH * kW * C), dtype=xT.dtype, device=xT.device)
    if colT.numel():
        with _on_device(xT), _timed(("dcn_im2col_nhwc[B=%d,C=%d,%dx%d,k=%d,e=%d,m=%d]", (B, C, H, W, kH, _ESIZE[xT.dtype], mask is not None)), xT, every=8):
            check(lib.detops_deformable_im2col_nhwc(ptr(xT), ptr(offset), ptr(mask), ptr(colT), _lib.DTYPE_CODE[xT.dtype],
                                                    B, C, H, W, *geom, stream_of(xT)), "deformable_im2col_nhwc")
    return colT


def _coord_nhwc(colsG, xT, offset, mask, grad_offset, grad_mask, B, C, H, W, geom):
    with _on_device(xT), _timed(("dcn_coord_nhwc[B=%d,C=%d,%dx%d,k=%d,e=%d,m=%d]", (B, C, H, W, geom[0], _ESIZE[xT.dtype], mask is not None)), xT, every=8):
        check(lib.detops_deformable_coord_nhwc(ptr(colsG), ptr(xT), ptr(offset), ptr(mask), ptr(grad_offset), ptr(grad_mask),
                                               _lib.DTYPE_CODE[xT.dtype], B, C, H, W, *geom, stream_of(xT)), "deformable_coord_nhwc")


def _transposed_sample(gT, offset, mask, B, C, H, W, Cout, geom):
    """S_T [B*H*W, kh*kw*Cout]: per gradient-map pixel and tap, the weighted sum of the output-gradient vectors of the
    sampling points that touch the pixel."""
    kH, kW = geom[0], geom[1]
    S_T = torch.empty((B * H * W, kH * kW * Cout), dtype=gT.dtype, device=gT.device)
    nbytes = int(lib.detops_deformable_transposed_sample_workspace_bytes(B, C, H, W, *geom))
    if nbytes == 0:
        raise RuntimeError("deformable_transposed_sample: shape outside the index plan")
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=gT.device)
    with _on_device(gT), _timed(("dcn_transposed_sample[B=%d,Cout=%d,%dx%d,k=%d,e=%d,m=%d]", (B, Cout, H, W, kH, _ESIZE[gT.dtype], mask is not None)), gT, every=8):
        check(lib.detops_deformable_transposed_sample(ptr(gT), ptr(offset), ptr(mask), ptr(S_T), _lib.DTYPE_CODE[gT.dtype],
                                                      B, C, H, W, Cout, *geom, ptr(ws), nbytes, stream_of(gT)),
              "deformable_transposed_sample")
    return S_T


DCN_INPUT_GRAD = os.environ.get("DETOPS_DCN_INPUT_GRAD", "col2im")   # col2im | transposed  (A/B switch)


def _col2im_nhwc(colsG, offset, mask, B, C, H, W, geom):
    """grad_in_T [B, H*W, C] from the channel-fastest column gradient [B*Ho*Wo, kh*kw*C]: col2im as a gather over the inverted
    index (csrc/deform_conv.hip col2im_nhwc_gather_kernel); None when the shape is outside the kernel's plan."""
    kH = geom[0]
    nbytes = int(lib.detops_deformable_transposed_sample_workspace_bytes(B, C, H, W, *geom))
    if nbytes == 0 or C // (8 if _ESIZE[colsG.dtype] == 2 else 4) > 256:
        return None
    ginT = torch.empty((B, H * W, C), dtype=colsG.dtype, device=colsG.device)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=colsG.device)
    with _on_device(colsG), _timed(("dcn_col2im_nhwc[B=%d,C=%d,%dx%d,k=%d,e=%d,m=%d]", (B, C, H, W, kH, _ESIZE[colsG.dtype], mask is not None)), colsG, every=8):
        rc = lib.detops_deformable_col2im_nhwc(ptr(colsG), ptr(offset), ptr(mask), ptr(ginT), _lib.DTYPE_CODE[colsG.dtype],
                                               B, C, H, W, *geom, ptr(ws), nbytes, stream_of(colsG))
    if rc == -3:
        return None
    check(rc, "deformable_col2im_nhwc")
    return ginT


def _w_tap_major(weight):
    """[Cout, C, kh, kw] -> W2 [Cout, kh*kw*C] (tap-major, channel-fastest: the column order of colT)"""
    Cout = weight.size(0)
    return weight.permute(0, 2, 3, 1).reshape(Cout, -1)


def _nhwc_forward(input, weight, offset, mask, bias, out, geom):
    """-> (xT, colT): the channel-fastest copy of the input and the column matrix, which the backward pass of the same
    layer needs again (deform_conv_backward_all(saved=...))"""
    B, C, H, W = input.shape
    Cout = weight.size(0)
    xT = _to_nhwc(input)
    colT = _im2col_nhwc(xT, offset, mask, B, C, H, W, geom)
    if is_channels_last(out):
        # channels-last output: ONE GEMM over the whole batch straight into the output's storage ([B*Ho*Wo, Cout])
        torch.mm(colT, _w_tap_major(weight).t(), out=out.permute(0, 2, 3, 1).reshape(-1, Cout))
    else:
        torch.bmm(_w_tap_major(weight).unsqueeze(0).expand(B, -1, -1), colT.view(B, -1, colT.size(1)).transpose(1, 2),
                  out=out.view(B, Cout, -1))
    if bias is not None:
        out += bias.to(out.dtype).view(1, -1, 1, 1)
    return xT, colT


def _nhwc_backward(input, weight, offset, mask, grad_output, grad_input, grad_offset, grad_mask, grad_weight, grad_bias,
                   geom, scale=1.0):
    """any of grad_input+grad_offset(+grad_mask) / grad_weight(+grad_bias) may be None (the v1 entry points ask for
    them in two separate calls); accumulate / overwrite semantics of the reference functions"""
    B, C, H, W = input.shape
    Cout = weight.size(0) if weight is not None else grad_weight.size(0)
    xT = _to_nhwc(input)
    gT = _to_nhwc(grad_output)                                   # [B, Ho*Wo, Cout]
    g2 = gT.view(-1, Cout)
    if grad_input is not None:
        W2 = _w_tap_major(weight)                                # [Cout, K*C]
        colsG = torch.mm(g2, W2)                                 # column gradient, channel-fastest: [B*Ho*Wo, K*C]
        _coord_nhwc(colsG, xT, offset, mask, grad_offset, grad_mask, B, C, H, W, geom)
        ginT = _col2im_nhwc(colsG, offset, mask, B, C, H, W, geom) if DCN_INPUT_GRAD == "col2im" else None
        del colsG
        if ginT is not None:       # [B, H*W, C]: accumulated into the caller's (reference semantics) NCHW gradient
            grad_input.view(B, C, -1).add_(ginT.transpose(1, 2))
        else:
            S_T = _transposed_sample(gT, offset, mask, B, C, H, W, Cout, geom)     # [B*H*W, K*Cout]
            W2T = weight.permute(1, 2, 3, 0).reshape(C, -1)          # [C, K*Cout]
            grad_input.view(B, C, -1).baddbmm_(W2T.unsqueeze(0).expand(B, -1, -1), S_T.view(B, H * W, -1).transpose(1, 2))
    if grad_weight is not None:
        colT = _im2col_nhwc(xT, offset, mask, B, C, H, W, geom)  # [B*Ho*Wo, K*C]
        gw2 = torch.mm(g2.t(), colT)                             # [Cout, K*C]
        kH, kW = geom[0], geom[1]
        grad_weight.add_(gw2.view(Cout, kH, kW, C).permute(0, 3, 1, 2), alpha=float(scale))
        if grad_bias is not None:
            grad_bias += g2.sum(0)


def deform_conv_backward_all(input, offset, mask, weight, grad_output, kH, kW, padH, padW, dH, dW, dilH, dilW, group,
                             deformable_group, need_input=True, need_weight=True, need_bias=False, saved=None):
    """Extension (not a name of the reference's `_C`): every gradient of one deformable convolution (v1: mask None;
    v2: modulated) from ONE pass over the channels-last pipeline.  The reference API (deform_conv_backward_input +
    deform_conv_backward_parameters, kept above) asks for them in two calls, each of which must rebuild the
    channel-fastest copies of the input and of the output gradient and accumulates into caller-zeroed tensors:
    per layer that is two extra transposes, three fills and an accumulate — about a third of the launches of a
    host-bound step (R-101 + DCN under fp16).

    `saved` = what the forward pass of the same layer kept (`keep=[]` of deform_conv_forward /
    modulated_deform_conv_forward: the channel-fastest input copy and the column matrix): neither is rebuilt then
    (one transpose and one im2col launch less per layer: 35-55 us of device time at the cfg-5 shapes, for 9x the
    layer input in memory until its backward pass has run).

    -> (grad_input, grad_offset, grad_mask, grad_weight, grad_bias), None where not asked for / not applicable;
    returns None when the shape is outside the channels-last plan (the caller then uses the reference entry points)."""
    _dcn_check("deform_conv_backward_all", input, offset, weight, grad_output)
    geom = (kH, kW, padH, padW, dH, dW, dilH, dilW, deformable_group)
    if not _nhwc_ok(input, weight, group, deformable_group, geom):
        return None
    cl_in = is_channels_last(input)
    cl_w = is_channels_last(weight)          # channels-last parameter: its tap-major matrix is a view, and so is its gradient
    input, offset, weight, grad_output = _dcn_in(input), offset.contiguous(), _dcn_in(weight), _dcn_in(grad_output)
    if mask is not None:
        mask = mask.contiguous()
    B, C, H, W = input.shape
    Cout = weight.size(0)
    xT, colT = saved if saved is not None else (_to_nhwc(input), None)
    gT = _to_nhwc(grad_output)                                   # [B, Ho*Wo, Cout]
    g2 = gT.view(-1, Cout)
    grad_input = grad_offset = grad_mask = grad_weight = grad_bias = None
    if need_input:
        colsG = torch.mm(g2, _w_tap_major(weight))               # column gradient, channel-fastest: [B*Ho*Wo, K*C]
        grad_offset = torch.empty_like(offset)                   # written in full by the coordinate kernel
        grad_mask = torch.empty_like(mask) if mask is not None else None
        _coord_nhwc(colsG, xT, offset, mask, grad_offset, grad_mask, B, C, H, W, geom)
        # input gradient: col2im of the column gradient as a gather (no S_T, no second conv-sized GEMM); the transposed
        # sampling of the output gradient + GEMM of rounds 3-5 stays as the A/B form and for shapes outside the gather's plan
        ginT = _col2im_nhwc(colsG, offset, mask, B, C, H, W, geom) if DCN_INPUT_GRAD == "col2im" else None
        del colsG
        if ginT is not None:
            if cl_in:  # the input was channels-last: so is its gradient ([B, H*W, C] viewed as [B, C, H, W])
                grad_input = ginT.view(B, H, W, C).permute(0, 3, 1, 2)
            else:      # [B, H*W, C] -> [B, C, H*W]: the same tile transpose with the roles of C and H*W swapped
                grad_input = torch.empty((B, C, H, W), dtype=ginT.dtype, device=ginT.device)
                with _on_device(ginT):
                    check(lib.detops_nchw_to_nhwc(ptr(ginT), ptr(grad_input), _lib.DTYPE_CODE[ginT.dtype], B, H * W, C,
                                                  stream_of(ginT)), "nhwc_to_nchw")
        else:
            S_T = _transposed_sample(gT, offset, mask, B, C, H, W, Cout, geom)     # [B*H*W, K*Cout]
            W2T = weight.permute(1, 2, 3, 0).reshape(C, -1)          # [C, K*Cout]
            if cl_in:     # the input was channels-last: so is its gradient ([B*H*W, C] from one GEMM, viewed as [B, C, H, W])
                grad_input = torch.mm(S_T, W2T.t()).view(B, H, W, C).permute(0, 3, 1, 2)
            else:
                grad_input = torch.bmm(W2T.unsqueeze(0).expand(B, -1, -1), S_T.view(B, H * W, -1).transpose(1, 2)).view(B, C, H, W)
    if need_weight:
        if colT is None:
            colT = _im2col_nhwc(xT, offset, mask, B, C, H, W, geom)  # [B*Ho*Wo, K*C]
        grad_weight = torch.mm(g2.t(), colT).view(Cout, kH, kW, C).permute(0, 3, 1, 2)   # [Cout, C, kH, kW], channels-last strides
        if not cl_w:
            grad_weight = grad_weight.contiguous()
    if need_bias:
        grad_bias = g2.sum(0)
    return grad_input, grad_offset, grad_mask, grad_weight, grad_bias


def _grouped_weight_times_cols(weight, col, group, out):
    """out[g] (+)= W[g] @ col[g]; out is [Cout, ncol] (written), fp32/half via rocBLAS/hipBLASLt."""
    Cout = weight.size(0)
    Mg = Cout // group
    Kg = col.size(0) // group
    w2 = weight.reshape(group, Mg, Kg)
    for g in range(group):
        torch.mm(w2[g], col[g * Kg:(g + 1) * Kg], out=out[g * Mg:(g + 1) * Mg])


def deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH,
                        dilationW, dilationH, group, deformable_group, im2col_step, keep=None):
    """reference csrc/deform_conv.h:11-42 / csrc/cuda/deform_conv_cuda.cu:158-266.
    Writes `output` in place; note the W-before-H argument order.  Returns 1.
    `keep` (extension): a list that receives what deform_conv_backward_all(saved=...) can reuse, when the
    channels-last pipeline served the call."""
    _dcn_check("deform_conv_forward", input, weight, offset, output)
    # (a channels-last weight — a model switched to channels-last carries its 4-d parameters that way — IS the tap-major matrix
    #  the channels-last pipeline multiplies with: _w_tap_major is a view of it, no copy launch)
    input, offset, weight = _dcn_in(input), offset.contiguous(), _dcn_in(weight)
    Ho, Wo = _shape_check(input, offset, None, weight, kH, kW, dH, dW, padH, padW, dilationH,
                          dilationW, group, deformable_group)
    B, C = input.shape[:2]
    Cout = weight.size(0)
    if B % im2col_step != 0:
        raise RuntimeError("im2col step must divide batchsize")
    out = output.view(B, Cout, Ho, Wo)
    cl = is_channels_last(input)
    if not cl and out.is_contiguous() and _fused_dcn_forward(input, weight.contiguous(), offset, None, None, out, kH, kW, padH, padW, dH, dW,
                                                             dilationH, dilationW, group, deformable_group):
        return 1
    if (out.is_contiguous() or is_channels_last(out)) and _nhwc_ok(input, weight, group, deformable_group):
        kept = _nhwc_forward(input, weight, offset, None, None, out, (kH, kW, padH, padW, dH, dW, dilationH, dilationW, deformable_group))
        if keep is not None:
            keep.append(kept)
        return 1
    weight = weight.contiguous()
    input = input.contiguous()
    for b0 in range(0, B, im2col_step):
        sl = slice(b0, b0 + im2col_step)
        col = deformable_im2col(input[sl], offset[sl], None, kH, kW, padH, padW, dH, dW, dilationH,
                                dilationW, deformable_group)
        buf = torch.empty((Cout, im2col_step * Ho * Wo), dtype=input.dtype, device=input.device)
        _grouped_weight_times_cols(weight, col, group, buf)
        out[sl].copy_(buf.view(Cout, im2col_step, Ho, Wo).transpose(0, 1))
    return 1


def deform_conv_backward_input(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW,
                               kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group,
                               im2col_step):
    """reference csrc/deform_conv.h:45-77 / deform_conv_cuda.cu:268-380: accumulates into the
    caller-zeroed gradInput, overwrites gradOffset.  Returns 1."""
    _dcn_check("deform_conv_backward_input", input, offset, gradOutput, gradInput, gradOffset, weight)
    input, offset = input.contiguous(), offset.contiguous()
    gradOutput, weight = gradOutput.contiguous(), weight.contiguous()
    Ho, Wo = _shape_check(input, offset, gradOutput, weight, kH, kW, dH, dW, padH, padW, dilationH,
                          dilationW, group, deformable_group)
    B, C = input.shape[:2]
    Cout = weight.size(0)
    if gradInput.is_contiguous() and gradOffset.is_contiguous() and _nhwc_ok(
            input, weight, group, deformable_group, (kH, kW, padH, padW, dH, dW, dilationH, dilationW, deformable_group)):
        _nhwc_backward(input, weight, offset, None, gradOutput, gradInput, gradOffset, None, None, None,
                       (kH, kW, padH, padW, dH, dW, dilationH, dilationW, deformable_group))
        return 1
    Mg, Kg = Cout // group, (C // group) * kH * kW
    w2 = weight.reshape(group, Mg, Kg)
    for b0 in range(0, B, im2col_step):
        sl = slice(b0, b0 + im2col_step)
        go = gradOutput[sl].transpose(0, 1).reshape(Cout, im2col_step * Ho * Wo)
        col = torch.empty((C * kH * kW, im2col_step * Ho * Wo), dtype=input.dtype, device=input.device)
        for g in range(group):  # columns = W^T * gradOut  (:338-341)
            torch.mm(w2[g].t(), go[g * Mg:(g + 1) * Mg], out=col[g * Kg:(g + 1) * Kg])
        deformable_col2im_coord(col, input[sl], offset[sl], None, gradOffset[sl], None, kH, kW, padH,
                                padW, dH, dW, dilationH, dilationW, deformable_group)
        deformable_col2im(col, offset[sl], None, gradInput[sl], kH, kW, padH, padW, dH, dW, dilationH,
                          dilationW, deformable_group)
    return 1


def deform_conv_backward_parameters(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW,
                                    dH, padW, padH, dilationW, dilationH, group, deformable_group,
                                    scale, im2col_step):
    """reference csrc/deform_conv.h:80-112 / deform_conv_cuda.cu:382-494:
    gradWeight += scale * gradOut * cols^T.  Returns 1."""
    _dcn_check("deform_conv_backward_parameters", input, offset, gradOutput, gradWeight)
    input, offset, gradOutput = input.contiguous(), offset.contiguous(), gradOutput.contiguous()
    Ho, Wo = _shape_check(input, offset, gradOutput, gradWeight, kH, kW, dH, dW, padH, padW,
                          dilationH, dilationW, group, deformable_group)
    B, C = input.shape[:2]
    Cout = gradWeight.size(0)
    if gradWeight.is_contiguous() and _nhwc_ok(input, gradWeight, group, deformable_group):
        _nhwc_backward(input, None, offset, None, gradOutput, None, None, None, gradWeight, None,
                       (kH, kW, padH, padW, dH, dW, dilationH, dilationW, deformable_group), scale=scale)
        return 1
    Mg, Kg = Cout // group, (C // group) * kH * kW
    gw = gradWeight.view(group, Mg, Kg)
    for b0 in range(0, B, im2col_step):
        sl = slice(b0, b0 + im2col_step)
        col = deformable_im2col(input[sl], offset[sl], None, kH, kW, padH, padW, dH, dW, dilationH,
                                dilationW, deformable_group)
        go = gradOutput[sl].transpose(0, 1).reshape(Cout, im2col_step * Ho * Wo)
        for g in range(group):  # :466-472
            gw[g].addmm_(go[g * Mg:(g + 1) * Mg], col[g * Kg:(g + 1) * Kg].t(), beta=1.0,
                         alpha=float(scale))
    return 1


def modulated_deform_conv_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h,
                                  kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                                  group, deformable_group, with_bias, keep=None):
    """reference csrc/deform_conv.h:115-149 / deform_conv_cuda.cu:496-575 (H-before-W order).
    `keep` (extension): see deform_conv_forward.
    The reference loops per image; one im2col + one GEMM per group over the whole batch gives the
    same sums."""
    _dcn_check("modulated_deform_conv_forward", input, weight, offset, mask, output)
    cl = is_channels_last(input)
    if not input.is_contiguous() and not cl:
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous() and not is_channels_last(weight):
        raise RuntimeError("weight tensor has to be contiguous")   # (or channels-last: a model switched to channels-last carries its 4-d parameters that way)
    B, C, H, W = input.shape
    Cout, Cker, kh_, kw_ = weight.shape
    if kh_ != kernel_h or kw_ != kernel_w:
        raise RuntimeError("Input shape and kernel shape wont match: (%d x %d vs %d x %d)."
                           % (kernel_h, kernel_w, kh_, kw_))
    if C != Cker * group:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (C, Cker * group))
    Ho, Wo = _out_hw(H, W, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w, dilation_h, dilation_w)
    offset, mask = offset.contiguous(), mask.contiguous()
    out = output.view(B, Cout, Ho, Wo)
    if not cl and out.is_contiguous() and _fused_dcn_forward(input, weight.contiguous(), offset, mask, bias.to(input.dtype).contiguous() if with_bias else None,
                                                             out, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w,
                                                             dilation_h, dilation_w, group, deformable_group):
        return
    if (out.is_contiguous() or is_channels_last(out)) and _nhwc_ok(input, weight, group, deformable_group):
        kept = _nhwc_forward(input, weight, offset, mask, bias if with_bias else None, out,
                             (kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w, dilation_h, dilation_w, deformable_group))
        if keep is not None:
            keep.append(kept)
        return
    input, weight = input.contiguous(), weight.contiguous()
    col = deformable_im2col(input, offset, mask, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w,
                            dilation_h, dilation_w, deformable_group)
    buf = torch.empty((Cout, B * Ho * Wo), dtype=input.dtype, device=input.device)
    _grouped_weight_times_cols(weight, col, group, buf)
    out = output.view(B, Cout, Ho, Wo)
    out.copy_(buf.view(Cout, B, Ho, Wo).transpose(0, 1))
    if with_bias:
        out += bias.view(1, -1, 1, 1)


def modulated_deform_conv_backward(input, weight, bias, ones, offset, mask, columns, grad_input,
                                   grad_weight, grad_bias, grad_offset, grad_mask, grad_output,
                                   kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
                                   dilation_w, group, deformable_group, with_bias):
    """reference csrc/deform_conv.h:152-191 / deform_conv_cuda.cu:577-691.  grad_input /
    grad_weight / grad_bias are accumulated into (caller passes zeros); grad_offset / grad_mask are
    overwritten."""
    _dcn_check("modulated_deform_conv_backward", input, weight, offset, mask, grad_input, grad_weight,
               grad_offset, grad_mask, grad_output)
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    B, C, H, W = input.shape
    Cout, Cker, kh_, kw_ = weight.shape
    if kh_ != kernel_h or kw_ != kernel_w:
        raise RuntimeError("Input shape and kernel shape wont match: (%d x %d vs %d x %d)."
                           % (kernel_h, kernel_w, kh_, kw_))
    if C != Cker * group:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (C, Cker * group))
    Ho, Wo = _out_hw(H, W, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w, dilation_h, dilation_w)
    offset, mask, grad_output = offset.contiguous(), mask.contiguous(), grad_output.contiguous()
    geom = (kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w, dilation_h, dilation_w,
            deformable_group)
    if (grad_input.is_contiguous() and grad_weight.is_contiguous() and grad_offset.is_contiguous() and grad_mask.is_contiguous()
            and _nhwc_ok(input, weight, group, deformable_group, geom)):
        _nhwc_backward(input, weight, offset, mask, grad_output, grad_input, grad_offset, grad_mask, grad_weight,
                       grad_bias if with_bias else None, geom)
        return
    Mg, Kg = Cout // group, Cker * kernel_h * kernel_w
    w2 = weight.reshape(group, Mg, Kg)
    go = grad_output.transpose(0, 1).reshape(Cout, B * Ho * Wo)
    col = torch.empty((C * kernel_h * kernel_w, B * Ho * Wo), dtype=input.dtype, device=input.device)
    for g in range(group):  # columns = W^T * gradOut (:628-631)
        torch.mm(w2[g].t(), go[g * Mg:(g + 1) * Mg], out=col[g * Kg:(g + 1) * Kg])
    deformable_col2im_coord(col, input, offset, mask, grad_offset, grad_mask, *geom)
    deformable_col2im(col, offset, mask, grad_input, *geom)
    col = deformable_im2col(input, offset, mask, *geom)
    gw = grad_weight.view(group, Mg, Kg)
    for g in range(group):  # :666-670
        gw[g].addmm_(go[g * Mg:(g + 1) * Mg], col[g * Kg:(g + 1) * Kg].t())
    if with_bias:  # :671-676 gradOut * ones
        grad_bias += go.sum(1)


def _psroi_args(name, input, bbox, trans, no_trans, part_size):
    _need_cuda(name, input, bbox)
    if not input.is_contiguous():  # deform_pool_cuda.cu:45
        raise RuntimeError("input tensor has to be contiguous")
    input = _f32c(name, input)
    bbox = _f32c(name, bbox)
    if no_trans:
        return input, bbox, None, 2  # deform_pool_cuda.cu:51 (`channels_trans = no_trans ? 2 : trans.size(1)`)
    _need_cuda(name, trans)
    trans = _f32c(name, trans)
    if trans.dim() != 4 or trans.size(0) != bbox.size(0) or trans.size(2) != part_size or trans.size(3) != part_size:
        raise RuntimeError("%s: trans must be [num_bbox, 2*num_classes, part_size, part_size]" % name)
    return input, bbox, trans, trans.size(1)


def deform_psroi_pooling_forward(input, bbox, trans, out, top_count, no_trans, spatial_scale, output_dim,
                                 group_size, pooled_size, part_size, sample_per_part, trans_std):
    """reference csrc/deform_pool.h:11-38 -> deform_pool_cuda.cu:38-62.  Writes into the
    caller-allocated `out` / `top_count` [num_bbox, output_dim, pooled_size, pooled_size]."""
    name = "deform_psroi_pooling_forward"
    input, bbox, trans, ct = _psroi_args(name, input, bbox, trans, no_trans, part_size)
    N, C, H, W = input.shape
    K = bbox.size(0)
    if K != out.size(0):  # deform_pool_cuda.cu:54-56
        raise RuntimeError("Output shape and bbox number wont match: (%d vs %d)." % (out.size(0), K))
    _need_cuda(name, out, top_count)
    if not (out.is_contiguous() and top_count.is_contiguous() and out.dtype == torch.float32
            and top_count.dtype == torch.float32):
        raise RuntimeError("%s: out/top_count must be contiguous float32" % name)
    with _on_device(input), _timed(("psroi_fwd[K=%d,D=%d,P=%d]", (K, output_dim, pooled_size)), input):
        check(lib.detops_deform_psroi_pool_forward_f32(
            ptr(input), ptr(bbox), ptr(trans), ptr(out), ptr(top_count), N, C, H, W, K, ct,
            int(bool(no_trans)), float(spatial_scale), output_dim, group_size, pooled_size, part_size,
            sample_per_part, float(trans_std), stream_of(input)), name)


def deform_psroi_pooling_backward(out_grad, input, bbox, trans, top_count, input_grad, trans_grad,
                                  no_trans, spatial_scale, output_dim, group_size, pooled_size,
                                  part_size, sample_per_part, trans_std):
    """reference csrc/deform_pool.h:41-70 -> deform_pool_cuda.cu:64-87.  Accumulates into the
    caller's (zeroed) `input_grad` / `trans_grad` like the reference's atomicAdd kernel."""
    name = "deform_psroi_pooling_backward"
    _need_cuda(name, out_grad, top_count, input_grad)
    if not out_grad.is_contiguous():  # deform_pool_cuda.cu:71
        raise RuntimeError("out_grad tensor has to be contiguous")
    out_grad = _f32c(name, out_grad)
    input, bbox, trans, ct = _psroi_args(name, input, bbox, trans, no_trans, part_size)
    top_count = _f32c(name, top_count)
    N, C, H, W = input.shape
    K = bbox.size(0)
    if K != out_grad.size(0):  # deform_pool_cuda.cu:80-82
        raise RuntimeError("Output shape and bbox number wont match: (%d vs %d)." % (out_grad.size(0), K))
    if not (input_grad.is_contiguous() and input_grad.dtype == torch.float32):
        raise RuntimeError("%s: input_grad must be contiguous float32" % name)
    if not no_trans and not (on_device(trans_grad) and trans_grad.is_contiguous()
                             and trans_grad.dtype == torch.float32 and trans_grad.shape == trans.shape):
        raise RuntimeError("%s: trans_grad must be a contiguous float32 CUDA tensor shaped like trans" % name)
    with _on_device(input), _timed(("psroi_bwd[K=%d,D=%d,P=%d]", (K, output_dim, pooled_size)), input):
        check(lib.detops_deform_psroi_pool_backward_f32(
            ptr(out_grad), ptr(input), ptr(bbox), ptr(trans), ptr(top_count), ptr(input_grad),
            None if no_trans else ptr(trans_grad), N, C, H, W, K, ct, int(bool(no_trans)),
            float(spatial_scale), output_dim, group_size, pooled_size, part_size, sample_per_part,
            float(trans_std), 0, stream_of(input)), name)
