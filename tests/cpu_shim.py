"""TEST-ONLY stand-ins for the HIP operators, backed by the CPU oracle.

The product package has no CPU path (its `_C` raises on CPU tensors).  To exercise the *host logic*
of the model surface (proposal selection, target assignment, samplers, losses, the DDP step) in the
`-m "not gpu"` suite, `install()` monkeypatches the handful of `_C` entry points the model calls
with implementations that run the oracle (oracle/detops_oracle.c) on numpy copies.  Nothing outside
tests/ imports this module.
"""
import contextlib
import os

import numpy as np
import torch

import oracle
from maskrcnn_benchmark import _C


def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy())


def _roi_align_forward(input, rois, scale, ph, pw, sr):
    return torch.from_numpy(oracle.roi_align_forward(_np(input.float()), _np(rois.float()), float(scale), ph, pw, sr))


def _roi_align_backward(grad, rois, scale, ph, pw, N, C, H, W, sr):
    return torch.from_numpy(oracle.roi_align_backward(_np(grad.float()), _np(rois.float()), float(scale), ph, pw,
                                                      N, C, H, W, sr))


def _fpn_forward(inputs, rois, scales, ph, pw, sr, k_min, k_max, canonical_scale=224.0, canonical_level=4.0,
                 eps=1e-6, out_channels_last=False):
    r = _np(rois.float())
    lv = oracle.fpn_level(r, k_min, k_max, canonical_scale, canonical_level, eps)
    K, C = r.shape[0], inputs[0].shape[1]
    out = np.zeros((K, C, ph, pw), np.float32)
    for l, (f, s) in enumerate(zip(inputs, scales)):
        sel = np.nonzero(lv == l)[0]
        if sel.size:
            out[sel] = oracle.roi_align_forward(_np(f.float()), r[sel], float(s), ph, pw, sr)
    return torch.from_numpy(out), torch.from_numpy(lv.astype(np.int32))


def _fpn_backward(grad, rois, levels, shapes, scales, ph, pw, sr, prepared=None, channels_last=False):
    r, g, lv = _np(rois.float()), _np(grad.float()), _np(levels)
    outs = []
    for l, (shp, s) in enumerate(zip(shapes, scales)):
        sel = np.nonzero(lv == l)[0]
        N, C, H, W = shp
        if sel.size:
            outs.append(torch.from_numpy(oracle.roi_align_backward(g[sel], r[sel], float(s), ph, pw, N, C, H, W, sr)))
        else:
            outs.append(torch.zeros(shp))
    return outs


def _nms(dets, scores, thr):
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.long)
    return torch.from_numpy(oracle.nms(_np(dets.float()), _np(scores.float()), float(thr)).astype(np.int64))


def _nms_batched_mask(boxes, scores, seg_offsets, max_n, thr):
    b, s, seg = _np(boxes.float()), _np(scores.float()), _np(seg_offsets)
    mask = np.zeros(b.shape[0], bool)
    num = np.zeros(len(seg) - 1, np.int32)
    for i in range(len(seg) - 1):
        lo, hi = int(seg[i]), int(seg[i + 1])
        if hi > lo:
            k = oracle.nms(b[lo:hi], s[lo:hi], float(thr))
            mask[lo + k] = True
            num[i] = len(k)
    return torch.from_numpy(mask), torch.from_numpy(num)


def _focal_sum(logits, targets, num_classes, gamma, alpha):
    return torch.from_numpy(oracle.sigmoid_focal_loss_forward(_np(logits.float()), _np(targets), gamma, alpha)).sum()


def _focal_bwd_scalar(logits, targets, d_loss, num_classes, gamma, alpha):
    d = np.full(tuple(logits.shape), float(d_loss), np.float32)
    return torch.from_numpy(oracle.sigmoid_focal_loss_backward(_np(logits.float()), _np(targets), d, gamma, alpha))


def _frozen_bn_fwd(x, scale, bias, residual, relu):
    y = x * scale.reshape(1, -1, 1, 1).to(x.dtype) + bias.reshape(1, -1, 1, 1).to(x.dtype)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y


def _frozen_bn_bwd(grad_y, y, scale, relu, need_residual):
    g = grad_y * (y > 0).to(grad_y.dtype) if relu else grad_y
    return g * scale.reshape(1, -1, 1, 1).to(g.dtype), (g.clone() if need_residual else None)


_PATCHES = {
    "frozen_bn_act_forward": _frozen_bn_fwd,
    "frozen_bn_act_backward": _frozen_bn_bwd,
    "roi_align_forward": _roi_align_forward,
    "roi_align_backward": _roi_align_backward,
    "roi_align_fpn_forward": _fpn_forward,
    "roi_align_fpn_backward": _fpn_backward,
    "nms": _nms,
    "nms_batched_mask": _nms_batched_mask,
    "sigmoid_focalloss_forward_sum": _focal_sum,
    "sigmoid_focalloss_backward_scalar": _focal_bwd_scalar,
}


# ---------------------------------------------------------------------------------------------
# Second backend: the HIP kernel SOURCES executed by the host emulation (tests/emu) instead of the
# oracle — the same entry points the GPU build exports, called with the model's real argument
# patterns (proposal distributions, padded ROI sets, multi-level launches).
# ---------------------------------------------------------------------------------------------
def _emu_patches():
    import emu

    def fpn_forward(inputs, rois, scales, ph, pw, sr, k_min, k_max, canonical_scale=224.0, canonical_level=4.0, eps=1e-6,
                    out_channels_last=False):
        out, lv = emu.roi_align_fpn_forward([_np(f.float()) for f in inputs], _np(rois.float()),
                                            [float(s) for s in scales], ph, pw, sr, k_min, k_max)
        return torch.from_numpy(out), torch.from_numpy(lv)

    def fpn_backward(grad, rois, levels, shapes, scales, ph, pw, sr, prepared=None, channels_last=False):
        outs = emu.roi_align_fpn_backward(_np(grad.float()), _np(rois.float()), _np(levels), [tuple(s) for s in shapes],
                                          [float(s) for s in scales], ph, pw, sr)
        return [torch.from_numpy(o) for o in outs]

    def nms_batched_mask(boxes, scores, seg_offsets, max_n, thr):
        km, num = emu.nms_batched(_np(boxes.float()), _np(scores.float()), _np(seg_offsets), int(max_n), float(thr), mask=True)
        return torch.from_numpy(km.astype(bool)), torch.from_numpy(num)

    def nms(dets, scores, thr):
        if dets.numel() == 0:
            return torch.empty((0,), dtype=torch.long)
        return torch.from_numpy(emu.nms(_np(dets.float()), _np(scores.float()), float(thr)))

    def focal_sum(logits, targets, num_classes, gamma, alpha):
        return torch.tensor(emu.focal_forward(_np(logits.float()), _np(targets), float(gamma), float(alpha), with_sum=True)[1])

    def focal_bwd_scalar(logits, targets, d_loss, num_classes, gamma, alpha):
        return torch.from_numpy(emu.focal_backward(_np(logits.float()), _np(targets), np.float32(float(d_loss)),
                                                   float(gamma), float(alpha)))

    def bn_fwd(x, scale, bias, residual, relu):
        return torch.from_numpy(emu.frozen_bn_forward(_np(x.float()), _np(scale), _np(bias),
                                                      None if residual is None else _np(residual.float()), bool(relu)))

    def bn_bwd(grad_y, y, scale, relu, need_residual):
        yy = grad_y if y is None else y   # y is only read for the ReLU mask
        gx, gr = emu.frozen_bn_backward(_np(grad_y.float()), _np(yy.float()), _np(scale), bool(relu), bool(need_residual))
        return torch.from_numpy(gx), (None if gr is None else torch.from_numpy(gr))

    return {"roi_align_fpn_forward": fpn_forward, "roi_align_fpn_backward": fpn_backward, "nms": nms,
            "nms_batched_mask": nms_batched_mask, "sigmoid_focalloss_forward_sum": focal_sum,
            "sigmoid_focalloss_backward_scalar": focal_bwd_scalar, "frozen_bn_act_forward": bn_fwd,
            "frozen_bn_act_backward": bn_bwd}


def _emu_device_patches():
    """The target-assignment / proposal kernels of csrc/targets.hip under the host emulation, plus `on_device` -> True:
    the model takes the branches it takes on the GPU (fused labels, sampler, sampled-slot targets, proposal decode,
    batched hand-over of the proposals) with CPU tensors."""
    import emu
    seed = [0]

    def match_boxes(gt_boxes, gt_valid, boxes, high, low, allow_lq):
        return torch.from_numpy(emu.match_boxes(_np(gt_boxes), _np(gt_valid), _np(boxes), float(high), float(low), bool(allow_lq)))

    def sample_labels(labels, B, max_pos, with_list=False, seed_=None):
        seed[0] += 1
        pos, neg, idx, val = emu.sample_labels(_np(labels), int(B), int(max_pos), seed=seed[0] if seed_ is None else seed_)
        out = (torch.from_numpy(pos), torch.from_numpy(neg))
        return out + (torch.from_numpy(idx), torch.from_numpy(val)) if with_list else out

    def match_labels(matched, gt_labels=None, valid=None, dtype=torch.int64):
        return torch.from_numpy(emu.match_labels(_np(matched), None if gt_labels is None else _np(gt_labels),
                                                 None if valid is None else _np(valid),
                                                 np.float32 if dtype == torch.float32 else np.int64))

    def roi_head_targets(boxes, matched, gt_boxes, gt_labels, valid, idx, slot_valid, objectness, weights):
        out = emu.roi_head_targets(_np(boxes), _np(matched), _np(gt_boxes), _np(gt_labels), None if valid is None else _np(valid),
                                   _np(idx), _np(slot_valid), None if objectness is None else _np(objectness), weights)
        return tuple(None if o is None else torch.from_numpy(o) for o in out)

    def rpn_decode(box_regression, topk_idx, topk_scores, anchors, image_hw, weights, clip, min_size, boxes, scores, col,
                   nms_boxes, nms_scores, ok, off):
        b, s, nb, ns, okk = emu.rpn_decode(_np(box_regression), _np(topk_idx), _np(topk_scores), _np(anchors), _np(image_hw),
                                           weights, clip, min_size)
        N, k = s.shape
        boxes[:, col:col + k] = torch.from_numpy(b)
        scores[:, col:col + k] = torch.from_numpy(s)
        nms_boxes[off:off + N * k] = torch.from_numpy(nb)
        nms_scores[off:off + N * k] = torch.from_numpy(ns)
        ok[off:off + N * k] = torch.from_numpy(okk.astype(np.uint8))

    def mask_targets(masks, mask_index, boxes, M):
        m = _np(masks)
        if m.dtype not in (np.uint8, np.float32, np.bool_):
            m = m.astype(np.uint8)
        return torch.from_numpy(emu.mask_targets(m, _np(mask_index), _np(boxes), int(M)))

    class _EmuHeadLoss(torch.autograd.Function):
        """the autograd contract of _C._HeadLoss with the kernels of csrc/head_loss.hip under the host emulation"""

        @staticmethod
        def forward(ctx, kind, aux, *inputs):
            ctx.kind, ctx.aux, ctx.n_in = kind, aux, len(inputs)
            ctx.save_for_backward(*inputs)
            if kind == "fastrcnn":
                lc, lb, _, _ = emu.fastrcnn_loss(*[_np(t) for t in inputs], aux[0], aux[1])
                return torch.tensor(lc), torch.tensor(lb)
            return (torch.tensor(emu.mask_loss(*[_np(t) for t in inputs])[0]),)

        @staticmethod
        def backward(ctx, *ups):
            args = [_np(t) for t in ctx.saved_tensors]
            up = [0.0 if u is None else float(u) for u in ups]
            if ctx.kind == "fastrcnn":
                _, _, gl, gb = emu.fastrcnn_loss(*args, ctx.aux[0], ctx.aux[1], upstream=(up[0], up[1]))
                return (None, None, torch.from_numpy(gl), torch.from_numpy(gb), None, None)
            return (None, None, torch.from_numpy(emu.mask_loss(*args, upstream=up[0])[1]), None, None)

    class _EmuRpnLoss(torch.autograd.Function):
        @staticmethod
        def forward(ctx, L, meta, *heads):
            anchors, matched, pos, neg, gt, beta, weights = meta
            lo, lb, gobj, gbox = emu.rpn_loss([_np(t) for t in heads[:L]], [_np(t) for t in heads[L:]], anchors, matched, pos, neg,
                                              gt, beta, weights)
            ctx.grads = [torch.from_numpy(g) for g in gobj], [torch.from_numpy(g) for g in gbox]
            return torch.tensor(lo), torch.tensor(lb)

        @staticmethod
        def backward(ctx, uo, ub):
            uo = 0.0 if uo is None else float(uo)
            ub = 0.0 if ub is None else float(ub)
            return (None, None) + tuple(g * uo for g in ctx.grads[0]) + tuple(g * ub for g in ctx.grads[1])

    def rpn_loss(objectness, box_regression, anchors, matched_idxs, pos_mask, neg_mask, gt_boxes, beta, weights):
        meta = (_np(anchors), _np(matched_idxs), _np(pos_mask), _np(neg_mask), _np(gt_boxes), float(beta), tuple(weights))
        return _EmuRpnLoss.apply(len(objectness), meta, *[t.float() for t in list(objectness) + list(box_regression)])

    def fastrcnn_loss(class_logits, box_regression, labels, regression_targets, cls_agnostic=False, beta=1.0):
        return _EmuHeadLoss.apply("fastrcnn", (bool(cls_agnostic), float(beta)), class_logits.float(), box_regression.float(),
                                  labels, regression_targets)

    def mask_loss(mask_logits, labels, mask_targets):
        return _EmuHeadLoss.apply("mask", None, mask_logits.float(), labels, mask_targets)[0]

    class _EmuTopDown(torch.autograd.Function):
        """_C._UpsampleAdd's contract with the kernels of csrc/fpn_topdown.hip under the host emulation (fp32)"""

        @staticmethod
        def forward(ctx, lateral, top):
            ctx.hw = (int(top.shape[2]), int(top.shape[3]))
            return torch.from_numpy(emu.fpn_topdown_forward(_np(lateral.float()), _np(top.float()))).to(lateral.dtype)

        @staticmethod
        def backward(ctx, g):
            return g, torch.from_numpy(emu.fpn_topdown_backward(_np(g.float()), *ctx.hw)).to(g.dtype)

    def fpn_topdown(lateral, top):
        return _EmuTopDown.apply(lateral, top)

    return {"on_device": lambda t: True, "rpn_loss": rpn_loss, "fastrcnn_loss": fastrcnn_loss, "mask_loss": mask_loss, "match_boxes": match_boxes, "sample_labels": sample_labels,
            "fpn_topdown": fpn_topdown,
            "match_labels": match_labels, "roi_head_targets": roi_head_targets, "rpn_decode": rpn_decode,
            "mask_targets": mask_targets}


def _emu_lib_patches():
    """The PRODUCT's own `_C` wrappers (argument checks, marshalling, workspaces, autograd functions) on CPU tensors: the
    library handle they call is the host-emulation build of the same HIP sources (same C ABI, host pointers), the
    CUDA-only guards are lifted and `on_device` says yes.  Nothing of `_C`'s operator surface is replaced."""
    import ctypes

    import emu
    from maskrcnn_benchmark import _lib
    emu.lib()                                   # builds tests/emu/libdetops_emu.so when needed
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(emu.__file__)), "libdetops_emu.so"))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:                      # the two *_cpu_* entry points and detops_version exist in the device build only
            fn.restype, fn.argtypes = res, args
    return {"lib": lib, "_need_cuda": lambda name, *tensors: None, "stream_of": lambda t: None,
            "_on_device": lambda t: _C._NOSPAN, "on_device": lambda t: True}


@contextlib.contextmanager
def install(backend="oracle"):
    """backend = "oracle" (C restatement), "emu" (the HIP sources under the host emulation), "emu-device" (emu + the
    model's device-only branches, see `_emu_device_patches`) or "emu-lib" (the product's own `_C` wrappers over the
    emulation library, see `_emu_lib_patches`)."""
    patches = dict(_PATCHES)
    if backend in ("emu", "emu-device"):
        patches.update(_emu_patches())
    if backend == "emu-device":
        patches.update(_emu_device_patches())
    if backend == "emu-lib":
        patches = _emu_lib_patches()
    saved = {k: getattr(_C, k) for k in patches}
    try:
        for k, v in patches.items():
            setattr(_C, k, v)
        yield
    finally:
        for k, v in saved.items():
            setattr(_C, k, v)
