"""TEST-ONLY: the HIP kernels compiled as host C++ and executed by a fiber-based emulation of the
HIP execution model (tests/emu/hip_cpu_emu.h).  Same C ABI as libdetops_gfx950.so, host pointers
instead of device pointers.  Used by tests/test_emu_kernels.py to check kernel LOGIC (indexing,
barriers, LDS layouts) against the oracle in the GPU-less container; parity proper is `-m gpu`."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_int, c_float, _P = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
SIGNATURES = {
    "detops_tuning_set": (c_int, [ctypes.c_char_p, c_int]),
    "detops_tuning_get": (c_int, [ctypes.c_char_p, _P]),
    "detops_roi_align_forward_f32": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, _P]),
    "detops_roi_align_backward_f32": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, c_int, _P]),
    "detops_roi_align_fpn_forward_f32": (
        c_int, [_P, _P, _P, _P, c_int, _P, _P, _P] + [c_int] * 8 + [c_float, c_float, c_float, _P]),
    "detops_roi_align_fpn_backward_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "detops_roi_align_fpn_backward_ws_f32": (
        c_int, [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 8 + [_P, ctypes.c_size_t, _P]),
    "detops_roi_align_fpn_backward_prepare_f32": (c_int, [_P] * 5 + [c_int] * 7 + [_P, ctypes.c_size_t, _P]),
    "detops_roi_align_fpn_backward_prepared_f32": (c_int, [_P] * 5 + [c_int] * 7 + [_P, ctypes.c_size_t, _P]),
    "detops_roi_align_backward_ws_f32": (
        c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, c_int, _P, ctypes.c_size_t, _P]),
    "detops_roi_align_backward_workspace_bytes": (ctypes.c_size_t, [_P, _P] + [c_int] * 6),
    "detops_roi_align_forward_workspace_bytes": (ctypes.c_size_t, [c_int] * 4),
    "detops_roi_align_forward_ws_f32": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, _P, ctypes.c_size_t, _P]),
    "detops_roi_align_fpn_forward_ws_f32": (
        c_int, [_P, _P, _P, _P, c_int, _P, _P, _P] + [c_int] * 8 + [c_float, c_float, c_float, _P, ctypes.c_size_t, _P]),
    "detops_match_boxes_workspace_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "detops_match_boxes_f32": (c_int, [_P, _P, _P] + [c_int] * 4 + [c_float, c_float, c_int, _P, _P, ctypes.c_size_t, _P]),
    "detops_sample_labels_workspace_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "detops_sample_labels": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint64, _P, _P, _P, _P, _P,
                                     ctypes.c_size_t, _P]),
    "detops_sample_labels_dseed": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint64, _P, _P, _P, _P, _P, _P,
                                     ctypes.c_size_t, _P]),
    "detops_mask_targets": (c_int, [_P, c_int, _P, _P] + [c_int] * 5 + [_P, _P]),
    "detops_match_labels": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "detops_roi_head_targets_f32": (c_int, [_P] * 8 + [c_int] * 4 + [c_float] * 4 + [_P] * 6),
    "detops_fastrcnn_loss_workspace_bytes": (ctypes.c_size_t, [c_int]),
    "detops_fastrcnn_loss_f32": (c_int, [_P] * 4 + [c_int] * 4 + [c_float] + [_P] * 4 + [ctypes.c_size_t, _P]),
    "detops_mask_loss_workspace_bytes": (ctypes.c_size_t, [c_int]),
    "detops_mask_loss_f32": (c_int, [_P] * 3 + [c_int] * 3 + [_P] * 3 + [ctypes.c_size_t, _P]),
    "detops_head_loss_backward_f32": (c_int, [_P, ctypes.c_int64, _P, _P, ctypes.c_int64, _P, _P]),
    "detops_rpn_decode_f32": (c_int, [_P] * 5 + [c_int] * 5 + [c_float] * 6 + [_P, ctypes.c_int64, _P, ctypes.c_int64, _P, _P, _P, _P]),
    "detops_rpn_loss_workspace_bytes": (ctypes.c_size_t, []),
    "detops_rpn_loss_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float,
                                    _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "detops_rpn_loss_backward_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "detops_roi_pool_forward_f32": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [c_float, _P]),
    "detops_roi_pool_backward_f32": (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "detops_nms_workspace_bytes": (ctypes.c_size_t, [c_int]),
    "detops_nms_f32": (c_int, [_P, _P, c_int, c_float, _P, _P, _P, ctypes.c_size_t, _P]),
    "detops_nms_batched_workspace_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "detops_nms_batched_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P, _P, ctypes.c_size_t, _P]),
    "detops_nms_batched_mask_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P, _P, ctypes.c_size_t, _P]),
    "detops_sigmoid_focal_loss_forward_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "detops_sigmoid_focal_loss_backward_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "detops_sigmoid_focal_loss_backward_scalar_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "detops_sigmoid_focal_loss_forward_sum_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "detops_sigmoid_focal_loss_sum_workspace_bytes": (ctypes.c_size_t, []),
    "detops_sigmoid_focal_loss_forward_sum_ws_f32": (
        c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P, ctypes.c_size_t, _P]),
    "detops_sigmoid_focal_loss_forward_partial_sums_f32": (
        c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P]),
    "detops_frozen_bn_act_forward": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [_P]),
    "detops_frozen_bn_act_backward": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [_P]),
    "detops_deform_psroi_pool_forward_f32": (c_int, [_P] * 5 + [c_int] * 7 + [c_float] + [c_int] * 5 + [c_float, _P]),
    "detops_deform_psroi_pool_backward_f32": (
        c_int, [_P] * 7 + [c_int] * 7 + [c_float] + [c_int] * 5 + [c_float, c_int, _P]),
    "detops_deformable_im2col": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P]),
    "detops_deformable_col2im": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P]),
    "detops_deformable_col2im_workspace_bytes": (ctypes.c_size_t, [c_int] * 13),
    "detops_deformable_col2im_ws": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P, ctypes.c_size_t, _P]),
    "detops_deform_conv_forward_fused_workspace_bytes": (ctypes.c_size_t, [c_int] * 15),
    "detops_deform_conv_forward_fused": (c_int, [_P] * 6 + [c_int] * 15 + [_P, ctypes.c_size_t, _P]),
    "detops_nchw_to_nhwc": (c_int, [_P, _P] + [c_int] * 4 + [_P]),
    "detops_deformable_nhwc_supported": (c_int, [c_int] * 4),
    "detops_deformable_im2col_nhwc": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P]),
    "detops_deformable_coord_nhwc": (c_int, [_P] * 6 + [c_int] * 14 + [_P]),
    "detops_deformable_transposed_sample_workspace_bytes": (ctypes.c_size_t, [c_int] * 13),
    "detops_deformable_transposed_sample": (c_int, [_P, _P, _P, _P] + [c_int] * 15 + [_P, ctypes.c_size_t, _P]),
    "detops_deformable_col2im_nhwc": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P, ctypes.c_size_t, _P]),
    "detops_deformable_col2im_coord": (c_int, [_P, _P, _P, _P, _P, _P] + [c_int] * 14 + [_P]),
}


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = ctypes.CDLL(os.path.join(_HERE, "libdetops_emu.so"))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_LIB, name)
            fn.restype, fn.argtypes = res, args
    return _LIB


TUNING_KEYS = ("roi_bwd_impl", "roi_bwd_seg", "nms_fault", "nms_spin_budget", "roi_bwd_ring", "roi_bwd_ct", "roi_bwd_split", "roi_bwd_maxseg", "roi_bwd_extras", "roi_bwd_groups", "roi_bwd_scan_ct", "roi_bwd_debug", "roi_fwd_impl", "roi_fwd_records", "roi_fwd_ct",
               "roi_fwd_order", "roi_fwd_order_mink", "dcn_col2im", "dcn_fused", "dcn_gather_xcd", "dcn_nhwc", "dcn_ell_build", "nms_fused", "nms_no_repair", "nms_no_presorted", "nms_debug")


def tuning_set(key, value):
    """tuning / test switch of the emulated library (include/detops.h: detops_tuning_set)"""
    rc = lib().detops_tuning_set(key.encode(), int(value))
    assert rc == 0, (key, rc)


def tuning_reset():
    if _LIB is not None:
        for k in TUNING_KEYS:
            _LIB.detops_tuning_set(k.encode(), 0)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_P)


def roi_align_forward(inp, rois, scale, ph, pw, sr):
    inp, rois = _f32(inp), _f32(rois)
    N, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.full((K, C, ph, pw), np.nan, np.float32)
    nbytes = lib().detops_roi_align_forward_workspace_bytes(K, ph, pw, sr)
    ws = np.full((max(nbytes, 1),), 0xAB, np.uint8)   # poisoned: the order pre-pass must write every slot
    rc = lib().detops_roi_align_forward_ws_f32(_p(inp), _p(rois), _p(out), N, C, H, W, K, ph, pw, scale, sr,
                                               _p(ws) if nbytes else None, nbytes, None)
    assert rc == 0, rc
    return out


def roi_align_backward(grad, rois, scale, ph, pw, N, C, H, W, sr, into=None):
    """into=None: zero_grad_in=1 on a NaN-filled buffer (every element must be written);
    into=array: accumulate (zero_grad_in=0)."""
    grad, rois = _f32(grad), _f32(rois)
    K = rois.shape[0]
    gin = np.full((N, C, H, W), np.nan, np.float32) if into is None else _f32(into).copy()
    Hs, Ws = (ctypes.c_int * 1)(H), (ctypes.c_int * 1)(W)
    nbytes = lib().detops_roi_align_backward_workspace_bytes(Hs, Ws, 1, N, C, K, ph, pw)
    ws = np.full((max(nbytes, 1),), 0xAB, np.uint8)   # poisoned: the pre-pass must write what the kernel reads
    rc = lib().detops_roi_align_backward_ws_f32(_p(grad), _p(rois), _p(gin), N, C, H, W, K, ph, pw, scale, sr,
                                                1 if into is None else 0, _p(ws) if nbytes else None, nbytes, None)
    assert rc == 0, rc
    return gin


def _host_arrays(arrs, scales):
    L = len(arrs)
    ptrs = (ctypes.c_void_p * L)(*[a.ctypes.data for a in arrs])
    Hs = (ctypes.c_int * L)(*[a.shape[2] for a in arrs])
    Ws = (ctypes.c_int * L)(*[a.shape[3] for a in arrs])
    sc = (ctypes.c_float * L)(*[float(s) for s in scales])
    return ptrs, Hs, Ws, sc


def roi_align_fpn_backward_two_calls(grad, rois, levels, shapes, scales, ph, pw, sr):
    """the ring backward as prepare (forward time) + prepared (backward time): -> list of gradient maps, or None when the
    shape is served by the one-call kernels"""
    grad, rois = _f32(grad), _f32(rois)
    levels = np.ascontiguousarray(levels, dtype=np.int32)
    K = rois.shape[0]
    gins = [np.full(s, np.nan, np.float32) for s in shapes]
    ptrs, Hs, Ws, sc = _host_arrays(gins, scales)
    N, C = shapes[0][:2]
    nbytes = lib().detops_roi_align_backward_workspace_bytes(Hs, Ws, len(shapes), N, C, K, ph, pw)
    if nbytes == 0:
        return None
    ws = np.full((nbytes,), 0xAB, np.uint8)
    rc = lib().detops_roi_align_fpn_backward_prepare_f32(_p(rois), _p(levels), Hs, Ws, sc, len(shapes), N, C, K, ph, pw, sr,
                                                         _p(ws), nbytes, None)
    if rc == -3:
        return None
    assert rc == 0, rc
    rc = lib().detops_roi_align_fpn_backward_prepared_f32(_p(grad), ptrs, Hs, Ws, sc, len(shapes), N, C, K, ph, pw, 1,
                                                          _p(ws), nbytes, None)
    assert rc == 0, rc
    return gins


def roi_align_fpn_forward(feats, rois, scales, ph, pw, sr, k_min, k_max):
    feats = [_f32(f) for f in feats]
    rois = _f32(rois)
    N, C = feats[0].shape[:2]
    K = rois.shape[0]
    out = np.full((K, C, ph, pw), np.nan, np.float32)
    levels = np.full((K,), -1, np.int32)
    ptrs, Hs, Ws, sc = _host_arrays(feats, scales)
    nbytes = lib().detops_roi_align_forward_workspace_bytes(K, ph, pw, sr)
    ws = np.full((max(nbytes, 1),), 0xAB, np.uint8)
    rc = lib().detops_roi_align_fpn_forward_ws_f32(ptrs, Hs, Ws, sc, len(feats), _p(rois), _p(out), _p(levels), N, C, K,
                                                   ph, pw, sr, k_min, k_max, 224.0, 4.0, 1e-6,
                                                   _p(ws) if nbytes else None, nbytes, None)
    assert rc == 0, rc
    return out, levels


def roi_align_fpn_backward(grad, rois, levels, shapes, scales, ph, pw, sr):
    grad, rois = _f32(grad), _f32(rois)
    levels = np.ascontiguousarray(levels, dtype=np.int32)
    gins = [np.full(s, np.nan, np.float32) for s in shapes]
    N, C = shapes[0][:2]
    ptrs, Hs, Ws, sc = _host_arrays(gins, scales)
    nbytes = lib().detops_roi_align_backward_workspace_bytes(Hs, Ws, len(shapes), N, C, rois.shape[0], ph, pw)
    ws = np.full((max(nbytes, 1),), 0xAB, np.uint8)
    rc = lib().detops_roi_align_fpn_backward_ws_f32(_p(grad), _p(rois), _p(levels), ptrs, Hs, Ws, sc, len(shapes), N, C,
                                                    rois.shape[0], ph, pw, sr, 1, _p(ws) if nbytes else None, nbytes,
                                                    None)
    assert rc == 0, rc
    return gins


# ---------------------------------------------------------------------------------- target assignment
def match_boxes(gt, valid, boxes, high, low, allow_lq):
    """gt [N,M,4], valid [N,M] bool, boxes [K,4] or [N,K,4] -> matched [N,K] int64"""
    gt, boxes = _f32(gt), _f32(boxes)
    valid = np.ascontiguousarray(valid, dtype=np.uint8)
    N, M = gt.shape[:2]
    batched = boxes.ndim == 3
    K = boxes.shape[-2]
    out = np.full((N, K), -99, np.int64)
    nbytes = lib().detops_match_boxes_workspace_bytes(N, M)
    ws = np.full((nbytes,), 0xAB, np.uint8)
    rc = lib().detops_match_boxes_f32(_p(gt), _p(valid), _p(boxes), int(batched), N, M, K, high, low, int(allow_lq),
                                      _p(out), _p(ws), nbytes, None)
    assert rc == 0, rc
    return out


def sample_labels(labels, B, max_pos, seed, with_list=True, seed_dev=None):
    """seed_dev: optional 64-bit word the kernels mix into the seed when they run (detops_sample_labels_dseed)"""
    labels = np.ascontiguousarray(labels)
    code = {np.dtype(np.float32): 0, np.dtype(np.int64): 1}[labels.dtype]
    N, n = labels.shape
    pos = np.full((N, n), 7, np.uint8)
    neg = np.full((N, n), 7, np.uint8)
    idx = np.full((N, B), -5, np.int64)
    val = np.full((N, B), 7, np.uint8)
    nbytes = lib().detops_sample_labels_workspace_bytes(N, B)
    ws = np.full((nbytes,), 0xAB, np.uint8)
    if seed_dev is None:
        rc = lib().detops_sample_labels(_p(labels), code, N, n, B, max_pos, seed, _p(pos), _p(neg),
                                        _p(idx) if with_list else None, _p(val) if with_list else None, _p(ws), nbytes, None)
    else:
        word = np.array([seed_dev], dtype=np.uint64)
        rc = lib().detops_sample_labels_dseed(_p(labels), code, N, n, B, max_pos, seed, _p(word), _p(pos), _p(neg),
                                              _p(idx) if with_list else None, _p(val) if with_list else None, _p(ws), nbytes, None)
    assert rc == 0, rc
    return pos.astype(bool), neg.astype(bool), idx, val.astype(bool)


def mask_targets(masks, mask_index, boxes, M):
    masks = np.ascontiguousarray(masks)
    code = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.bool_): 2}[masks.dtype]
    mask_index = np.ascontiguousarray(mask_index, dtype=np.int64)
    boxes = _f32(boxes)
    G, H, W = masks.shape
    P = boxes.shape[0]
    out = np.full((P, M, M), np.nan, np.float32)
    rc = lib().detops_mask_targets(_p(masks), code, _p(mask_index), _p(boxes), G, H, W, P, M, _p(out), None)
    assert rc == 0, rc
    return out


def match_labels(matched, gt_labels=None, valid=None, dtype=np.int64):
    matched = np.ascontiguousarray(matched, dtype=np.int64)
    N, K = matched.shape
    gl = None if gt_labels is None else np.ascontiguousarray(gt_labels, dtype=np.int64)
    va = None if valid is None else np.ascontiguousarray(valid).view(np.uint8)
    out = np.full((N, K), 77, dtype)
    rc = lib().detops_match_labels(_p(matched), None if gl is None else _p(gl), None if va is None else _p(va), N, K,
                                   0 if gl is None else gl.shape[1], 0 if np.dtype(dtype) == np.float32 else 1, _p(out), None)
    assert rc == 0, rc
    return out


def roi_head_targets(boxes, matched, gt, gt_labels, valid, idx, slot_valid, objectness, weights):
    boxes, gt = _f32(boxes), _f32(gt)
    matched, idx, gl = (np.ascontiguousarray(a, dtype=np.int64) for a in (matched, idx, gt_labels))
    sv = np.ascontiguousarray(slot_valid).view(np.uint8)
    va = None if valid is None else np.ascontiguousarray(valid).view(np.uint8)
    ob = None if objectness is None else _f32(objectness)
    N, K = matched.shape
    M, B = gt.shape[1], idx.shape[1]
    o_b = np.full((N, B, 4), np.nan, np.float32)
    o_r = np.full((N, B, 4), np.nan, np.float32)
    o_l = np.full((N, B), 77, np.int64)
    o_m = np.full((N, B), 77, np.int64)
    o_o = None if ob is None else np.full((N, B), np.nan, np.float32)
    rc = lib().detops_roi_head_targets_f32(_p(boxes), _p(matched), _p(gt), _p(gl), None if va is None else _p(va), _p(idx),
                                           _p(sv), None if ob is None else _p(ob), N, K, M, B, *[float(w) for w in weights],
                                           _p(o_b), _p(o_l), _p(o_r), _p(o_m), None if o_o is None else _p(o_o), None)
    assert rc == 0, rc
    return o_b, o_l, o_r, o_m, o_o


def fastrcnn_loss(class_logits, box_regression, labels, regression_targets, cls_agnostic=False, beta=1.0, upstream=(1.0, 1.0)):
    """-> (classification loss, box loss, d/d class_logits, d/d box_regression) with the backward scaling applied."""
    lg, bx, tg = _f32(class_logits), _f32(box_regression), _f32(regression_targets)
    lab = np.ascontiguousarray(labels, dtype=np.int64)
    R, C = lg.shape
    D = bx.shape[1]
    gl = np.full((R, C), np.nan, np.float32)
    gb = np.full((R, D), np.nan, np.float32)
    out = np.full((2,), np.nan, np.float32)
    nbytes = lib().detops_fastrcnn_loss_workspace_bytes(R)
    ws = np.full((nbytes,), 0xAB, np.uint8)
    rc = lib().detops_fastrcnn_loss_f32(_p(lg), _p(bx), _p(lab), _p(tg), R, C, D, int(cls_agnostic), float(beta), _p(gl), _p(gb),
                                        _p(out), _p(ws), nbytes, None)
    assert rc == 0, rc
    ua, ub = np.float32([upstream[0]]), np.float32([upstream[1]])
    rc = lib().detops_head_loss_backward_f32(_p(gl), gl.size, _p(ua), _p(gb), gb.size, _p(ub), None)
    assert rc == 0, rc
    return out[0], out[1], gl, gb


def mask_loss(mask_logits, labels, mask_targets, upstream=1.0):
    lg, tg = _f32(mask_logits), _f32(mask_targets)
    lab = np.ascontiguousarray(labels, dtype=np.int64)
    P, C, M, _ = lg.shape
    g = np.full(lg.shape, np.nan, np.float32)
    out = np.full((1,), np.nan, np.float32)
    nbytes = lib().detops_mask_loss_workspace_bytes(P)
    ws = np.full((nbytes,), 0xAB, np.uint8)
    rc = lib().detops_mask_loss_f32(_p(lg), _p(lab), _p(tg), P, C, M, _p(g), _p(out), _p(ws), nbytes, None)
    assert rc == 0, rc
    ua = np.float32([upstream])
    rc = lib().detops_head_loss_backward_f32(_p(g), g.size, _p(ua), None, 0, None, None)
    assert rc == 0, rc
    return out[0], g


def rpn_decode(box_regression, topk_idx, topk_scores, anchors, image_hw, weights, xform_clip, min_size):
    """-> boxes [N,k,4], scores [N,k], nms_boxes [N*k,4], nms_scores [N*k], ok [N*k] bool for one level."""
    reg, sc, an, hw = _f32(box_regression), _f32(topk_scores), _f32(anchors), _f32(image_hw)
    idx = np.ascontiguousarray(topk_idx, dtype=np.int64)
    N, A4, H, W = reg.shape
    A, k = A4 // 4, idx.shape[1]
    boxes = np.full((N, k, 4), np.nan, np.float32)
    scores = np.full((N, k), np.nan, np.float32)
    nb = np.full((N * k, 4), np.nan, np.float32)
    ns = np.full((N * k,), np.nan, np.float32)
    ok = np.full((N * k,), 7, np.uint8)
    rc = lib().detops_rpn_decode_f32(_p(reg), _p(idx), _p(sc), _p(an), _p(hw), N, A, H, W, k, *[float(w) for w in weights],
                                     float(xform_clip), float(min_size), _p(boxes), 4 * k, _p(scores), k, _p(nb), _p(ns),
                                     _p(ok), None)
    assert rc == 0, rc
    return boxes, scores, nb, ns, ok.astype(bool)


# ---------------------------------------------------------------------------------- deformable conv
_DT = {np.dtype(np.float32): 0, np.dtype(np.float16): 1}


def deform_conv_forward_fused(im, weight, offset, mask, bias, pad, stride, dil, dg):
    """fp16 im [B,C,H,W], weight [Cout,C,kh,kw] -> out [B,Cout,Ho,Wo] fp16 through the fused MFMA kernel
    (the host emulation runs the same fragment-layout arithmetic); None when the shape is outside its plan."""
    im, weight, offset = (np.ascontiguousarray(a, dtype=np.float16) for a in (im, weight, offset))
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.float16)
    bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.float16)
    B, C, H, W = im.shape
    Cout, _, kh, kw = weight.shape
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    geo = (1, B, C, H, W, Cout, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0], dil[1], dg)
    tuning_set("dcn_fused", 1)   # small test shapes: bypass the "is it faster" dispatch rule
    try:
        nbytes = lib().detops_deform_conv_forward_fused_workspace_bytes(*geo)
    finally:
        tuning_set("dcn_fused", 0)
    if nbytes == 0:
        return None
    ws = np.full((nbytes,), 0xAB, np.uint8)
    out = np.full((B, Cout, Ho, Wo), np.nan, np.float16)
    rc = lib().detops_deform_conv_forward_fused(_p(im), _p(weight), _p(offset), None if mask is None else _p(mask),
                                                None if bias is None else _p(bias), _p(out), *geo, _p(ws), nbytes, None)
    assert rc == 0, rc
    return out


def _geom(B, C, H, W, kh, kw, pad, stride, dil, dg):
    return (B, C, H, W, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0], dil[1], dg)


def _out_hw(H, W, kh, kw, pad, stride, dil):
    return ((H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1,
            (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1)


def deformable_im2col(im, offset, mask, kh, kw, pad, stride, dil, dg):
    im = np.ascontiguousarray(im)
    dt = im.dtype
    offset = np.ascontiguousarray(offset, dtype=dt)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=dt)
    B, C, H, W = im.shape
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    col = np.full((C * kh * kw, B * Ho * Wo), np.nan, dt)
    rc = lib().detops_deformable_im2col(_p(im), _p(offset), None if mask is None else _p(mask), _p(col), _DT[dt],
                                        *_geom(B, C, H, W, kh, kw, pad, stride, dil, dg), None)
    assert rc == 0, rc
    return col


def deformable_col2im(col, offset, mask, B, C, H, W, kh, kw, pad, stride, dil, dg, gather=True, into=None, mode=None):
    """mode: "gather" (CSR inverted index), "ell" (experimental fixed-width index) or "scatter"; the
    legacy `gather` flag maps to "gather" / "scatter"."""
    mode = mode or ("gather" if gather else "scatter")
    gather = mode != "scatter"
    col = np.ascontiguousarray(col)
    dt = col.dtype
    offset = np.ascontiguousarray(offset, dtype=dt)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=dt)
    gim = np.zeros((B, C, H, W), dt) if into is None else np.ascontiguousarray(into, dtype=dt).copy()
    g = _geom(B, C, H, W, kh, kw, pad, stride, dil, dg)
    mp = None if mask is None else _p(mask)
    if gather:
        tuning_set("dcn_col2im", {"gather": 1, "scatter": 2, "ell": 3}[mode])
        nbytes = lib().detops_deformable_col2im_workspace_bytes(*g)
        assert nbytes > 0
        ws = np.full((nbytes,), 0xAB, np.uint8)  # arbitrary contents
        rc = lib().detops_deformable_col2im_ws(_p(col), _p(offset), mp, _p(gim), _DT[dt], *g, _p(ws), nbytes, None)
        tuning_set("dcn_col2im", 0)
    else:
        rc = lib().detops_deformable_col2im(_p(col), _p(offset), mp, _p(gim), _DT[dt], *g, None)
    assert rc == 0, rc
    return gim


def nchw_to_nhwc(x):
    x = np.ascontiguousarray(x)
    B, C, H, W = x.shape
    out = np.full((B, H * W, C), np.nan, x.dtype)
    rc = lib().detops_nchw_to_nhwc(_p(x), _p(out), _DT[x.dtype], B, C, H * W, None)
    assert rc == 0, rc
    return out


def deformable_nhwc(im, offset, mask, weight, grad_out, kh, kw, pad, stride, dil, input_grad="col2im"):
    """The channels-last pipeline with numpy standing in for the library GEMMs (what _C.py does with torch.mm):
    -> (out, grad_input, grad_offset, grad_mask, grad_weight), everything in the reference's layouts."""
    im = np.ascontiguousarray(im)
    dt = im.dtype
    offset = np.ascontiguousarray(offset, dtype=dt)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=dt)
    weight, grad_out = np.ascontiguousarray(weight, dtype=dt), np.ascontiguousarray(grad_out, dtype=dt)
    B, C, H, W = im.shape
    Cout = weight.shape[0]
    K = kh * kw
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    g = _geom(B, C, H, W, kh, kw, pad, stride, dil, 1)
    assert lib().detops_deformable_nhwc_supported(_DT[dt], C, Cout, 1) == 1
    mp = None if mask is None else _p(mask)
    xT = nchw_to_nhwc(im)
    colT = np.full((B * Ho * Wo, K * C), np.nan, dt)
    rc = lib().detops_deformable_im2col_nhwc(_p(xT), _p(offset), mp, _p(colT), _DT[dt], *g, None)
    assert rc == 0, rc
    f = np.float32
    W2 = weight.transpose(0, 2, 3, 1).reshape(Cout, K * C).astype(f)
    out = np.einsum("ok,bpk->bop", W2, colT.astype(f).reshape(B, Ho * Wo, K * C)).reshape(B, Cout, Ho, Wo)
    gT = nchw_to_nhwc(grad_out)                                              # [B, Ho*Wo, Cout]
    g2 = gT.reshape(-1, Cout).astype(f)
    colsG = np.ascontiguousarray((g2 @ W2).astype(dt))
    goff = np.full(offset.shape, np.nan, dt)
    gmask = None if mask is None else np.full(mask.shape, np.nan, dt)
    rc = lib().detops_deformable_coord_nhwc(_p(colsG), _p(xT), _p(offset), mp, _p(goff), None if gmask is None else _p(gmask),
                                            _DT[dt], *g, None)
    assert rc == 0, rc
    nbytes = lib().detops_deformable_transposed_sample_workspace_bytes(*g)
    assert nbytes > 0
    ws = np.full((nbytes,), 0xAB, np.uint8)
    S_T = np.full((B * H * W, K * Cout), np.nan, dt)
    rc = lib().detops_deformable_transposed_sample(_p(gT), _p(offset), mp, _p(S_T), _DT[dt], B, C, H, W, Cout, *g[4:],
                                                   _p(ws), nbytes, None)
    assert rc == 0, rc
    W2T = weight.transpose(1, 2, 3, 0).reshape(C, K * Cout).astype(f)
    gin = np.einsum("ck,bpk->bcp", W2T, S_T.astype(f).reshape(B, H * W, K * Cout)).reshape(B, C, H, W)
    gw = (g2.T @ colT.astype(f)).reshape(Cout, kh, kw, C).transpose(0, 3, 1, 2)
    if input_grad == "col2im":      # round 6: the input gradient as a col2im GATHER of the column gradient (what _C.py runs by default)
        ws = np.full((nbytes,), 0xCD, np.uint8)
        ginT = np.full((B, H * W, C), np.nan, dt)
        rc = lib().detops_deformable_col2im_nhwc(_p(colsG), _p(offset), mp, _p(ginT), _DT[dt], *g, _p(ws), nbytes, None)
        assert rc == 0, rc
        gin = ginT.astype(f).transpose(0, 2, 1).reshape(B, C, H, W)
    return out, gin, goff, gmask, gw


def deformable_col2im_coord(col, im, offset, mask, kh, kw, pad, stride, dil, dg):
    col = np.ascontiguousarray(col)
    dt = col.dtype
    im, offset = np.ascontiguousarray(im, dtype=dt), np.ascontiguousarray(offset, dtype=dt)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=dt)
    B, C, H, W = im.shape
    goff = np.full(offset.shape, np.nan, dt)
    gmask = None if mask is None else np.full(mask.shape, np.nan, dt)
    rc = lib().detops_deformable_col2im_coord(_p(col), _p(im), _p(offset), None if mask is None else _p(mask), _p(goff),
                                              None if gmask is None else _p(gmask), _DT[dt],
                                              *_geom(B, C, H, W, kh, kw, pad, stride, dil, dg), None)
    assert rc == 0, rc
    return goff, gmask


# ---------------------------------------------------------------------------------- NMS
def nms(boxes, scores, thr):
    boxes, scores = _f32(boxes), _f32(scores)
    n = boxes.shape[0]
    keep = np.full((max(n, 1),), -1, np.int64)
    num = np.full((1,), -1, np.int32)
    nbytes = lib().detops_nms_workspace_bytes(n)
    ws = np.empty((nbytes,), np.uint8)
    rc = lib().detops_nms_f32(_p(boxes), _p(scores), n, thr, _p(keep), _p(num), _p(ws), nbytes, None)
    assert rc == 0, rc
    return keep[:num[0]].copy()


def nms_batched(boxes, scores, offsets, max_n, thr, mask=False, status=None):
    """status: an int32 [1] array -> the call goes through detops_nms_batched_status_f32 (the entry point the product's
    wrappers use) and the array is incremented once per segment the repair launch redid"""
    boxes, scores = _f32(boxes), _f32(scores)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    S = offsets.shape[0] - 1
    num = np.full((S,), -7, np.int32)
    nbytes = lib().detops_nms_batched_workspace_bytes(S, max_n)
    ws = np.full((nbytes,), 0xA5, np.uint8)       # the library must not rely on any workspace content
    if status is not None:
        km = np.full((boxes.shape[0],), 7, np.uint8) if mask else None
        keep = None if mask else np.full((boxes.shape[0],), -1, np.int64)
        fn = lib().detops_nms_batched_status_f32
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 5 + \
                      [ctypes.c_size_t, ctypes.c_void_p]
        rc = fn(_p(boxes), _p(scores), _p(offsets), S, max_n, thr, None if mask else _p(keep), _p(km) if mask else None,
                _p(num), _p(status), _p(ws), nbytes, None)
        assert rc == 0, rc
        return (km if mask else keep), num
    if mask:
        km = np.full((boxes.shape[0],), 7, np.uint8)
        rc = lib().detops_nms_batched_mask_f32(_p(boxes), _p(scores), _p(offsets), S, max_n, thr, _p(km), _p(num),
                                               _p(ws), nbytes, None)
        assert rc == 0, rc
        return km, num
    keep = np.full((boxes.shape[0],), -1, np.int64)
    rc = lib().detops_nms_batched_f32(_p(boxes), _p(scores), _p(offsets), S, max_n, thr, _p(keep), _p(num), _p(ws),
                                      nbytes, None)
    assert rc == 0, rc
    return keep, num


# ---------------------------------------------------------------------------------- focal loss
def focal_forward(logits, targets, gamma, alpha, with_sum=False):
    logits = _f32(logits)
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    R, C = logits.shape
    out = np.full((R, C), np.nan, np.float32)
    if with_sum:
        tot = np.zeros((1,), np.float32)
        rc = lib().detops_sigmoid_focal_loss_forward_sum_f32(_p(logits), _p(targets), _p(out), _p(tot), R, C, gamma,
                                                             alpha, None)
        assert rc == 0, rc
        part = np.zeros((5,), np.float32)
        rc = lib().detops_sigmoid_focal_loss_forward_partial_sums_f32(_p(logits), _p(targets), None, _p(part), 5, R, C,
                                                                      gamma, alpha, None)
        assert rc == 0, rc
        assert abs(float(part.sum()) - float(tot[0])) <= 1e-5 * max(1.0, abs(float(tot[0])))
        # two-stage form: poisoned workspace and result word, must be overwritten; bit-reproducible
        nbytes = lib().detops_sigmoid_focal_loss_sum_workspace_bytes()
        two = []
        for fill in (0xAB, 0x7F):
            ws = np.full((nbytes,), fill, np.uint8)
            t2 = np.full((1,), np.nan, np.float32)
            rc = lib().detops_sigmoid_focal_loss_forward_sum_ws_f32(_p(logits), _p(targets), None, _p(t2), R, C, gamma,
                                                                    alpha, _p(ws), nbytes, None)
            assert rc == 0, rc
            two.append(t2[0])
        assert two[0] == two[1]
        assert abs(float(two[0]) - float(tot[0])) <= 1e-5 * max(1.0, abs(float(tot[0])))
        return out, float(two[0])
    rc = lib().detops_sigmoid_focal_loss_forward_f32(_p(logits), _p(targets), _p(out), R, C, gamma, alpha, None)
    assert rc == 0, rc
    return out


def focal_backward(logits, targets, d, gamma, alpha):
    logits = _f32(logits)
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    R, C = logits.shape
    out = np.full((R, C), np.nan, np.float32)
    if np.ndim(d) == 0:
        ds = np.full((1,), d, np.float32)
        rc = lib().detops_sigmoid_focal_loss_backward_scalar_f32(_p(logits), _p(targets), _p(ds), _p(out), R, C, gamma,
                                                                 alpha, None)
    else:
        d = _f32(d)
        rc = lib().detops_sigmoid_focal_loss_backward_f32(_p(logits), _p(targets), _p(d), _p(out), R, C, gamma, alpha,
                                                          None)
    assert rc == 0, rc
    return out


# ---------------------------------------------------------------------------------- fused FrozenBN
def frozen_bn_forward(x, scale, bias, residual, relu):
    x = np.ascontiguousarray(x)
    dt = x.dtype
    N, C = x.shape[:2]
    HW = int(np.prod(x.shape[2:]))
    y = np.full(x.shape, np.nan, dt)
    residual = None if residual is None else np.ascontiguousarray(residual, dtype=dt)
    rc = lib().detops_frozen_bn_act_forward(_p(x), _p(_f32(scale)), _p(_f32(bias)),
                                            None if residual is None else _p(residual), _p(y), _DT[dt], N, C, HW,
                                            int(relu), None)
    assert rc == 0, rc
    return y


def frozen_bn_backward(gy, y, scale, relu, want_residual):
    gy = np.ascontiguousarray(gy)
    dt = gy.dtype
    y = np.ascontiguousarray(y, dtype=dt)
    N, C = gy.shape[:2]
    HW = int(np.prod(gy.shape[2:]))
    gx = np.full(gy.shape, np.nan, dt)
    gr = np.full(gy.shape, np.nan, dt) if want_residual else None
    rc = lib().detops_frozen_bn_act_backward(_p(gy), _p(y), _p(_f32(scale)), _p(gx), None if gr is None else _p(gr),
                                             _DT[dt], N, C, HW, int(relu), None)
    assert rc == 0, rc
    return gx, gr


# ---------------------------------------------------------------------------------- deformable PS-ROI pooling
def psroi_forward(data, rois, trans, no_trans, scale, output_dim, group_size, pooled, part, spp, trans_std):
    data, rois = _f32(data), _f32(rois)
    trans = None if trans is None else _f32(trans)
    N, C, H, W = data.shape
    K = rois.shape[0]
    ct = 0 if trans is None else trans.shape[1]
    out = np.full((K, output_dim, pooled, pooled), np.nan, np.float32)
    cnt = np.full((K, output_dim, pooled, pooled), np.nan, np.float32)
    rc = lib().detops_deform_psroi_pool_forward_f32(_p(data), _p(rois), None if trans is None else _p(trans), _p(out),
                                                    _p(cnt), N, C, H, W, K, ct, int(no_trans), scale, output_dim,
                                                    group_size, pooled, part, spp, trans_std, None)
    assert rc == 0, rc
    return out, cnt


def roi_pool_forward(data, rois, scale, PH, PW):
    data, rois = _f32(data), _f32(rois)
    N, C, H, W = data.shape
    K = rois.shape[0]
    out = np.full((K, C, PH, PW), np.nan, np.float32)
    amax = np.full((K, C, PH, PW), -7, np.int32)
    rc = lib().detops_roi_pool_forward_f32(_p(data), _p(rois), _p(out), _p(amax), N, C, H, W, K, PH, PW, scale, None)
    assert rc == 0, rc
    return out, amax


def roi_pool_backward(grad, rois, argmax, N, C, H, W, into=None):
    """into: accumulate onto this array (zero_grad_in = 0) instead of starting from zeros"""
    grad, rois = _f32(grad), _f32(rois)
    argmax = np.ascontiguousarray(argmax, dtype=np.int32)
    K, _, PH, PW = grad.shape
    gin = np.full((N, C, H, W), np.nan, np.float32) if into is None else np.ascontiguousarray(into, dtype=np.float32).copy()
    rc = lib().detops_roi_pool_backward_f32(_p(grad), _p(rois), _p(argmax), _p(gin), N, C, H, W, K, PH, PW,
                                            1 if into is None else 0, None)
    assert rc == 0, rc
    return gin


def psroi_backward(grad, data, rois, trans, cnt, no_trans, scale, output_dim, group_size, pooled, part, spp, trans_std):
    grad, data, rois, cnt = _f32(grad), _f32(data), _f32(rois), _f32(cnt)
    trans = None if trans is None else _f32(trans)
    N, C, H, W = data.shape
    K = rois.shape[0]
    ct = 0 if trans is None else trans.shape[1]
    dg = np.full(data.shape, np.nan, np.float32)
    tg = None if trans is None else np.full(trans.shape, np.nan, np.float32)
    rc = lib().detops_deform_psroi_pool_backward_f32(_p(grad), _p(data), _p(rois), None if trans is None else _p(trans),
                                                     _p(cnt), _p(dg), None if tg is None else _p(tg), N, C, H, W, K, ct,
                                                     int(no_trans), scale, output_dim, group_size, pooled, part, spp,
                                                     trans_std, 1, None)
    assert rc == 0, rc
    return dg, tg


def stats(reset=True):
    """work counters accumulated by DETOPS_STAT in the emulated kernels -> {name: value}"""
    buf = ctypes.create_string_buffer(1 << 16)
    lib().detops_emu_stats.restype = c_int
    lib().detops_emu_stats.argtypes = [ctypes.c_char_p, c_int, c_int]
    lib().detops_emu_stats(buf, len(buf), int(reset))
    out = {}
    for line in buf.value.decode().splitlines():
        k, v = line.split("=")
        out[k] = float(v)
    return out


def rpn_loss(objectness, box_regression, anchors, matched, pos, neg, gt, beta, weights, upstream=(1.0, 1.0)):
    """-> (objectness loss, box loss, grads wrt the objectness levels, grads wrt the box-regression levels) with the
    backward scaling applied for the given upstream gradients."""
    obj = [_f32(t) for t in objectness]
    box = [_f32(t) for t in box_regression]
    anchors, gt = _f32(anchors), _f32(gt)
    matched = np.ascontiguousarray(matched, dtype=np.int64)
    pos = np.ascontiguousarray(pos, dtype=np.uint8)
    neg = np.ascontiguousarray(neg, dtype=np.uint8)
    L, N, A = len(obj), obj[0].shape[0], obj[0].shape[1]
    T, M = anchors.shape[0], gt.shape[1]
    gobj = [np.full(t.shape, np.nan, np.float32) for t in obj]
    gbox = [np.full(t.shape, np.nan, np.float32) for t in box]
    arr = lambda ts: (ctypes.c_void_p * L)(*[t.ctypes.data for t in ts])
    Hs = (ctypes.c_int * L)(*[t.shape[2] for t in obj])
    Ws = (ctypes.c_int * L)(*[t.shape[3] for t in obj])
    w4 = (ctypes.c_float * 4)(*[float(w) for w in weights])
    out3 = np.full((3,), np.nan, np.float32)
    nbytes = lib().detops_rpn_loss_workspace_bytes()
    ws = np.full((nbytes,), 0xAB, np.uint8)
    rc = lib().detops_rpn_loss_f32(arr(obj), arr(box), Hs, Ws, L, A, _p(anchors), _p(matched), _p(pos), _p(neg), _p(gt),
                                   N, M, T, beta, w4, arr(gobj), arr(gbox), _p(out3), _p(ws), nbytes, None)
    assert rc == 0, rc
    uo, ub = np.full((1,), upstream[0], np.float32), np.full((1,), upstream[1], np.float32)
    rc = lib().detops_rpn_loss_backward_f32(arr(gobj), arr(gbox), Hs, Ws, L, A, N, T, _p(uo), _p(ub), _p(out3[2:]), None)
    assert rc == 0, rc
    return float(out3[0]), float(out3[1]), gobj, gbox


# ---------------------------------------------------------------------------------- data-parallel bucket kernels (optim.hip)
def pack(dst, arrays, offsets):
    """dst[offsets[i] : offsets[i] + arrays[i].size] = arrays[i] through detops_pack_f32 (<= detops_pack_max_tensors per call)"""
    n = len(arrays)
    arrays = [_f32(a).reshape(-1) for a in arrays]
    srcs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrays])
    cnts = (ctypes.c_int64 * n)(*[a.size for a in arrays])
    offs = (ctypes.c_int64 * n)(*[int(o) for o in offsets])
    rc = lib().detops_pack_f32(srcs, cnts, offs, n, _p(dst), None)
    return rc


def sgd_momentum_flat(p, g, m, split, lr_w, wd_w, lr_b, wd_b, momentum):
    fn = lib().detops_sgd_momentum_flat_f32
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int64] + [ctypes.c_float] * 5 + [ctypes.c_void_p]
    return fn(_p(p), _p(g), _p(m), p.size, int(split), lr_w, wd_w, lr_b, wd_b, momentum, None)


# ---------------------------------------------------------------------------------- FPN top-down step (fpn_topdown.hip)
def fpn_topdown_forward(lateral, top):
    lateral, top = _f32(lateral), _f32(top)
    N, C, H, W = lateral.shape
    out = np.full_like(lateral, np.nan)
    rc = lib().detops_fpn_topdown_forward(_p(lateral), _p(top), _p(out), 0, N * C, H, W, top.shape[2], top.shape[3], None)
    assert rc == 0, rc
    return out


def fpn_topdown_backward(grad_out, h, w):
    grad_out = _f32(grad_out)
    N, C, H, W = grad_out.shape
    gtop = np.full((N, C, h, w), np.nan, np.float32)
    rc = lib().detops_fpn_topdown_backward(_p(grad_out), _p(gtop), 0, N * C, H, W, h, w, None)
    assert rc == 0, rc
    return gtop
