"""CPU oracle for the detection-head hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package,
and only as the checker (never as the thing measured or shipped).  The product path
(maskrcnn-benchmark_amd/) must not import it; it raises when its HIP library is missing.

Two layers:
  * `oracle.*`  — numpy front-end over oracle/libdetops_oracle.so, the plain-C restatement in
    oracle/detops_oracle.c (each C function cites the reference file:line it follows).
  * `oracle.ref()` — the reference's OWN CPU kernels (`nms`, `roi_align_forward`) compiled in
    place from /root/reference by oracle/build_ref.py into oracle/_ref/ (None if never built).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdetops_oracle.so")
_lib = None


def build():
    """Compile the C restatement (gcc, ~1 s)."""
    src = os.path.join(_HERE, "detops_oracle.c")
    if (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def ref():
    """The reference's own CPU kernels (torch extension module) or None."""
    from . import build_ref

    return build_ref.load_ref()


_f = ctypes.c_float
_i = ctypes.c_int
_P = ctypes.c_void_p


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def roi_align_forward(inp, rois, spatial_scale, ph, pw, sampling_ratio):
    inp, rois = _f32(inp), _f32(rois).reshape(-1, 5)
    N, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.empty((K, C, ph, pw), np.float32)
    lib().oracle_roi_align_forward_f32(_p(inp), _p(rois), _p(out), _i(N), _i(C), _i(H), _i(W),
                                       _i(K), _i(ph), _i(pw), _f(spatial_scale),
                                       _i(sampling_ratio))
    return out


def roi_align_backward(grad, rois, spatial_scale, ph, pw, N, C, H, W, sampling_ratio,
                       acc64=False):
    grad, rois = _f32(grad), _f32(rois).reshape(-1, 5)
    K = rois.shape[0]
    assert grad.shape == (K, C, ph, pw)
    gin = np.empty((N, C, H, W), np.float32)
    lib().oracle_roi_align_backward_f32(_p(grad), _p(rois), _p(gin), _i(N), _i(C), _i(H), _i(W),
                                        _i(K), _i(ph), _i(pw), _f(spatial_scale),
                                        _i(sampling_ratio), _i(1 if acc64 else 0))
    return gin


def roi_pool_forward(inp, rois, spatial_scale, ph, pw):
    inp, rois = _f32(inp), _f32(rois).reshape(-1, 5)
    N, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.empty((K, C, ph, pw), np.float32)
    argmax = np.empty((K, C, ph, pw), np.int32)
    lib().oracle_roi_pool_forward_f32(_p(inp), _p(rois), _p(out), _p(argmax), _i(N), _i(C),
                                      _i(H), _i(W), _i(K), _i(ph), _i(pw), _f(spatial_scale))
    return out, argmax


def roi_pool_backward(grad, rois, argmax, N, C, H, W):
    grad, rois, argmax = _f32(grad), _f32(rois).reshape(-1, 5), _i32(argmax)
    K, _, ph, pw = grad.shape
    gin = np.empty((N, C, H, W), np.float32)
    lib().oracle_roi_pool_backward_f32(_p(grad), _p(rois), _p(argmax), _p(gin), _i(N), _i(C),
                                       _i(H), _i(W), _i(K), _i(ph), _i(pw))
    return gin


def nms(boxes, scores, threshold):
    boxes, scores = _f32(boxes).reshape(-1, 4), _f32(scores).reshape(-1)
    n = boxes.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    fn = lib().oracle_nms_f32
    fn.restype = ctypes.c_int
    k = fn(_p(boxes), _p(scores), _i(n), _f(threshold), _p(keep))
    return keep[:k].copy()


def sigmoid_focal_loss_forward(logits, targets, gamma, alpha):
    logits, targets = _f32(logits), _i32(targets)
    R, C = logits.shape
    out = np.empty((R, C), np.float32)
    lib().oracle_sigmoid_focal_loss_forward_f32(_p(logits), _p(targets), _p(out), _i(R), _i(C),
                                                _f(gamma), _f(alpha))
    return out


def sigmoid_focal_loss_backward(logits, targets, d_losses, gamma, alpha):
    logits, targets, d_losses = _f32(logits), _i32(targets), _f32(d_losses)
    R, C = logits.shape
    out = np.empty((R, C), np.float32)
    lib().oracle_sigmoid_focal_loss_backward_f32(_p(logits), _p(targets), _p(d_losses), _p(out),
                                                 _i(R), _i(C), _f(gamma), _f(alpha))
    return out


def _conv_out(H, W, kh, kw, pad, stride, dil):
    Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    return Ho, Wo


def _geom(kh, kw, pad, stride, dil, dg):
    return [_i(kh), _i(kw), _i(pad[0]), _i(pad[1]), _i(stride[0]), _i(stride[1]), _i(dil[0]),
            _i(dil[1]), _i(dg)]


def deformable_im2col(im, offset, mask, kh, kw, pad, stride, dil, dg):
    im, offset = _f32(im), _f32(offset)
    mask = None if mask is None else _f32(mask)
    B, C, H, W = im.shape
    Ho, Wo = _conv_out(H, W, kh, kw, pad, stride, dil)
    col = np.empty((C * kh * kw, B * Ho * Wo), np.float32)
    lib().oracle_deformable_im2col_f32(_p(im), _p(offset), _p(mask), _p(col), _i(B), _i(C),
                                       _i(H), _i(W), *_geom(kh, kw, pad, stride, dil, dg))
    return col


def deformable_col2im(col, offset, mask, B, C, H, W, kh, kw, pad, stride, dil, dg):
    col, offset = _f32(col), _f32(offset)
    mask = None if mask is None else _f32(mask)
    gim = np.zeros((B, C, H, W), np.float32)
    lib().oracle_deformable_col2im_f32(_p(col), _p(offset), _p(mask), _p(gim), _i(B), _i(C),
                                       _i(H), _i(W), *_geom(kh, kw, pad, stride, dil, dg))
    return gim


def deformable_col2im_coord(col, im, offset, mask, kh, kw, pad, stride, dil, dg):
    col, im, offset = _f32(col), _f32(im), _f32(offset)
    mask = None if mask is None else _f32(mask)
    B, C, H, W = im.shape
    goff = np.zeros_like(offset)
    gmask = None if mask is None else np.zeros_like(mask)
    lib().oracle_deformable_col2im_coord_f32(_p(col), _p(im), _p(offset), _p(mask), _p(goff),
                                             _p(gmask), _i(B), _i(C), _i(H), _i(W),
                                             *_geom(kh, kw, pad, stride, dil, dg))
    return goff, gmask


def deform_conv_forward(inp, offset, mask, weight, bias, pad, stride, dil, group, dg):
    inp, offset, weight = _f32(inp), _f32(offset), _f32(weight)
    mask = None if mask is None else _f32(mask)
    bias = None if bias is None else _f32(bias)
    B, C, H, W = inp.shape
    Cout, _, kh, kw = weight.shape
    Ho, Wo = _conv_out(H, W, kh, kw, pad, stride, dil)
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    lib().oracle_deform_conv_forward_f32(_p(inp), _p(offset), _p(mask), _p(weight), _p(bias),
                                         _p(out), _i(B), _i(C), _i(H), _i(W), _i(Cout), _i(kh),
                                         _i(kw), _i(pad[0]), _i(pad[1]), _i(stride[0]),
                                         _i(stride[1]), _i(dil[0]), _i(dil[1]), _i(group), _i(dg))
    return out


def deform_conv_backward(inp, offset, mask, weight, grad_out, with_bias, pad, stride, dil, group,
                         dg):
    inp, offset, weight, grad_out = _f32(inp), _f32(offset), _f32(weight), _f32(grad_out)
    mask = None if mask is None else _f32(mask)
    B, C, H, W = inp.shape
    Cout, _, kh, kw = weight.shape
    gin = np.empty_like(inp)
    goff = np.empty_like(offset)
    gmask = None if mask is None else np.empty_like(mask)
    gw = np.empty_like(weight)
    gb = np.empty((Cout,), np.float32) if with_bias else None
    lib().oracle_deform_conv_backward_f32(_p(inp), _p(offset), _p(mask), _p(weight),
                                          _p(grad_out), _p(gin), _p(goff), _p(gmask), _p(gw),
                                          _p(gb), _i(B), _i(C), _i(H), _i(W), _i(Cout), _i(kh),
                                          _i(kw), _i(pad[0]), _i(pad[1]), _i(stride[0]),
                                          _i(stride[1]), _i(dil[0]), _i(dil[1]), _i(group),
                                          _i(dg))
    return gin, goff, gmask, gw, gb


def fpn_level(rois, k_min, k_max, canonical_scale=224.0, canonical_level=4.0, eps=1e-6):
    rois = _f32(rois).reshape(-1, 5)
    K = rois.shape[0]
    lv = np.empty((K,), np.int32)
    lib().oracle_fpn_level_f32(_p(rois), _i(K), _i(k_min), _i(k_max), _f(canonical_scale),
                               _f(canonical_level), _f(eps), _p(lv))
    return lv


def deform_psroi_pool_forward(data, rois, trans, no_trans, spatial_scale, output_dim, group_size,
                              pooled_size, part_size, sample_per_part, trans_std):
    """-> (out, top_count), both [K, output_dim, P, P] (deform_pool_kernel_cuda.cu:53-147)."""
    data, rois = _f32(data), _f32(rois).reshape(-1, 5)
    trans = None if no_trans else _f32(trans)
    N, C, H, W = data.shape
    K = rois.shape[0]
    ct = 2 if no_trans else trans.shape[1]
    out = np.empty((K, output_dim, pooled_size, pooled_size), np.float32)
    cnt = np.empty_like(out)
    lib().oracle_deform_psroi_pool_forward_f32(
        _p(data), _p(rois), _p(trans), _p(out), _p(cnt), _i(N), _i(C), _i(H), _i(W), _i(K), _i(ct),
        _i(int(bool(no_trans))), _f(spatial_scale), _i(output_dim), _i(group_size), _i(pooled_size),
        _i(part_size), _i(sample_per_part), _f(trans_std))
    return out, cnt


def deform_psroi_pool_backward(grad, data, rois, trans, top_count, no_trans, spatial_scale,
                               output_dim, group_size, pooled_size, part_size, sample_per_part,
                               trans_std, acc64=False):
    """-> (data_grad [N,C,H,W], trans_grad like trans or None) (deform_pool_kernel_cuda.cu:149-264)."""
    grad, data, rois, top_count = _f32(grad), _f32(data), _f32(rois).reshape(-1, 5), _f32(top_count)
    trans = None if no_trans else _f32(trans)
    N, C, H, W = data.shape
    K = rois.shape[0]
    ct = 2 if no_trans else trans.shape[1]
    gin = np.empty_like(data)
    gtr = None if no_trans else np.empty_like(trans)
    lib().oracle_deform_psroi_pool_backward_f32(
        _p(grad), _p(data), _p(rois), _p(trans), _p(top_count), _p(gin), _p(gtr), _i(N), _i(C),
        _i(H), _i(W), _i(K), _i(ct), _i(int(bool(no_trans))), _f(spatial_scale), _i(output_dim),
        _i(group_size), _i(pooled_size), _i(part_size), _i(sample_per_part), _f(trans_std),
        _i(int(acc64)))
    return gin, gtr
