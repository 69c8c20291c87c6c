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
    "detops_roi_align_forward_f32": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, _P]),
    "detops_roi_align_backward_f32": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, c_int, _P]),
    "detops_roi_align_fpn_forward_f32": (
        c_int, [_P, _P, _P, _P, c_int, _P, _P, _P] + [c_int] * 8 + [c_float, c_float, c_float, _P]),
    "detops_roi_align_fpn_backward_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "detops_roi_pool_forward_f32": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [c_float, _P]),
    "detops_roi_pool_backward_f32": (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "detops_deformable_im2col": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P]),
    "detops_deformable_col2im": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P]),
    "detops_deformable_col2im_workspace_bytes": (ctypes.c_size_t, [c_int] * 13),
    "detops_deformable_col2im_ws": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P, ctypes.c_size_t, _P]),
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


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_P)


def roi_align_forward(inp, rois, scale, ph, pw, sr):
    inp, rois = _f32(inp), _f32(rois)
    N, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.full((K, C, ph, pw), np.nan, np.float32)
    rc = lib().detops_roi_align_forward_f32(_p(inp), _p(rois), _p(out), N, C, H, W, K, ph, pw, scale, sr, None)
    assert rc == 0, rc
    return out


def roi_align_backward(grad, rois, scale, ph, pw, N, C, H, W, sr, into=None):
    """into=None: zero_grad_in=1 on a NaN-filled buffer (every element must be written);
    into=array: accumulate (zero_grad_in=0)."""
    grad, rois = _f32(grad), _f32(rois)
    K = rois.shape[0]
    gin = np.full((N, C, H, W), np.nan, np.float32) if into is None else _f32(into).copy()
    rc = lib().detops_roi_align_backward_f32(_p(grad), _p(rois), _p(gin), N, C, H, W, K, ph, pw, scale, sr,
                                             1 if into is None else 0, None)
    assert rc == 0, rc
    return gin


def _host_arrays(arrs, scales):
    L = len(arrs)
    ptrs = (ctypes.c_void_p * L)(*[a.ctypes.data for a in arrs])
    Hs = (ctypes.c_int * L)(*[a.shape[2] for a in arrs])
    Ws = (ctypes.c_int * L)(*[a.shape[3] for a in arrs])
    sc = (ctypes.c_float * L)(*[float(s) for s in scales])
    return ptrs, Hs, Ws, sc


def roi_align_fpn_forward(feats, rois, scales, ph, pw, sr, k_min, k_max):
    feats = [_f32(f) for f in feats]
    rois = _f32(rois)
    N, C = feats[0].shape[:2]
    K = rois.shape[0]
    out = np.full((K, C, ph, pw), np.nan, np.float32)
    levels = np.full((K,), -1, np.int32)
    ptrs, Hs, Ws, sc = _host_arrays(feats, scales)
    rc = lib().detops_roi_align_fpn_forward_f32(ptrs, Hs, Ws, sc, len(feats), _p(rois), _p(out), _p(levels), N, C, K,
                                                ph, pw, sr, k_min, k_max, 224.0, 4.0, 1e-6, None)
    assert rc == 0, rc
    return out, levels


def roi_align_fpn_backward(grad, rois, levels, shapes, scales, ph, pw, sr):
    grad, rois = _f32(grad), _f32(rois)
    levels = np.ascontiguousarray(levels, dtype=np.int32)
    gins = [np.full(s, np.nan, np.float32) for s in shapes]
    N, C = shapes[0][:2]
    ptrs, Hs, Ws, sc = _host_arrays(gins, scales)
    rc = lib().detops_roi_align_fpn_backward_f32(_p(grad), _p(rois), _p(levels), ptrs, Hs, Ws, sc, len(shapes), N, C,
                                                 rois.shape[0], ph, pw, sr, 1, None)
    assert rc == 0, rc
    return gins


# ---------------------------------------------------------------------------------- deformable conv
_DT = {np.dtype(np.float32): 0, np.dtype(np.float16): 1}


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


def deformable_col2im(col, offset, mask, B, C, H, W, kh, kw, pad, stride, dil, dg, gather=True, into=None):
    col = np.ascontiguousarray(col)
    dt = col.dtype
    offset = np.ascontiguousarray(offset, dtype=dt)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=dt)
    gim = np.zeros((B, C, H, W), dt) if into is None else np.ascontiguousarray(into, dtype=dt).copy()
    g = _geom(B, C, H, W, kh, kw, pad, stride, dil, dg)
    mp = None if mask is None else _p(mask)
    if gather:
        nbytes = lib().detops_deformable_col2im_workspace_bytes(*g)
        assert nbytes > 0
        ws = np.full((nbytes,), 0xAB, np.uint8)  # arbitrary contents
        rc = lib().detops_deformable_col2im_ws(_p(col), _p(offset), mp, _p(gim), _DT[dt], *g, _p(ws), nbytes, None)
    else:
        rc = lib().detops_deformable_col2im(_p(col), _p(offset), mp, _p(gim), _DT[dt], *g, None)
    assert rc == 0, rc
    return gim


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
