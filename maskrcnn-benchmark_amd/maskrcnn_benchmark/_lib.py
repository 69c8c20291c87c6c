"""ctypes binding of libdetops_gfx950.so (C ABI declared in include/detops.h).

The library is built in-tree by `make -C maskrcnn-benchmark_amd/csrc` (hipcc, gfx950) into
maskrcnn_benchmark/lib/.  There is NO fallback: if the shared object is missing or does not export
a symbol the import fails loudly — the detection-head operators exist only as HIP kernels.
"""
import ctypes
import os

import torch  # noqa: F401  — loads the HIP runtime (libamdhip64) this library links against

_HERE = os.path.dirname(os.path.abspath(__file__))
# DETOPS_LIB_PATH: another build of the SAME library (same ABI version, checked below) — same-box A/B measurements of a
# kernel change (tools/gpu/ab_build.sh puts the previous commit's build next to the current one)
LIB_PATH = os.environ.get("DETOPS_LIB_PATH") or os.path.join(_HERE, "lib", "libdetops_gfx950.so")

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/detops.h one to one
_P = c_void_p
SIGNATURES = {
    "detops_version": (c_int, [_P]),
    "detops_tuning_set": (c_int, [ctypes.c_char_p, c_int]),
    "detops_tuning_get": (c_int, [ctypes.c_char_p, _P]),
    "detops_roi_align_forward_f32": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, _P]),
    "detops_roi_align_backward_f32": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, c_int, _P]),
    "detops_roi_align_fpn_forward_f32": (
        c_int, [_P, _P, _P, _P, c_int, _P, _P, _P] + [c_int] * 8 + [c_float, c_float, c_float, _P]),
    "detops_roi_align_fpn_backward_f32": (
        c_int, [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "detops_roi_align_fpn_backward_ws_f32": (
        c_int, [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 8 + [_P, c_size_t, _P]),
    "detops_roi_align_fpn_backward_prepare_f32": (c_int, [_P] * 5 + [c_int] * 7 + [_P, c_size_t, _P]),
    "detops_roi_align_fpn_backward_prepared_f32": (c_int, [_P] * 5 + [c_int] * 7 + [_P, c_size_t, _P]),
    "detops_roi_align_backward_ws_f32": (
        c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, c_int, _P, c_size_t, _P]),
    "detops_roi_align_backward_workspace_bytes": (c_size_t, [_P, _P] + [c_int] * 6),
    "detops_roi_align_forward_workspace_bytes": (c_size_t, [c_int] * 4),
    "detops_nms_cpu_f32": (c_int, [_P, _P, c_int, c_float, _P, _P]),
    "detops_rpn_loss_workspace_bytes": (c_size_t, []),
    "detops_rpn_loss_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float,
                                    _P, _P, _P, _P, _P, c_size_t, _P]),
    "detops_rpn_loss_backward_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "detops_roi_align_forward_cpu_f32": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int]),
    "detops_roi_align_forward_ws_f32": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, _P, c_size_t, _P]),
    "detops_roi_align_fpn_forward_ws_f32": (
        c_int, [_P, _P, _P, _P, c_int, _P, _P, _P] + [c_int] * 8 + [c_float, c_float, c_float, _P, c_size_t, _P]),
    "detops_match_boxes_workspace_bytes": (c_size_t, [c_int, c_int]),
    "detops_match_boxes_f32": (c_int, [_P, _P, _P] + [c_int] * 4 + [c_float, c_float, c_int, _P, _P, c_size_t, _P]),
    "detops_sample_labels_workspace_bytes": (c_size_t, [c_int, c_int]),
    "detops_sample_labels": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint64, _P, _P, _P, _P, _P, c_size_t, _P]),
    "detops_sample_labels_dseed": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint64, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "detops_mask_targets": (c_int, [_P, c_int, _P, _P] + [c_int] * 5 + [_P, _P]),
    "detops_match_labels": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "detops_roi_head_targets_f32": (c_int, [_P] * 8 + [c_int] * 4 + [c_float] * 4 + [_P] * 6),
    "detops_fastrcnn_loss_workspace_bytes": (c_size_t, [c_int]),
    "detops_fastrcnn_loss_f32": (c_int, [_P] * 4 + [c_int] * 4 + [c_float] + [_P] * 4 + [c_size_t, _P]),
    "detops_mask_loss_workspace_bytes": (c_size_t, [c_int]),
    "detops_mask_loss_f32": (c_int, [_P] * 3 + [c_int] * 3 + [_P] * 3 + [c_size_t, _P]),
    "detops_head_loss_backward_f32": (c_int, [_P, ctypes.c_int64, _P, _P, ctypes.c_int64, _P, _P]),
    "detops_roi_align_fpn_forward_nhwc_workspace_bytes": (c_size_t, [c_int]),
    "detops_roi_align_fpn_forward_nhwc_f32": (c_int, [_P, _P, _P, _P, c_int, _P, _P, c_int, _P] + [c_int] * 8 + [c_float] * 3 + [_P, c_size_t, _P]),
    "detops_roi_align_fpn_backward_ring_nhwc_f32": (c_int, [_P] * 7 + [c_int] * 8 + [_P, c_size_t, _P]),
    "detops_roi_align_fpn_backward_nhwc_workspace_bytes": (c_size_t, [_P, _P] + [c_int] * 6),
    "detops_roi_align_fpn_backward_nhwc_f32": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P] + [c_int] * 8 + [_P, c_size_t, _P]),
    "detops_fpn_topdown_forward": (c_int, [_P, _P, _P] + [c_int] * 6 + [_P]),
    "detops_fpn_topdown_backward": (c_int, [_P, _P] + [c_int] * 6 + [_P]),
    "detops_fpn_topdown_forward_nhwc": (c_int, [_P, _P, _P] + [c_int] * 7 + [_P]),
    "detops_fpn_topdown_backward_nhwc": (c_int, [_P, _P] + [c_int] * 7 + [_P]),
    "detops_pack_max_tensors": (c_int, []),
    "detops_pack_f32": (c_int, [_P, _P, _P, c_int, _P, _P]),
    "detops_bias_act_supported": (c_int, [c_int]),
    "detops_bias_act_backward_workspace_bytes": (c_size_t, [ctypes.c_int64, c_int]),
    "detops_bias_act_backward_nhwc_f32": (c_int, [_P, _P, _P, _P, ctypes.c_int64, c_int, c_int, _P, c_size_t, _P]),
    "detops_bias_act_backward_nhwc": (c_int, [_P, _P, _P, _P, c_int, ctypes.c_int64, c_int, c_int, _P, c_size_t, _P]),
    "detops_column_sum": (c_int, [_P, _P, c_int, ctypes.c_int64, c_int, _P, c_size_t, _P]),
    "detops_column_sum_workspace_bytes": (c_size_t, [ctypes.c_int64, c_int]),
    "detops_column_sum_f32": (c_int, [_P, _P, ctypes.c_int64, c_int, _P, c_size_t, _P]),
    "detops_roi_align_forward_f64": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, _P]),
    "detops_roi_align_backward_f64": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int, c_int, _P]),
    "detops_roi_pool_forward_f64": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [c_float, _P]),
    "detops_roi_pool_backward_f64": (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "detops_sigmoid_focal_loss_forward_f64": (c_int, [_P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "detops_sigmoid_focal_loss_backward_f64": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "detops_nms_sorted_f64_workspace_bytes": (c_size_t, [c_int]),
    "detops_nms_sorted_f64": (c_int, [_P, c_int, c_float, _P, _P, c_size_t, _P]),
    "detops_roi_align_forward_cpu_f64": (c_int, [_P, _P, _P] + [c_int] * 7 + [c_float, c_int]),
    "detops_nms_cpu_f64": (c_int, [_P, _P, c_int, c_float, _P, _P]),
    "detops_debug_occupy": (c_int, [c_int, c_int, _P]),
    "detops_debug_nms_timeline": (c_int, [_P, c_int]),
    "detops_sgd_momentum_flat_f32": (c_int, [_P, _P, _P, ctypes.c_int64, ctypes.c_int64] + [c_float] * 5 + [_P]),
    "detops_rpn_decode_f32": (c_int, [_P] * 5 + [c_int] * 5 + [c_float] * 6 + [_P, ctypes.c_int64, _P, ctypes.c_int64, _P, _P, _P, _P]),
    "detops_roi_pool_forward_f32": (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [c_float, _P]),
    "detops_roi_pool_backward_f32": (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "detops_nms_workspace_bytes": (c_size_t, [c_int]),
    "detops_nms_f32": (c_int, [_P, _P, c_int, c_float, _P, _P, _P, c_size_t, _P]),
    "detops_nms_batched_workspace_bytes": (c_size_t, [c_int, c_int]),
    "detops_nms_batched_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P, _P, c_size_t, _P]),
    "detops_nms_batched_mask_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P, _P, c_size_t, _P]),
    "detops_nms_batched_status_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "detops_sigmoid_focal_loss_forward_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "detops_sigmoid_focal_loss_backward_f32": (
        c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "detops_sigmoid_focal_loss_backward_scalar_f32": (
        c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "detops_sigmoid_focal_loss_forward_sum_f32": (
        c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "detops_sigmoid_focal_loss_forward_partial_sums_f32": (
        c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P]),
    "detops_sigmoid_focal_loss_sum_workspace_bytes": (c_size_t, []),
    "detops_sigmoid_focal_loss_forward_sum_ws_f32": (
        c_int, [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P, c_size_t, _P]),
    "detops_frozen_bn_act_forward": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [_P]),
    "detops_frozen_bn_act_backward": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 5 + [_P]),
    "detops_frozen_bn_act_forward_nhwc": (c_int, [_P, _P, _P, _P, _P, c_int, ctypes.c_int64, c_int, c_int, _P]),
    "detops_frozen_bn_act_backward_nhwc": (c_int, [_P, _P, _P, _P, _P, c_int, ctypes.c_int64, c_int, c_int, _P]),
    "detops_deformable_im2col": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P]),
    "detops_deformable_col2im": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P]),
    "detops_deformable_col2im_workspace_bytes": (c_size_t, [c_int] * 13),
    "detops_deformable_col2im_ws": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P, c_size_t, _P]),
    "detops_deform_conv_forward_fused_workspace_bytes": (c_size_t, [c_int] * 15),
    "detops_deform_conv_forward_fused": (c_int, [_P] * 6 + [c_int] * 15 + [_P, c_size_t, _P]),
    "detops_nchw_to_nhwc": (c_int, [_P, _P] + [c_int] * 4 + [_P]),
    "detops_deformable_nhwc_supported": (c_int, [c_int] * 4),
    "detops_deformable_im2col_nhwc": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P]),
    "detops_deformable_coord_nhwc": (c_int, [_P] * 6 + [c_int] * 14 + [_P]),
    "detops_deformable_transposed_sample_workspace_bytes": (c_size_t, [c_int] * 13),
    "detops_deformable_transposed_sample": (c_int, [_P, _P, _P, _P] + [c_int] * 15 + [_P, c_size_t, _P]),
    "detops_deformable_col2im_nhwc": (c_int, [_P, _P, _P, _P] + [c_int] * 14 + [_P, c_size_t, _P]),
    "detops_deformable_col2im_coord": (c_int, [_P, _P, _P, _P, _P, _P] + [c_int] * 14 + [_P]),
    "detops_deform_psroi_pool_forward_f32": (
        c_int, [_P] * 5 + [c_int] * 7 + [c_float] + [c_int] * 5 + [c_float, _P]),
    "detops_deform_psroi_pool_backward_f32": (
        c_int, [_P] * 7 + [c_int] * 7 + [c_float] + [c_int] * 5 + [c_float, c_int, _P]),
}

_ERRORS = {-1: "DETOPS_EINVAL (bad shape / null pointer)", -2: "DETOPS_EWORKSPACE (workspace too small)",
           -3: "DETOPS_EUNSUPPORTED (configuration not implemented)"}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libdetops_gfx950.so not found at %s — build it with `make -C maskrcnn-benchmark_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). The detection-head operators "
            "have no CPU / PyTorch fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.detops_version(None) != 1:
        raise ImportError("libdetops_gfx950.so: ABI version mismatch")
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        msg = _ERRORS.get(rc, "hipError_t %d" % rc)
        raise RuntimeError("%s failed: %s" % (what, msg))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t):
    """The current HIP stream of the tensor's device, as an integer handle (the raw-handle query: building a
    torch.cuda.Stream object per launch costs ~5 us of host time, ~200 launches per training step)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def tuning_set(key, value):
    """Tuning / test switch of the library (include/detops.h: detops_tuning_set)."""
    check(lib.detops_tuning_set(key.encode(), int(value)), "detops_tuning_set(%s)" % key)


def tuning_get(key):
    v = ctypes.c_int(0)
    check(lib.detops_tuning_get(key.encode(), ctypes.byref(v)), "detops_tuning_get(%s)" % key)
    return v.value
