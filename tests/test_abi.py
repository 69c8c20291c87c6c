"""CPU-side checks of the drop-in boundary (`-m "not gpu"`): the C-ABI library builds for gfx950,
loads, exports every symbol include/detops.h declares, and the Python mirror of the reference
operator API has the reference's names and fails loudly without a GPU path."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "detops.h")
CSRC = os.path.join(ROOT, "maskrcnn-benchmark_amd", "csrc")
LIB = os.path.join(ROOT, "maskrcnn-benchmark_amd", "maskrcnn_benchmark", "lib", "libdetops_gfx950.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(detops_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", CSRC, "-s"])
    return ctypes.CDLL(LIB)


def test_header_declares_the_hot_path_entry_points():
    names = declared_symbols()
    for must in ("detops_roi_align_forward_f32", "detops_roi_align_backward_f32", "detops_nms_f32",
                 "detops_nms_batched_f32", "detops_roi_pool_forward_f32", "detops_roi_pool_backward_f32",
                 "detops_sigmoid_focal_loss_forward_f32", "detops_sigmoid_focal_loss_backward_f32",
                 "detops_deformable_im2col", "detops_deformable_col2im", "detops_deformable_col2im_coord"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing


def test_library_identifies_itself(lib):
    arch = ctypes.c_char_p()
    lib.detops_version.restype = ctypes.c_int
    assert lib.detops_version(ctypes.byref(arch)) == 1
    assert arch.value == b"gfx950"


def test_python_binding_covers_header_and_reference_names(lib):
    from maskrcnn_benchmark import _C, _lib

    assert sorted(_lib.SIGNATURES) == declared_symbols()
    # the 14 names of the reference's pybind module (csrc/vision.cpp:10-24)
    for name in ("nms", "roi_align_forward", "roi_align_backward", "roi_pool_forward",
                 "roi_pool_backward", "sigmoid_focalloss_forward", "sigmoid_focalloss_backward",
                 "deform_conv_forward", "deform_conv_backward_input", "deform_conv_backward_parameters",
                 "modulated_deform_conv_forward", "modulated_deform_conv_backward",
                 "deform_psroi_pooling_forward", "deform_psroi_pooling_backward"):
        assert callable(getattr(_C, name)), name


def test_layers_api_names_match_reference():
    import maskrcnn_benchmark.layers as L

    expected = ["nms", "roi_align", "ROIAlign", "roi_pool", "ROIPool", "smooth_l1_loss", "Conv2d",
                "DFConv2d", "ConvTranspose2d", "interpolate", "BatchNorm2d", "FrozenBatchNorm2d",
                "SigmoidFocalLoss", "deform_conv", "modulated_deform_conv", "DeformConv",
                "ModulatedDeformConv", "ModulatedDeformConvPack", "deform_roi_pooling",
                "DeformRoIPooling", "DeformRoIPoolingPack", "ModulatedDeformRoIPoolingPack"]
    assert L.__all__ == expected  # reference layers/__init__.py:23-46
    for n in expected:
        assert hasattr(L, n)


def test_no_cpu_fallback_ops_fail_loudly():
    """CPU tensors raise instead of silently computing — except for the two operators the reference itself serves
    on the CPU (nms, ROIAlign_forward: tests/test_cpu_branch.py); their backward and every other operator raise,
    and mixing devices raises."""
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.layers import ROIAlign, SigmoidFocalLoss, deform_conv

    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        _C.roi_align_backward(torch.zeros(1, 2, 7, 7), torch.zeros(1, 5), 0.25, 7, 7, 1, 2, 8, 8, 2)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        x = torch.zeros(1, 2, 8, 8, requires_grad=True)
        ROIAlign((7, 7), 0.25, 2)(x, torch.zeros(1, 5)).sum().backward()
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        _C.nms_batched_mask(torch.zeros(2, 4), torch.zeros(2), torch.tensor([0, 2], dtype=torch.int32), 2, 0.5)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        _C.roi_pool_forward(torch.zeros(1, 2, 8, 8), torch.zeros(1, 5), 0.25, 7, 7)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        SigmoidFocalLoss(2.0, 0.25)(torch.zeros(4, 3), torch.zeros(4, dtype=torch.int32))
    with pytest.raises(NotImplementedError):
        deform_conv(torch.zeros(1, 2, 5, 5), torch.zeros(1, 18, 5, 5), torch.zeros(2, 2, 3, 3), 1, 1)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        _C.sigmoid_focalloss_forward(torch.zeros(4, 3), torch.zeros(4, dtype=torch.int32), 3, 2.0, 0.25)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        _C.deform_psroi_pooling_forward(torch.zeros(1, 9, 4, 4), torch.zeros(1, 5), torch.zeros(0),
                                        torch.zeros(1, 1, 3, 3), torch.zeros(1, 1, 3, 3), 1, 1.0, 1, 3, 3, 3, 4, 0.0)
    from maskrcnn_benchmark.layers import DeformRoIPooling

    with pytest.raises(NotImplementedError):  # reference layers/dcn/deform_pool_func.py:29-30
        DeformRoIPooling(1.0, 3, 1, True, 3)(torch.zeros(1, 9, 4, 4), torch.zeros(1, 5), torch.zeros(0))


def test_argument_validation_without_a_gpu(lib):
    """Entry points validate shapes before touching the device (no GPU needed)."""
    lib.detops_roi_align_forward_f32.restype = ctypes.c_int
    f = ctypes.c_float
    rc = lib.detops_roi_align_forward_f32(None, None, None, 1, 4, 8, 8, 3, 0, 7, f(1.0), 2, None)
    assert rc == -1  # PH == 0 -> DETOPS_EINVAL
    rc = lib.detops_roi_align_forward_f32(None, None, None, 1, 4, 8, 8, 0, 7, 7, f(1.0), 2, None)
    assert rc == 0  # K == 0 is a no-op
    ps = lib.detops_deform_psroi_pool_forward_f32
    ps.restype = ctypes.c_int
    # C = 8 < output_dim * group_size^2 = 9 -> DETOPS_EINVAL; K == 0 -> no-op
    assert ps(None, None, None, None, None, 1, 8, 4, 4, 1, 2, 1, f(1.0), 1, 3, 3, 3, 4, f(0.0), None) == -1
    assert ps(None, None, None, None, None, 1, 9, 4, 4, 0, 2, 1, f(1.0), 1, 3, 3, 3, 4, f(0.0), None) == 0
    lib.detops_nms_workspace_bytes.restype = ctypes.c_size_t
    assert lib.detops_nms_workspace_bytes(2000) >= 2000 * 32 * 8
    assert lib.detops_nms_workspace_bytes(0) > 0


def test_empty_layers_shims_on_cpu():
    from maskrcnn_benchmark.layers import Conv2d, ConvTranspose2d, FrozenBatchNorm2d, interpolate, smooth_l1_loss

    x = torch.zeros(0, 4, 14, 14, requires_grad=True)
    y = Conv2d(4, 8, 3, stride=2, padding=1)(x)
    assert y.shape == (0, 8, 7, 7)
    y.sum().backward()
    assert ConvTranspose2d(4, 6, 2, 2, 0)(torch.zeros(0, 4, 14, 14)).shape == (0, 6, 28, 28)
    assert interpolate(torch.zeros(0, 4, 14, 14), scale_factor=2).shape == (0, 4, 28, 28)
    bn = FrozenBatchNorm2d(4)
    bn.weight.fill_(2.0); bn.running_var.fill_(4.0); bn.running_mean.fill_(1.0); bn.bias.fill_(0.5)
    torch.testing.assert_close(bn(torch.ones(1, 4, 2, 2)), torch.full((1, 4, 2, 2), 0.5))
    a, b = torch.tensor([0.0, 1.0, 0.05]), torch.tensor([0.0, 0.0, 0.0])
    torch.testing.assert_close(smooth_l1_loss(a, b, beta=0.11, size_average=False),
                               torch.tensor(0.0 + (1.0 - 0.055) + 0.5 * 0.05 ** 2 / 0.11))
