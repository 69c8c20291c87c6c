"""GPU parity of the OPT-IN kernel variants that round 1 could only check in the host emulation
(DESIGN.md section 7).  Skipped unless DETOPS_TEST_EXPERIMENTAL=1, so the default `-m gpu` suite only
covers the validated defaults:

    DETOPS_TEST_EXPERIMENTAL=1 python -m pytest tests/test_experimental_gpu.py -m gpu -q
"""
import os

import numpy as np
import pytest
import torch

import oracle
import synth

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("DETOPS_TEST_EXPERIMENTAL") != "1",
                                 reason="opt-in variants: set DETOPS_TEST_EXPERIMENTAL=1")]
DEV = "cuda"


def _C():
    from maskrcnn_benchmark import _C as C
    return C


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _fpn_case(C, K, ph):
    shapes = [(2, C, h, w) for (h, w) in synth.fpn_shapes()[:4]]
    scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
    rois = synth.fpn_rois(seed=7, per_image=K // 2)
    return shapes, scales, rois, synth.level_map(rois)


@pytest.mark.parametrize("env", [{"DETOPS_ROIALIGN_BWD_WALK": "lane"},
                                 {"DETOPS_ROIALIGN_BWD": "gather3"},
                                 {"DETOPS_ROIALIGN_BWD": "gather3", "DETOPS_ROIALIGN_BWD_G": "2"},
                                 {"DETOPS_ROIALIGN_BWD": "gather3", "DETOPS_ROIALIGN_BWD_BATCH": "3"}])
@pytest.mark.parametrize("K,ph", [(1024, 7), (256, 14)])
def test_roi_align_backward_variants_match_default_and_oracle(env, K, ph, monkeypatch):
    C = _C()
    shapes, scales, rois, lv = _fpn_case(256, K, ph)
    g = torch.randn(K, 256, ph, ph, device=DEV)
    base = C.roi_align_fpn_backward(g, _t(rois), _t(lv), shapes, scales, ph, ph, 2)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got = C.roi_align_fpn_backward(g, _t(rois), _t(lv), shapes, scales, ph, ph, 2)
    again = C.roi_align_fpn_backward(g, _t(rois), _t(lv), shapes, scales, ph, ph, 2)
    for a, b, c in zip(base, got, again):
        torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-4)
        assert torch.equal(b, c), "variants are atomic-free too: bit-reproducible"
    l = 2  # P4 against the oracle
    idx = np.nonzero(lv == l)[0]
    ref = oracle.roi_align_backward(g[idx].cpu().numpy(), rois[idx], scales[l], ph, ph, *shapes[l], 2, acc64=True)
    assert np.abs(got[l].cpu().numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("K,ph", [(1024, 7), (256, 14), (3, 7)])
def test_roi_align_forward_ordered_is_bit_identical(K, ph, monkeypatch):
    C = _C()
    shapes, scales, rois, lv = _fpn_case(64, max(K, 2), ph)
    rois = rois[:K]
    feats = [torch.randn(*s, device=DEV) for s in shapes]
    base, lv0 = C.roi_align_fpn_forward(feats, _t(rois), scales, ph, ph, 2, 2, 5)
    monkeypatch.setenv("DETOPS_ROIALIGN_FWD_ORDER", "1")
    out, lv1 = C.roi_align_fpn_forward(feats, _t(rois), scales, ph, ph, 2, 2, 5)
    assert torch.equal(out, base) and torch.equal(lv0, lv1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(128, 100, 168), (512, 25, 42)])
def test_deformable_col2im_ell_matches_default(shape, dtype, monkeypatch):
    C = _C()
    Cc, H, W = shape
    x = torch.randn(2, Cc, H, W, device=DEV).to(dtype)
    off = (torch.randn(2, 18, H, W, device=DEV) * 2).to(dtype)
    off[:, :, :3] += 7.0   # a band of larger offsets: some (pixel, tap) columns overflow 8 slots
    geo = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    col = torch.randn(Cc * 9, 2 * H * W, device=DEV).to(dtype)
    base = torch.zeros_like(x)
    C.deformable_col2im(col, off, None, base, *geo)
    monkeypatch.setenv("DETOPS_DCN_COL2IM", "ell")
    got = torch.zeros_like(x)
    C.deformable_col2im(col, off, None, got, *geo)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    torch.testing.assert_close(got.float(), base.float(), rtol=tol, atol=tol * max(1.0, float(base.float().abs().max())))
