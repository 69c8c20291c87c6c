"""Randomised small-shape sweeps of the emulated HIP kernels against the oracle (CPU only).  Shapes,
ROI geometry, bin counts, sampling ratios and channel counts are drawn per case from a seeded
generator, so every run covers the same 60-odd configurations — odd sizes, ROIs outside the map,
degenerate ROIs, maps smaller than a tile, channel counts that are not multiples of any chunk."""
import numpy as np
import pytest

import emu
import oracle


def _rois(rng, K, N, H, W, scale):
    img_w, img_h = W / scale, H / scale
    x1 = rng.uniform(-0.3 * img_w, 1.1 * img_w, K)
    y1 = rng.uniform(-0.3 * img_h, 1.1 * img_h, K)
    w = np.exp(rng.uniform(np.log(0.5), np.log(1.5 * img_w), K))
    h = np.exp(rng.uniform(np.log(0.5), np.log(1.5 * img_h), K))
    r = np.stack([rng.randint(0, N, K), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    if K > 2:
        r[0, 3:] = r[0, 1:3]           # zero-size ROI
        r[1, 1:] = [-1e4, -1e4, -9e3, -9e3]  # far outside
    return r


@pytest.mark.parametrize("seed", range(24))
def test_emu_fuzz_roi_align(seed):
    rng = np.random.RandomState(1000 + seed)
    N = int(rng.randint(1, 3))
    C = int(rng.choice([1, 3, 4, 5, 16, 17, 33]))
    H, W = int(rng.randint(1, 40)), int(rng.randint(1, 70))
    K = int(rng.choice([1, 2, 7, 40, 70]))
    ph, pw = [(7, 7), (14, 14), (1, 1), (2, 5), (5, 2), (7, 7), (3, 3), (9, 4)][seed % 8]
    sr = int(rng.choice([0, 1, 2, 3]))
    scale = float(rng.choice([1.0, 0.5, 0.25, 0.0625]))
    x = rng.randn(N, C, H, W).astype(np.float32)
    rois = _rois(rng, K, N, H, W, scale)
    out = emu.roi_align_forward(x, rois, scale, ph, pw, sr)
    assert np.array_equal(out, oracle.roi_align_forward(x, rois, scale, ph, pw, sr)), "forward must be bit-equal"
    g = rng.randn(K, C, ph, pw).astype(np.float32)
    ref = oracle.roi_align_backward(g, rois, scale, ph, pw, N, C, H, W, sr, acc64=True)
    tol = 2e-5 * max(1.0, np.abs(ref).max())
    impls = [("scan", 2), ("atomic", 3)] + ([("ring", 1), ("acc", 4)] if (ph, pw) in ((7, 7), (14, 14)) else [])
    for name, impl in impls:
        emu.tuning_set("roi_bwd_impl", impl)
        if impl == 1:
            emu.tuning_set("roi_bwd_seg", 8 + seed % 3)        # split whatever is crowded
        if impl == 4:
            emu.tuning_set("roi_bwd_groups", [0, 1, 3][seed % 3])   # auto / one group (direct store) / three partial maps
        emu.stats(reset=True)
        got = emu.roi_align_backward(g, rois, scale, ph, pw, N, C, H, W, sr)
        assert np.abs(got - ref).max() <= tol, name
        if impl == 4:   # maps beyond 32 x 32 (or beyond the LDS budget) fall through to the scan kernel
            if H > 32 or W > 32:
                assert emu.stats().get("bwda.units", 0) == 0, (H, W)
            elif ph == 7:
                assert emu.stats().get("bwda.units", 0) > 0, (H, W)
            emu.tuning_set("roi_bwd_groups", 0)
    if seed % 3 == 0:
        emu.tuning_set("roi_bwd_impl", 2)
        emu.tuning_set("roi_bwd_groups", 5)
        got = emu.roi_align_backward(g, rois, scale, ph, pw, N, C, H, W, sr)
        assert np.abs(got - ref).max() <= tol, "ROI-list split"


@pytest.mark.parametrize("seed", range(16))
def test_emu_fuzz_nms(seed):
    rng = np.random.RandomState(2000 + seed)
    n = int(rng.choice([1, 3, 64, 65, 127, 128, 300, 513]))
    cx, cy = rng.uniform(0, 200, n), rng.uniform(0, 200, n)
    w, h = rng.uniform(1, 80, n), rng.uniform(1, 80, n)
    boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
    if seed % 2:
        boxes = np.round(boxes)                     # integer boxes: many exact IoU == threshold ties
    scores = rng.rand(n).astype(np.float32)
    if seed % 4 == 1:
        scores = np.round(scores, 1)                # score ties: stable order
    if seed % 4 == 2 and n > 3:
        boxes[2] = boxes[1]; scores[2] = scores[1]  # duplicates
    thr = float(rng.choice([0.0, 0.3, 0.5, 0.7, 1.0]))
    assert np.array_equal(emu.nms(boxes, scores, thr), oracle.nms(boxes, scores, thr))


@pytest.mark.parametrize("seed", range(10))
def test_emu_fuzz_deformable(seed):
    rng = np.random.RandomState(3000 + seed)
    B = int(rng.randint(1, 3))
    dg = int(rng.choice([1, 2]))
    C = dg * int(rng.choice([1, 3, 8, 17]))
    H, W = int(rng.randint(3, 14)), int(rng.randint(3, 20))
    k = int(rng.choice([1, 3]))
    stride, pad, dil = int(rng.choice([1, 2])), int(rng.choice([0, 1, 2])), int(rng.choice([1, 2]))
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    if Ho < 1 or Wo < 1:
        pytest.skip("empty output")
    x = rng.randn(B, C, H, W).astype(np.float32)
    off = (rng.randn(B, dg * 2 * k * k, Ho, Wo) * rng.choice([0.0, 0.7, 3.0])).astype(np.float32)
    mask = rng.rand(B, dg * k * k, Ho, Wo).astype(np.float32) if seed % 2 else None
    geo = dict(kh=k, kw=k, pad=(pad, pad), stride=(stride, stride), dil=(dil, dil), dg=dg)
    ref_col = oracle.deformable_im2col(x, off, mask, **geo)
    np.testing.assert_allclose(emu.deformable_im2col(x, off, mask, **geo), ref_col, rtol=1e-5, atol=1e-5)
    gcol = rng.randn(*ref_col.shape).astype(np.float32)
    ref = oracle.deformable_col2im(gcol, off, mask, B, C, H, W, **geo)
    tol = dict(rtol=1e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(emu.deformable_col2im(gcol, off, mask, B, C, H, W, gather=True, **geo), ref, **tol)
    np.testing.assert_allclose(emu.deformable_col2im(gcol, off, mask, B, C, H, W, gather=False, **geo), ref, **tol)
    goff, gmask = emu.deformable_col2im_coord(gcol, x, off, mask, **geo)
    roff, rmask = oracle.deformable_col2im_coord(gcol, x, off, mask, **geo)
    np.testing.assert_allclose(goff, roff, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(roff).max()))
    if mask is not None:
        np.testing.assert_allclose(gmask, rmask, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(rmask).max()))


@pytest.mark.parametrize("seed", range(8))
def test_emu_fuzz_nms_batched_single_launch(seed):
    """ragged segment sets through the ONE-launch path (sort, mask tiles and scan as workgroup roles of one kernel) and
    through the three launches: random segment counts and lengths over every scan width, integer boxes (exact IoU ties),
    score ties, empty segments"""
    rng = np.random.RandomState(5000 + seed)
    S = int(rng.randint(1, 9))
    sizes = [int(rng.choice([0, 1, 2, 63, 64, 65, 130, 700, 1025, 1500, 2049])) for _ in range(S)]
    segs = []
    for n in sizes:
        cx, cy = rng.uniform(0, 300, n), rng.uniform(0, 300, n)
        w, h = rng.uniform(1, 90, n), rng.uniform(1, 90, n)
        b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
        if seed % 2:
            b = np.round(b)
        s = rng.rand(n).astype(np.float32)
        if seed % 3 == 1:
            s = np.round(s, 1)
        segs.append((b.reshape(n, 4), s))
    boxes = np.concatenate([b for b, _ in segs]) if sum(sizes) else np.zeros((0, 4), np.float32)
    scores = np.concatenate([s for _, s in segs]) if sum(sizes) else np.zeros((0,), np.float32)
    offs = np.cumsum([0] + sizes).astype(np.int32)
    thr = float(rng.choice([0.3, 0.5, 0.7]))
    max_n = max(max(sizes), 1)
    for fused in (0, 2):
        emu.tuning_set("nms_fused", fused)
        keep, num = emu.nms_batched(boxes, scores, offs, max_n, thr)
        km, num2 = emu.nms_batched(boxes, scores, offs, max_n, thr, mask=True)
        assert np.array_equal(num, num2)
        for i, (b, s) in enumerate(segs):
            ref = oracle.nms(b, s, thr) if len(s) else np.zeros(0, np.int64)
            assert num[i] == len(ref), (fused, i, sizes[i])
            assert np.array_equal(keep[offs[i]:offs[i] + num[i]], ref), (fused, i, sizes[i])
            want = np.zeros(len(s), np.uint8)
            want[ref] = 1
            assert np.array_equal(km[offs[i]:offs[i + 1]], want)


@pytest.mark.parametrize("seed", range(8))
def test_emu_fuzz_roi_align_fpn_fused(seed):
    """the multi-level launches the pooler uses (level mapping on the device, LDS-DMA forward, ring backward with split
    hit lists) on random pyramids: odd map sizes, 2-4 levels, ROIs of every scale incl. degenerate ones"""
    rng = np.random.RandomState(6000 + seed)
    N = int(rng.randint(1, 3))
    C = int(rng.choice([4, 16, 17, 32]))
    L = int(rng.randint(2, 5))
    H0, W0 = int(rng.randint(24, 64)), int(rng.randint(24, 80))
    shapes = [(N, C, -(-H0 // (1 << l)), -(-W0 // (1 << l))) for l in range(L)]
    scales = [1.0 / (4 << l) for l in range(L)]
    feats = [rng.randn(*s).astype(np.float32) for s in shapes]
    K = int(rng.choice([3, 40, 150]))
    rois = _rois(rng, K, N, shapes[0][2], shapes[0][3], scales[0])
    ph = [7, 14][seed % 2]
    out, lv = emu.roi_align_fpn_forward(feats, rois, scales, ph, ph, 2, 2, 2 + L - 1)
    ref_lv = oracle.fpn_level(rois, 2, 2 + L - 1, 224.0, 4.0, 1e-6)
    assert np.array_equal(lv, ref_lv)
    for l in range(L):
        sel = np.nonzero(lv == l)[0]
        if sel.size:
            assert np.array_equal(out[sel], oracle.roi_align_forward(feats[l], rois[sel], scales[l], ph, ph, 2)), l
    g = rng.randn(K, C, ph, ph).astype(np.float32)
    emu.tuning_set("roi_bwd_seg", 4 + seed % 5)
    gins = emu.roi_align_fpn_backward(g, rois, lv, shapes, scales, ph, ph, 2)
    again = emu.roi_align_fpn_backward(g, rois, lv, shapes, scales, ph, ph, 2)
    for l in range(L):
        sel = np.nonzero(lv == l)[0]
        ref = oracle.roi_align_backward(g[sel], rois[sel], scales[l], ph, ph, *shapes[l], 2, acc64=True) if sel.size \
            else np.zeros(shapes[l], np.float32)
        assert np.abs(gins[l] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), l
        assert np.array_equal(gins[l], again[l]), "bit-reproducible"


@pytest.mark.parametrize("seed", range(6))
def test_emu_fuzz_deformable_channels_last(seed):
    """the channels-last pipeline (NHWC im2col, in-wave coordinate-gradient reduction, transposed sampling + GEMMs) on
    random geometries against the oracle's reference-layout arithmetic"""
    rng = np.random.RandomState(7000 + seed)
    B = int(rng.randint(1, 3))
    C, Cout = int(rng.choice([64, 128])), int(rng.choice([64, 128, 256]))   # fp32: channel counts / 4 a power of two >= 16
    H, W = int(rng.randint(4, 10)), int(rng.randint(4, 12))
    k = 3
    stride, pad, dil = int(rng.choice([1, 2])), int(rng.choice([0, 1, 2])), int(rng.choice([1, 2]))
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    if Ho < 1 or Wo < 1:
        pytest.skip("empty output")
    x = rng.randn(B, C, H, W).astype(np.float32)
    off = (rng.randn(B, 2 * k * k, Ho, Wo) * rng.choice([0.0, 0.7, 3.0])).astype(np.float32)
    mask = rng.rand(B, k * k, Ho, Wo).astype(np.float32) if seed % 2 else None
    wgt = (rng.randn(Cout, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    go = rng.randn(B, Cout, Ho, Wo).astype(np.float32)
    geo = ((pad, pad), (stride, stride), (dil, dil))
    out, gin, goff, gmask, gw = emu.deformable_nhwc(x, off, mask, wgt, go, k, k, *geo)
    ref_out = oracle.deform_conv_forward(x, off, mask, wgt, None, *geo, 1, 1)
    np.testing.assert_allclose(out, ref_out, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref_out).max()))
    rin, roff, rmask, rw, _ = oracle.deform_conv_backward(x, off, mask, wgt, go, False, *geo, 1, 1)
    tol = lambda r: dict(rtol=1e-4, atol=2e-4 * max(1.0, np.abs(r).max()))  # noqa: E731
    np.testing.assert_allclose(gin, rin, **tol(rin))
    np.testing.assert_allclose(goff, roff, **tol(roff))
    np.testing.assert_allclose(gw, rw, **tol(rw))
    if mask is not None:
        np.testing.assert_allclose(gmask, rmask, **tol(rmask))
