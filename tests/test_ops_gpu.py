"""Parity tests of the HIP operators against the oracle — run with `-m gpu` on an MI355X.

Every call goes through `maskrcnn_benchmark._C` / `maskrcnn_benchmark.layers`, i.e. through the
C ABI of libdetops_gfx950.so.  Tolerances are the ones BASELINE.json's north_star states:
NMS kept indices bit-exact; ROIAlign / focal loss within 1e-4 (fp32); DCN fp32 1e-4, half 2e-2.
"""
import os

import numpy as np
import pytest
import torch

import oracle
import synth

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _C():
    from maskrcnn_benchmark import _C as C

    return C


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _tune(key, value):
    from maskrcnn_benchmark import _lib
    _lib.tuning_set(key, value)


def _close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ============================================================================ ROIAlign forward
@pytest.mark.parametrize("ph,pw,sr", [(7, 7, 2), (14, 14, 2), (7, 7, 0), (3, 5, 3), (1, 1, 1)])
def test_roi_align_forward_cfg1_bit_exact(ph, pw, sr):
    """BASELINE configs[0]: 256x14x14 map, 512 ROIs — bit-identical to the reference CPU kernel
    (the oracle is pinned bit-exact to it)."""
    inp, rois, scale = synth.cfg1_roi_align()
    ref = oracle.roi_align_forward(inp, rois, scale, ph, pw, sr)
    out = _C().roi_align_forward(_t(inp), _t(rois), scale, ph, pw, sr).cpu().numpy()
    assert np.abs(out - ref).max() <= 1e-4  # the stated tolerance
    assert np.array_equal(out, ref), "max abs diff %g" % np.abs(out - ref).max()


def test_roi_align_forward_golden_reference_vectors(golden_dir):
    g = _load(golden_dir, "ref_cpu_vectors.npz")
    for i in range(4):
        ph, pw, sr = [int(v) for v in g[f"ra_cfg_{i}"]]
        out = _C().roi_align_forward(_t(g["ra_input"]), _t(g["ra_rois"]), float(g["ra_scale"]), ph, pw, sr)
        assert np.array_equal(out.cpu().numpy(), g[f"ra_out_{i}"])
    for i in range(2):
        ph, pw, sr = [int(v) for v in g[f"ra2_cfg_{i}"]]
        out = _C().roi_align_forward(_t(g["ra2_input"]), _t(g["ra2_rois"]), 1.0 / 32, ph, pw, sr)
        assert np.array_equal(out.cpu().numpy(), g[f"ra2_out_{i}"])


def test_roi_align_forward_fpn_full_size_and_fused():
    """cfg-2: 1024 ROIs on the 2-image 800x1344 pyramid, per level and in the fused launch."""
    feats = synth.fpn_features()
    rois = synth.fpn_rois()
    lv = synth.level_map(rois)
    scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
    tf = [_t(f) for f in feats]
    expected = np.zeros((rois.shape[0], 256, 7, 7), np.float32)
    for l in range(4):
        idx = np.nonzero(lv == l)[0]
        assert len(idx) > 0
        ref = oracle.roi_align_forward(feats[l], rois[idx], scales[l], 7, 7, 2)
        out = _C().roi_align_forward(tf[l], _t(rois[idx]), scales[l], 7, 7, 2).cpu().numpy()
        assert np.array_equal(out, ref)
        expected[idx] = ref
    out, levels = _C().roi_align_fpn_forward(tf, _t(rois), scales, 7, 7, 2, 2, 5)
    np.testing.assert_array_equal(levels.cpu().numpy(), lv)
    assert np.array_equal(out.cpu().numpy(), expected)


def test_roi_align_forward_variants_bit_equal():
    """The ROI ranking pre-pass only changes the order workgroups visit the ROIs in, the LDS-DMA kernel only how
    the footprint reaches LDS: every variant gives the same bits (cfg-2 box head and cfg-3 mask head, full size)."""
    feats = [torch.randn(2, 256, h, w, device=DEV) for (h, w) in synth.fpn_shapes()[:4]]
    scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
    for K, ph in ((1024, 7), (256, 14)):
        rois = _t(synth.fpn_rois(per_image=K // 2))
        base, lv = _C().roi_align_fpn_forward(feats, rois, scales, ph, ph, 2, 2, 5)
        assert torch.isfinite(base).all()
        from maskrcnn_benchmark import _lib
        # roi_fwd_records: 1 = no per-ROI sample records (every workgroup derives its ROI's geometry itself), 2 = records
        # with the incremental staging offsets; default = records + per-lane static staging offsets
        for env in ({"roi_fwd_order": 1}, {"roi_fwd_order": 2, "roi_fwd_order_mink": 64}, {"roi_fwd_impl": 1},
                    {"roi_fwd_records": 1}, {"roi_fwd_records": 2}, {"roi_fwd_records": 1, "roi_fwd_order": 1},
                    {"roi_fwd_ct": 64}):
            for k, v in env.items():
                _lib.tuning_set(k, v)
            out, lv2 = _C().roi_align_fpn_forward(feats, rois, scales, ph, ph, 2, 2, 5)
            for k in env:
                _lib.tuning_set(k, 0)
            assert torch.equal(out, base) and torch.equal(lv, lv2), env


@pytest.mark.parametrize("K", [384, 415, 4096, 4097])
def test_roi_align_forward_ranking_prepass_size_limits(K):
    """ROI counts at the edges of the ranking pre-pass (384 .. 4096; 4097 runs unranked), not multiples of its
    LDS padding, duplicates included: bit-equal to the oracle either way"""
    from maskrcnn_benchmark import _lib
    _lib.tuning_set("roi_fwd_order", 2)
    rng = np.random.RandomState(K)
    feats = [rng.randn(2, 8, 50, 84).astype(np.float32), rng.randn(2, 8, 25, 42).astype(np.float32)]
    scales = [0.25, 0.125]
    rois = synth.fpn_rois(seed=K, per_image=(K + 1) // 2, n_images=2, smin=8, smax=300)[:K]
    rois[:, 1:] *= 0.25
    rois[5] = rois[6]
    out, levels = _C().roi_align_fpn_forward([_t(f) for f in feats], _t(rois), scales, 7, 7, 2, 2, 3)
    out, levels = out.cpu().numpy(), levels.cpu().numpy()
    for l in range(2):
        sel = levels == l
        assert np.array_equal(out[sel], oracle.roi_align_forward(feats[l], rois[sel], scales[l], 7, 7, 2))


def test_roi_align_forward_edge_cases():
    C = _C()
    x = torch.randn(2, 3, 10, 12, device=DEV)
    assert C.roi_align_forward(x, torch.zeros(0, 5, device=DEV), 0.5, 7, 7, 2).shape == (0, 3, 7, 7)
    rois = np.array([[0, -500, -500, -400, -400],     # entirely outside -> zeros
                     [1, 0, 0, 5000, 4000],           # huge adaptive grid -> on-the-fly path
                     [1, 3, 3, 3, 3],                 # degenerate -> forced 1x1
                     [0, 5.5, 2.25, 20.75, 17.5]], np.float32)
    xn = x.cpu().numpy()
    for sr in (0, 2):
        ref = oracle.roi_align_forward(xn, rois, 1.0, 4, 6, sr)
        out = C.roi_align_forward(x, _t(rois), 1.0, 4, 6, sr).cpu().numpy()
        assert np.array_equal(out, ref), sr
        assert not out[0].any()
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):     # mixed devices: no silent copies
        C.roi_align_forward(x, _t(rois).cpu(), 1.0, 4, 6, 2)
    assert np.array_equal(C.roi_align_forward(x.cpu(), _t(rois).cpu(), 1.0, 4, 6, 2).numpy(),   # CPU branch = device
                          C.roi_align_forward(x, _t(rois), 1.0, 4, 6, 2).cpu().numpy())


# ============================================================================ ROIAlign backward
@pytest.fixture(params=["ring", "ring-ct16", "scan", "atomic", "acc"])
def bwd_impl(request):
    """The three backward kernels behind the same entry points: the ring pixel-owner kernel (default for filled
    launches of the model's bin shapes; forced here also for under-filled ones), the scan pixel-owner kernel (small
    maps, other shapes), the atomic scatter kernel (universal fallback) and the acc kernel (one small map: the map lives
    in LDS; shapes outside its plan fall through to the scan kernel).  tests/conftest.py resets the switch."""
    from maskrcnn_benchmark import _lib
    _lib.tuning_set("roi_bwd_impl", {"ring": 1, "ring-ct16": 1, "scan": 2, "atomic": 3, "acc": 4}[request.param])
    _lib.tuning_set("roi_bwd_ct", 16 if request.param == "ring-ct16" else 0)
    return request.param


@pytest.mark.parametrize("ph,pw,sr", [(7, 7, 2), (14, 14, 2), (7, 7, 0), (3, 5, 3)])
def test_roi_align_backward_cfg1(ph, pw, sr, bwd_impl):
    inp, rois, scale = synth.cfg1_roi_align()
    g = np.random.RandomState(1).randn(rois.shape[0], 256, ph, pw).astype(np.float32)
    ref = oracle.roi_align_backward(g, rois, scale, ph, pw, *inp.shape, sr, acc64=True)
    out = _C().roi_align_backward(_t(g), _t(rois), scale, ph, pw, *inp.shape, sr).cpu().numpy()
    # 512 ROIs pile onto a 14x14 map: |grad_in| ~ 1e2; 1e-4 relative to the result's scale
    assert np.abs(out - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


def test_roi_align_backward_fpn_full_size_and_fused(bwd_impl):
    feats_shapes = [(2, 256, h, w) for (h, w) in synth.fpn_shapes()[:4]]
    rois = synth.fpn_rois()
    lv = synth.level_map(rois)
    scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
    g = np.random.RandomState(2).randn(rois.shape[0], 256, 7, 7).astype(np.float32)
    refs = []
    for l in (0, 1, 2, 3):  # every level against the fp64-accumulated oracle (P2 = 75 % of the gradient bytes)
        idx = np.nonzero(lv == l)[0]
        ref = oracle.roi_align_backward(g[idx], rois[idx], scales[l], 7, 7, *feats_shapes[l], 2, acc64=True)
        out = _C().roi_align_backward(_t(g[idx]), _t(rois[idx]), scales[l], 7, 7, *feats_shapes[l], 2)
        _close(out, ref, rtol=1e-4, atol=1e-4)
        refs.append(ref)
    gins = _C().roi_align_fpn_backward(_t(g), _t(rois), _t(lv), feats_shapes, scales, 7, 7, 2)
    for l in (0, 1, 2, 3):
        _close(gins[l], refs[l], rtol=1e-4, atol=1e-4)


def test_roi_align_backward_tile_seams_and_accumulate_flag(bwd_impl):
    """ROIs straddling several 32x64 gradient tiles (odd map size, multi-image) against the oracle,
    and the C ABI's zero_grad_in = 0 mode (accumulate into the caller's buffer)."""
    import ctypes
    from maskrcnn_benchmark import _lib
    rng = np.random.RandomState(11)
    N, C, H, W = 2, 6, 75, 150
    K = 160
    x1 = rng.uniform(-20, 560, K); y1 = rng.uniform(-20, 280, K)
    rois = np.stack([rng.randint(0, N, K), x1, y1, x1 + rng.uniform(2, 400, K), y1 + rng.uniform(2, 250, K)], 1).astype(np.float32)
    for (ph, pw, sr) in ((7, 7, 2), (14, 14, 2), (5, 3, 0)):
        g = rng.randn(K, C, ph, pw).astype(np.float32)
        ref = oracle.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr, acc64=True)
        out = _C().roi_align_backward(_t(g), _t(rois), 0.25, ph, pw, N, C, H, W, sr)
        _close(out, ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref).max()))
        base = rng.randn(N, C, H, W).astype(np.float32)
        buf = _t(base)
        tg, tr = _t(g), _t(rois)
        # straight through the C ABI with a caller workspace (the binned kernel needs one; the others ignore it)
        Hs, Ws = (ctypes.c_int * 1)(H), (ctypes.c_int * 1)(W)
        nbytes = int(_lib.lib.detops_roi_align_backward_workspace_bytes(Hs, Ws, 1, N, C, K, ph, pw))
        ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=DEV)
        rc = _lib.lib.detops_roi_align_backward_ws_f32(tg.data_ptr(), tr.data_ptr(), buf.data_ptr(), N, C, H, W, K, ph, pw,
                                                       ctypes.c_float(0.25), sr, 0, ws.data_ptr(), nbytes,
                                                       torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        _close(buf, base + ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref).max()))
    # K = 0 with zero-fill: a pure clear
    buf = _t(rng.randn(N, C, H, W).astype(np.float32))
    e = torch.empty((0, 5), device=DEV)
    rc = _lib.lib.detops_roi_align_backward_f32(None, e.data_ptr(), buf.data_ptr(), N, C, H, W, 0, 7, 7, ctypes.c_float(0.25), 2, 1,
                                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0 and float(buf.abs().sum()) == 0.0


def test_roi_align_forward_lds_path_large_and_tiny_rois():
    """fixed-grid fast path: footprints larger than the LDS budget (falls back to gathers inside the
    kernel), 1-pixel ROIs, ROIs hanging over every border — all bit-equal to the oracle."""
    rng = np.random.RandomState(12)
    N, C, H, W = 2, 40, 120, 200
    inp = rng.randn(N, C, H, W).astype(np.float32)
    rois = np.array([[0, 0, 0, 799, 479], [1, -50, -50, 900, 600], [0, 10.3, 20.7, 10.9, 21.2], [1, 790, 470, 830, 500],
                     [0, 400, 0, 420, 479], [1, 0, 200, 799, 210], [0, 795.9, 475.9, 796.0, 476.0]], np.float32)
    rois = np.concatenate([rois, np.stack([rng.randint(0, N, 60), rng.uniform(0, 700, 60), rng.uniform(0, 400, 60),
                                           rng.uniform(0, 799, 60), rng.uniform(0, 479, 60)], 1).astype(np.float32)])
    rois[7:, 3] = np.maximum(rois[7:, 3], rois[7:, 1]); rois[7:, 4] = np.maximum(rois[7:, 4], rois[7:, 2])
    for (ph, pw, sr) in ((7, 7, 2), (14, 14, 2), (7, 7, 1)):
        out = _C().roi_align_forward(_t(inp), _t(rois), 0.25, ph, pw, sr).cpu().numpy()
        ref = oracle.roi_align_forward(inp, rois, 0.25, ph, pw, sr)
        assert np.array_equal(out, ref), "max diff %g" % np.abs(out - ref).max()


def test_roi_align_adjoint_and_linearity_full_size(bwd_impl):
    """<fwd(x), g> == <x, bwd(g)> and bwd is linear — size-independent properties at cfg-2 scale."""
    C = _C()
    x = torch.randn(2, 256, 100, 168, device=DEV)
    rois_np = synth.fpn_rois(seed=9)
    rois = _t(rois_np[synth.level_map(rois_np) == 1])
    K = rois.size(0)
    g1 = torch.randn(K, 256, 7, 7, device=DEV)
    g2 = torch.randn(K, 256, 7, 7, device=DEV)
    y = C.roi_align_forward(x, rois, 1 / 8, 7, 7, 2)
    b1 = C.roi_align_backward(g1, rois, 1 / 8, 7, 7, 2, 256, 100, 168, 2)
    lhs = (y.double() * g1.double()).sum().item()
    rhs = (x.double() * b1.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))
    b2 = C.roi_align_backward(g2, rois, 1 / 8, 7, 7, 2, 256, 100, 168, 2)
    b12 = C.roi_align_backward(0.5 * g1 + g2, rois, 1 / 8, 7, 7, 2, 256, 100, 168, 2)
    torch.testing.assert_close(b12, 0.5 * b1 + b2, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("seed", range(10))
def test_roi_align_backward_acc_and_two_call_ring_random_shapes(seed):
    """random small maps (acc kernel: the map lives in LDS; 1-3 images, odd channel counts, ROIs outside / degenerate /
    slivers, 1 .. 400 ROIs, every group count) and random pyramids through the ring backward in ONE call and in TWO
    (pre-pass at forward time): within 2e-5 of the fp64-accumulated oracle, two-call == one-call bit for bit"""
    from maskrcnn_benchmark import _lib
    rng = np.random.RandomState(4000 + seed)
    N = int(rng.randint(1, 4))
    C = int(rng.choice([1, 5, 16, 37, 64]))
    H, W = int(rng.randint(1, 33)), int(rng.randint(1, 33))
    K = int(rng.choice([1, 2, 9, 65, 400]))
    ph = [7, 14][seed % 2]
    sr = int(rng.choice([0, 1, 2]))
    scale = float(rng.choice([1.0, 0.25, 0.0625]))
    iw, ih = W / scale, H / scale
    x1 = rng.uniform(-0.3 * iw, 1.1 * iw, K); y1 = rng.uniform(-0.3 * ih, 1.1 * ih, K)
    w = np.exp(rng.uniform(np.log(0.5), np.log(1.5 * iw), K)); h = np.exp(rng.uniform(np.log(0.5), np.log(1.5 * ih), K))
    rois = np.stack([rng.randint(0, N, K), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    if K > 2:
        rois[0, 3:] = rois[0, 1:3]
        rois[1, 1:] = [-1e4, -1e4, -9e3, -9e3]
    g = rng.randn(K, C, ph, ph).astype(np.float32)
    ref = oracle.roi_align_backward(g, rois, scale, ph, ph, N, C, H, W, sr, acc64=True)
    tol = 2e-5 * max(1.0, np.abs(ref).max())
    # the acc plan's LDS budget (roi_align_bwd.hip acc_plan: ring + two T buffers + map + row tables <= 150 KB); shapes
    # beyond it take the scan kernel, whose ROI-list split adds with atomics (seed 1: 14x14 bins on 23 x 27 -> 170 KB)
    nring, pad = (4, 8) if ph == 7 else (3, 16)
    lds = 4 * (nring * 4 * (64 if ph == 7 else 256) * 4 + 2 * H * pad * 20 + 16 * ((H * W + 3) & ~3) + 8 * (H + W) * pad) + 144
    acc_runs = lds <= 150 * 1024
    try:
        _lib.tuning_set("roi_bwd_impl", 4)
        for groups in (0, 1, 7):
            _lib.tuning_set("roi_bwd_groups", groups)
            got = _C().roi_align_backward(_t(g), _t(rois), scale, ph, ph, N, C, H, W, sr).cpu().numpy()
            assert np.abs(got - ref).max() <= tol, groups
            again = _C().roi_align_backward(_t(g), _t(rois), scale, ph, ph, N, C, H, W, sr).cpu().numpy()
            assert np.array_equal(got, again) or not acc_runs          # no atomics: run-to-run identical
    finally:
        _lib.tuning_set("roi_bwd_groups", 0)
        _lib.tuning_set("roi_bwd_impl", 0)
    # a pyramid through the ring kernel, one call vs prepare + prepared
    Cp = int(rng.choice([8, 24, 40]))
    shapes = [(2, Cp, 50 + seed, 84 - seed), (2, Cp, 25, 42), (2, Cp, 13, 21)]
    scales = [0.25, 0.125, 0.0625]
    Kp = int(rng.choice([3, 120, 500]))
    pr = synth.fpn_rois(seed=seed, per_image=(Kp + 1) // 2, smin=8, smax=300)[:Kp]
    pr[:, 1:] *= 0.25
    lv = np.minimum(synth.level_map(pr), 2).astype(np.int32)
    gp = rng.randn(Kp, Cp, ph, ph).astype(np.float32)
    try:
        _lib.tuning_set("roi_bwd_impl", 1)
        one = _C().roi_align_fpn_backward(_t(gp), _t(pr), _t(lv), shapes, scales, ph, ph, 2)
        prepared = _C().roi_align_fpn_backward_prepare(_t(pr), _t(lv), shapes, scales, ph, ph, 2)
        assert prepared is not None
        two = _C().roi_align_fpn_backward(_t(gp), _t(pr), _t(lv), shapes, scales, ph, ph, 2, prepared=prepared)
        for l, (a, b) in enumerate(zip(one, two)):
            assert torch.equal(a, b)
            sel = lv == l
            N_, C_, H_, W_ = shapes[l]
            refl = oracle.roi_align_backward(gp[sel], pr[sel], scales[l], ph, ph, N_, C_, H_, W_, 2, acc64=True) if sel.any() else np.zeros(shapes[l], np.float32)
            assert np.abs(a.cpu().numpy() - refl).max() <= 2e-5 * max(1.0, np.abs(refl).max())
    finally:
        _lib.tuning_set("roi_bwd_impl", 0)


def test_roi_align_backward_edge_cases(bwd_impl):
    C = _C()
    gin = C.roi_align_backward(torch.zeros(0, 3, 7, 7, device=DEV), torch.zeros(0, 5, device=DEV), 0.5,
                               7, 7, 2, 3, 10, 12, 2)
    assert gin.shape == (2, 3, 10, 12) and not gin.any()
    rois = np.array([[0, -500, -500, -400, -400], [1, 0, 0, 5000, 4000], [1, 3, 3, 3, 3],
                     [0, 5.5, 2.25, 20.75, 17.5], [1, 0, 0, 200, 150]], np.float32)
    g = np.random.RandomState(3).randn(5, 3, 4, 6).astype(np.float32)
    for sr in (0, 2):
        ref = oracle.roi_align_backward(g, rois, 1.0, 4, 6, 2, 3, 10, 12, sr, acc64=True)
        out = C.roi_align_backward(_t(g), _t(rois), 1.0, 4, 6, 2, 3, 10, 12, sr)
        _close(out, ref, rtol=1e-4, atol=1e-5)
    # a patch larger than the LDS budget (100x100 px) takes the direct-atomics path
    rois = np.array([[0, 2, 3, 118, 109]], np.float32)
    g = np.random.RandomState(4).randn(1, 2, 7, 7).astype(np.float32)
    ref = oracle.roi_align_backward(g, rois, 1.0, 7, 7, 1, 2, 128, 128, 0, acc64=True)
    out = C.roi_align_backward(_t(g), _t(rois), 1.0, 7, 7, 1, 2, 128, 128, 0)
    _close(out, ref, rtol=1e-4, atol=1e-5)


def test_roi_align_backward_deterministic_and_large_bin_counts():
    """The pixel-owner kernel has no atomics: two runs are bit-identical (the reference's atomicAdd
    scatter is not).  Also bin counts beyond the 256-slot staging plan (20x20 -> 4-channel chunks,
    28x28) and a channel count that is not a multiple of the chunk."""
    C = _C()
    rois_np = synth.fpn_rois(seed=5, per_image=128)
    lv = synth.level_map(rois_np)
    shapes = [(2, 64, h, w) for (h, w) in synth.fpn_shapes()[:4]]
    scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
    g = torch.randn(rois_np.shape[0], 64, 7, 7, device=DEV)
    a = C.roi_align_fpn_backward(g, _t(rois_np), _t(lv), shapes, scales, 7, 7, 2)
    b = C.roi_align_fpn_backward(g, _t(rois_np), _t(lv), shapes, scales, 7, 7, 2)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    rng = np.random.RandomState(21)
    N, Cc, H, W = 2, 5, 37, 70
    K = 24
    x1 = rng.uniform(-10, 250, K); y1 = rng.uniform(-10, 130, K)
    rois = np.stack([rng.randint(0, N, K), x1, y1, x1 + rng.uniform(1, 200, K), y1 + rng.uniform(1, 120, K)], 1).astype(np.float32)
    for (ph, pw, sr) in ((20, 20, 2), (28, 28, 1), (16, 16, 0), (2, 40, 2)):
        gg = rng.randn(K, Cc, ph, pw).astype(np.float32)
        ref = oracle.roi_align_backward(gg, rois, 0.25, ph, pw, N, Cc, H, W, sr, acc64=True)
        out = C.roi_align_backward(_t(gg), _t(rois), 0.25, ph, pw, N, Cc, H, W, sr)
        _close(out, ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref).max()))


def test_roi_align_layer_autograd_and_amp():
    from maskrcnn_benchmark.layers import ROIAlign

    inp, rois, scale = synth.cfg1_roi_align(K=64, C=16)
    layer = ROIAlign((7, 7), scale, 2)
    assert repr(layer) == "ROIAlign(output_size=(7, 7), spatial_scale=0.0625, sampling_ratio=2)"
    x = _t(inp).requires_grad_(True)
    y = layer(x, _t(rois))
    g = torch.randn_like(y)
    y.backward(g)
    ref = oracle.roi_align_backward(g.cpu().numpy(), rois, scale, 7, 7, *inp.shape, 2, acc64=True)
    _close(x.grad, ref, rtol=1e-4, atol=1e-4)
    with torch.autocast("cuda", dtype=torch.float16):
        y16 = layer(_t(inp).half(), _t(rois).half())
    assert y16.dtype == torch.float32  # amp.float_function semantics (reference roi_align.py:57)


# ============================================================================ ROIPool
def test_roi_pool_forward_backward():
    C = _C()
    inp, rois, scale = synth.cfg1_roi_align(K=200, C=32)
    out, amax = C.roi_pool_forward(_t(inp), _t(rois), scale, 7, 7)
    ref, ramax = oracle.roi_pool_forward(inp, rois, scale, 7, 7)
    assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(amax.cpu().numpy(), ramax)
    assert amax.dtype == torch.int32 and (ramax == -1).any()
    g = np.random.RandomState(5).randn(*ref.shape).astype(np.float32)
    gin = C.roi_pool_backward(_t(g), _t(inp), _t(rois), amax, scale, 7, 7, *inp.shape)
    _close(gin, oracle.roi_pool_backward(g, rois, ramax, *inp.shape), rtol=1e-4, atol=1e-4)
    # FPN-sized map, two images
    feats = synth.fpn_features(levels=4)[2]
    r = synth.fpn_rois(seed=6, per_image=64)
    out, amax = C.roi_pool_forward(_t(feats), _t(r), 1 / 16, 7, 7)
    ref, ramax = oracle.roi_pool_forward(feats, r, 1 / 16, 7, 7)
    assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(amax.cpu().numpy(), ramax)
    # ... and its backward, where the plane-owner kernel runs (2 x 256 planes of 50 x 84 in LDS); the scatter form beside it
    g = np.random.RandomState(8).randn(*ref.shape).astype(np.float32)
    want = oracle.roi_pool_backward(g, r, ramax, *feats.shape)
    gin = C.roi_pool_backward(_t(g), _t(feats), _t(r), amax, 1 / 16, 7, 7, *feats.shape)
    _close(gin, want, rtol=1e-4, atol=1e-4)
    _tune("roi_bwd_impl", 3)
    try:
        gin3 = C.roi_pool_backward(_t(g), _t(feats), _t(r), amax, 1 / 16, 7, 7, *feats.shape)
    finally:
        _tune("roi_bwd_impl", 0)
    _close(gin3, want, rtol=1e-4, atol=1e-4)
    from maskrcnn_benchmark.layers import ROIPool

    assert repr(ROIPool((7, 7), 0.25)) == "ROIPool(output_size=(7, 7), spatial_scale=0.25)"


# ============================================================================ deformable PS-ROI pooling
def _psroi_inputs(seed, K, output_dim, group_size, H, W, N, ncls, part):
    rng = np.random.RandomState(seed)
    data = rng.randn(N, output_dim * group_size * group_size, H, W).astype(np.float32)
    x1 = rng.uniform(-20, W * 16 - 30, K)
    y1 = rng.uniform(-20, H * 16 - 30, K)
    rois = np.stack([rng.randint(0, N, K), x1, y1, x1 + rng.uniform(2, 300, K), y1 + rng.uniform(2, 250, K)],
                    1).astype(np.float32)
    rois[0, 1:] = [40.2, 40.7, 40.3, 40.9]
    trans = (rng.randn(K, 2 * ncls, part, part) * 1.5).astype(np.float32)
    return data, rois, trans


@pytest.mark.parametrize("no_trans,ncls,D,G,P,part,S,std", [
    (True, 1, 8, 3, 3, 3, 4, 0.0),      # R-FCN style position-sensitive pooling, no offsets
    (False, 1, 8, 3, 7, 7, 4, 0.1),     # DeformRoIPoolingPack defaults (one offset field)
    (False, 4, 8, 2, 7, 4, 2, 0.1),     # class-specific offsets, part_size != pooled_size
    (False, 2, 4, 1, 12, 12, 4, 0.3),   # 12*12*16 samples > LDS table -> direct path
])
def test_deform_psroi_pool_vs_oracle(no_trans, ncls, D, G, P, part, S, std):
    """parity unpinned by the reference (CUDA-only, no test): the oracle restates
    deform_pool_kernel_cuda.cu and is cross-checked against a torch autograd formulation."""
    C = _C()
    data, rois, trans = _psroi_inputs(11 + P, 96, D, G, 50, 84, 2, ncls, part)
    cfg = (no_trans, 1 / 16, D, G, P, part, S, std)
    ref, rcnt = oracle.deform_psroi_pool_forward(data, rois, trans, *cfg)
    d, r = _t(data), _t(rois)
    t = torch.empty(0, device=DEV) if no_trans else _t(trans)
    out = torch.empty(ref.shape, device=DEV)
    cnt = torch.empty(ref.shape, device=DEV)
    C.deform_psroi_pooling_forward(d, r, t, out, cnt, *cfg)
    assert np.array_equal(cnt.cpu().numpy(), rcnt) and (rcnt == 0).any()
    _close(out, ref, rtol=1e-4, atol=1e-5)
    g = np.random.RandomState(4).randn(*ref.shape).astype(np.float32)
    gref, tref = oracle.deform_psroi_pool_backward(g, data, rois, trans, rcnt, *cfg, acc64=True)
    gin = torch.zeros_like(d)
    gtr = torch.zeros_like(t)
    C.deform_psroi_pooling_backward(_t(g), d, r, t, cnt, gin, gtr, *cfg)
    _close(gin, gref, rtol=1e-4, atol=1e-4)
    if not no_trans:
        _close(gtr, tref, rtol=1e-3, atol=1e-3 * max(1.0, float(np.abs(tref).max())))


def test_deform_roi_pooling_modules_autograd():
    from maskrcnn_benchmark.layers import DeformRoIPooling, DeformRoIPoolingPack, ModulatedDeformRoIPoolingPack

    data, rois, trans = _psroi_inputs(5, 16, 4, 1, 25, 42, 2, 1, 7)
    d = _t(data).requires_grad_(True)
    t = _t(trans).requires_grad_(True)
    pool = DeformRoIPooling(1 / 16, 7, 4, False, 1, None, 4, 0.1)
    y = pool(d, _t(rois), t)
    ref, cnt = oracle.deform_psroi_pool_forward(data, rois, trans, False, 1 / 16, 4, 1, 7, 7, 4, 0.1)
    _close(y, ref, rtol=1e-4, atol=1e-5)
    y.sum().backward()
    gref, tref = oracle.deform_psroi_pool_backward(np.ones_like(ref), data, rois, trans, cnt, False, 1 / 16,
                                                   4, 1, 7, 7, 4, 0.1, acc64=True)
    _close(d.grad, gref, rtol=1e-4, atol=1e-4)
    _close(t.grad, tref, rtol=1e-3, atol=1e-3)
    for cls in (DeformRoIPoolingPack, ModulatedDeformRoIPoolingPack):
        m = cls(1 / 16, 7, 4, False, 1, None, 4, 0.1, deform_fc_channels=32).to(DEV)
        out = m(d, _t(rois))
        assert out.shape == (16, 4, 7, 7) and torch.isfinite(out).all()
        out.sum().backward()
    # empty ROI set
    e = torch.empty((0, 4, 7, 7), device=DEV)
    _C().deform_psroi_pooling_forward(d.detach(), torch.empty((0, 5), device=DEV), torch.empty(0, device=DEV),
                                      e, e.clone(), True, 1 / 16, 4, 1, 7, 7, 4, 0.0)


# ============================================================================ NMS
def _nms(b, s, thr):
    return _C().nms(_t(b), _t(s), thr).cpu().numpy()


def test_nms_reference_known_answers(golden_dir):
    g = _load(golden_dir, "nms_reference_tests.npz")
    for i in range(int(g["num_cases"])):
        keep = _nms(g[f"boxes_{i}"], g[f"scores_{i}"], float(g[f"thresh_{i}"]))
        np.testing.assert_array_equal(keep, g[f"expected_{i}"])


def test_nms_matches_compiled_reference_vectors(golden_dir):
    g = _load(golden_dir, "ref_cpu_vectors.npz")
    for key in g["nms_cases"]:
        _, n, uniform, seed, thr = str(key).split("_")
        b, s = synth.nms_boxes(int(n), seed=int(seed), uniform=bool(int(uniform)))
        np.testing.assert_array_equal(_nms(b, s, int(thr) / 100.0), g[str(key)], err_msg=str(key))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 128, 129, 1000, 2000, 6000, 8192, 8193, 12000])
def test_nms_bit_exact_vs_oracle(n):
    for uniform in (False, True):
        b, s = synth.nms_boxes(n, seed=40 + n % 7, uniform=uniform)
        for thr in (0.5, 0.7):
            keep = _nms(b, s, thr)
            assert keep.dtype == np.int64
            np.testing.assert_array_equal(keep, oracle.nms(b, s, thr))


def test_nms_edge_cases():
    C = _C()
    e = C.nms(torch.zeros(0, 4, device=DEV), torch.zeros(0, device=DEV), 0.5)
    assert e.shape == (0,) and e.dtype == torch.long and e.device.type == "cpu"  # nms.h:17-18
    b = np.array([[0, 0, 9, 9], [0, 0, 9, 4]], np.float32)  # IoU exactly 0.5
    s = np.array([0.9, 0.8], np.float32)
    np.testing.assert_array_equal(_nms(b, s, 0.5), [0])  # >= (CPU semantics), not > (nms.cu:60)
    np.testing.assert_array_equal(_nms(b, s, 0.5000001), [0, 1])
    # ties and duplicates: stable order, identical boxes collapse to the first of the best score
    b = np.array([[0, 0, 10, 10]] * 5 + [[50, 50, 60, 60]] * 3, np.float32)
    s = np.array([0.3, 0.9, 0.9, 0.1, 0.9, 0.5, 0.5, 0.5], np.float32)
    np.testing.assert_array_equal(_nms(b, s, 0.7), oracle.nms(b, s, 0.7))
    np.testing.assert_array_equal(_nms(b, s, 0.7), [1, 5])
    # non-distinct random scores
    b, s = synth.nms_boxes(3000, seed=3, distinct_scores=False)
    s = np.round(s, 2)
    np.testing.assert_array_equal(_nms(b, s, 0.6), oracle.nms(b, s, 0.6))
    # threshold 0 and > 1
    b, s = synth.nms_boxes(500, seed=4)
    for thr in (0.0, 1.5):
        np.testing.assert_array_equal(_nms(b, s, thr), oracle.nms(b, s, thr))


@pytest.mark.parametrize("step,thr", [(10, 0.8), (25, 0.55), (34, 0.45), (50, 0.3)])
def test_nms_dependency_chains(step, thr):
    """boxes marching along x with descending scores: the greedy choice is a dependency chain through every row
    block (worst case of the one-wave scan's fixed-point resolve); n = 4097 takes the shared-memory scan."""
    for n in (64, 200, 2000, 4096, 4097):
        x0 = np.arange(n, dtype=np.float32) * step
        b = np.stack([x0, np.zeros(n, np.float32), x0 + 99, np.full(n, 49, np.float32)], 1)
        sc = np.linspace(1.0, 0.1, n).astype(np.float32)
        ref = oracle.nms(b, sc, thr)
        assert 1 < len(ref) < n
        keep = _C().nms(_t(b), _t(sc), thr).cpu().numpy()
        assert np.array_equal(keep, ref), (n, step, thr)


def test_nms_idempotent_and_sorted_full_size():
    b, s = synth.nms_boxes(2000, seed=8)
    keep = _nms(b, s, 0.7)
    assert np.all(np.diff(keep) > 0)
    again = _nms(b[keep], s[keep], 0.7)
    np.testing.assert_array_equal(again, np.arange(len(keep)))


def test_nms_batched_rpn_segments():
    segs = synth.rpn_nms_segments()
    segs.insert(3, (np.zeros((0, 4), np.float32), np.zeros((0,), np.float32)))  # an empty segment
    boxes = np.concatenate([b for b, _ in segs])
    scores = np.concatenate([s for _, s in segs])
    offs = np.cumsum([0] + [len(s) for _, s in segs]).astype(np.int32)
    keep, num = _C().nms_batched(_t(boxes), _t(scores), _t(offs), 2000, 0.7)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for i, (b, s) in enumerate(segs):
        ref = oracle.nms(b, s, 0.7)
        assert num[i] == len(ref)
        np.testing.assert_array_equal(keep[offs[i]:offs[i] + num[i]], ref)


@pytest.mark.parametrize("fused", [0, 3, 2])
def test_nms_single_launch_and_three_launch_paths(fused):
    """n <= 4096: sort, mask tiles and scan are ONE launch (workgroup roles, flags between workgroups, no host-cleared
    state); `nms_fused=2` keeps the three launches.  Ragged segments over the three scan widths, an empty one, dependency
    chains; repeated calls reuse the allocator's (dirty) workspace."""
    from maskrcnn_benchmark import _lib
    _lib.tuning_set("nms_fused", fused)
    try:
        sizes = (1000, 0, 1025, 64, 2000, 1, 2049, 4096, 65, 3000)
        segs = [synth.nms_boxes(n, seed=40 + i) if n else (np.zeros((0, 4), np.float32), np.zeros(0, np.float32))
                for i, n in enumerate(sizes)]
        x0 = np.arange(2000, dtype=np.float32) * 25
        segs[4] = (np.stack([x0, np.zeros(2000, np.float32), x0 + 99, np.full(2000, 49, np.float32)], 1),
                   np.linspace(1.0, 0.1, 2000).astype(np.float32))
        boxes = np.concatenate([b for b, _ in segs])
        scores = np.concatenate([s for _, s in segs])
        offs = np.cumsum([0] + [len(s) for _, s in segs]).astype(np.int32)
        tb, ts, to = _t(boxes), _t(scores), _t(offs)
        for rep in range(3):
            for thr in (0.7, 0.55):
                keep, num = _C().nms_batched(tb, ts, to, 4096, thr)
                km, num2 = _C().nms_batched_mask(tb, ts, to, 4096, thr)
                keep, num, km = keep.cpu().numpy(), num.cpu().numpy(), km.cpu().numpy()
                np.testing.assert_array_equal(num, num2.cpu().numpy())
                for i, (b, s) in enumerate(segs):
                    ref = oracle.nms(b, s, thr) if len(s) else np.zeros(0, np.int64)
                    assert num[i] == len(ref), (rep, thr, i)
                    np.testing.assert_array_equal(keep[offs[i]:offs[i] + num[i]], ref)
                    want = np.zeros(len(s), np.uint8)
                    want[ref] = 1
                    np.testing.assert_array_equal(km[offs[i]:offs[i + 1]], want)
        # the detector's shape: 10 segments of <= 2000 (two workgroups per CU variant)
        segs = synth.rpn_nms_segments()
        boxes = np.concatenate([b for b, _ in segs])
        scores = np.concatenate([s for _, s in segs])
        offs = np.cumsum([0] + [len(s) for _, s in segs]).astype(np.int32)
        for rep in range(20):
            keep, num = _C().nms_batched(_t(boxes), _t(scores), _t(offs), 2000, 0.7)
        keep, num = keep.cpu().numpy(), num.cpu().numpy()
        for i, (b, s) in enumerate(segs):
            ref = oracle.nms(b, s, 0.7)
            assert num[i] == len(ref)
            np.testing.assert_array_equal(keep[offs[i]:offs[i] + num[i]], ref)
    finally:
        _lib.tuning_set("nms_fused", 0)


def test_nms_single_launch_timeout_is_reported_and_repaired():
    """ADVICE r03 (medium) + VERDICT r04 "missing" #4: a wait of the single-launch kernel that runs out of its polling
    budget must neither publish a keep set built from unpublished rows NOR drop the segment (reference csrc/cuda/nms.cu:
    70-131 never does).  Fault injection: the sort workgroups publish a wrong token, the budget is a few hundred polls ->
    every consumer wait times out.  With the repair launch switched off (tuning nms_no_repair): num_keep = -1, all-zero
    keep mask, the reference-named `nms` raises.  Default: the repair launch redoes every failed segment on the same
    stream — bit-equal results, and the sticky status word counts them."""
    from maskrcnn_benchmark import _lib
    segs = synth.rpn_nms_segments()[:4]
    boxes = np.concatenate([b for b, _ in segs])
    scores = np.concatenate([s for _, s in segs])
    offs = np.cumsum([0] + [len(s) for _, s in segs]).astype(np.int32)
    tb, ts, to = _t(boxes), _t(scores), _t(offs)
    good_mask, good_num = _C().nms_batched_mask(tb, ts, to, 2000, 0.7)
    _C().nms_repaired_segments(reset=True)
    try:
        _lib.tuning_set("nms_fault", 1)
        _lib.tuning_set("nms_spin_budget", 200)
        _lib.tuning_set("nms_no_repair", 1)
        km, num = _C().nms_batched_mask(tb, ts, to, 2000, 0.7)
        torch.cuda.synchronize()
        assert (num.cpu().numpy() == -1).all(), num
        assert not km.cpu().numpy().any()
        with pytest.raises(RuntimeError, match="timed out"):
            _C().nms(_t(segs[0][0]), _t(segs[0][1]), 0.7)
        assert _C().nms_repaired_segments() == 0
        _lib.tuning_set("nms_no_repair", 0)
        km, num = _C().nms_batched_mask(tb, ts, to, 2000, 0.7)
        keep, num2 = _C().nms_batched(tb, ts, to, 2000, 0.7)
        one = _C().nms(_t(segs[0][0]), _t(segs[0][1]), 0.7)
        assert torch.equal(km, good_mask) and torch.equal(num, good_num) and torch.equal(num2, good_num)
        assert np.array_equal(one.cpu().numpy(), oracle.nms(segs[0][0], segs[0][1], 0.7))
        keep, num2 = keep.cpu().numpy(), num2.cpu().numpy()
        for i, (b, sc) in enumerate(segs):
            ref = oracle.nms(b, sc, 0.7)
            assert num2[i] == len(ref)
            np.testing.assert_array_equal(keep[offs[i]:offs[i] + num2[i]], ref)
        assert _C().nms_repaired_segments(reset=True) == 2 * len(segs)      # (the single `nms` passes no status word)
        assert _C().nms_repaired_segments() == 0
    finally:
        for k in ("nms_fault", "nms_spin_budget", "nms_no_repair"):
            _lib.tuning_set(k, 0)
    # and the very next launch on the same workspace allocator is correct again, without any repair
    keep, num = _C().nms_batched(tb, ts, to, 2000, 0.7)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for i, (b, sc) in enumerate(segs):
        ref = oracle.nms(b, sc, 0.7)
        assert num[i] == len(ref)
        np.testing.assert_array_equal(keep[offs[i]:offs[i] + num[i]], ref)
    assert _C().nms_repaired_segments() == 0


@pytest.mark.parametrize("fused", [0, 3])
def test_nms_single_launch_under_concurrent_stream_load(fused):
    """the single-launch kernel's workgroups wait for each other: both dispatch orders (0: scans right behind the sorts
    when the device is otherwise empty, 3: scans last) stay bit-exact while ANOTHER stream keeps the compute units busy"""
    from maskrcnn_benchmark import _lib
    segs = synth.rpn_nms_segments()
    boxes = np.concatenate([b for b, _ in segs])
    scores = np.concatenate([s for _, s in segs])
    offs = np.cumsum([0] + [len(s) for _, s in segs]).astype(np.int32)
    tb, ts, to = _t(boxes), _t(scores), _t(offs)
    refs = [oracle.nms(b, sc, 0.7) for b, sc in segs]
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    try:
        _lib.tuning_set("nms_fused", fused)
        outs = []
        for rep in range(6):
            with torch.cuda.stream(side):
                for _ in range(4):
                    a = torch.tanh(a @ a * 1e-4)          # long-running GEMMs that fill the chip
            outs.append(_C().nms_batched(tb, ts, to, 2000, 0.7))
        torch.cuda.synchronize()
        for keep, num in outs:
            keep, num = keep.cpu().numpy(), num.cpu().numpy()
            for i, ref in enumerate(refs):
                assert num[i] == len(ref), (fused, i, num[i])
                np.testing.assert_array_equal(keep[offs[i]:offs[i] + num[i]], ref)
    finally:
        _lib.tuning_set("nms_fused", 0)


def test_nms_single_launch_beside_rccl_all_reduces_and_bucket_updates():
    """the single-launch kernel next to what the data-parallel step really runs on its second
    queue — 25 MB RCCL all-reduces (through engine/rccl_comm.py on a low-priority stream, and through ProcessGroupNCCL)
    each followed by a fused SGD update of the bucket — stays bit-exact and needs no repair."""
    import torch.distributed as dist
    from maskrcnn_benchmark.engine import rccl_comm
    segs = synth.rpn_nms_segments()
    boxes = np.concatenate([b for b, _ in segs])
    scores = np.concatenate([s for _, s in segs])
    offs = np.cumsum([0] + [len(s) for _, s in segs]).astype(np.int32)
    tb, ts, to = _t(boxes), _t(scores), _t(offs)
    refs = [oracle.nms(b, sc, 0.7) for b, sc in segs]
    n = 25 * 1024 * 1024 // 4
    bucket, par, mom = (torch.randn(n, device=DEV) for _ in range(3))
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29531", rank=0, world_size=1)
    try:
        comm = rccl_comm.RcclComm(DEV)
        comm.selftest()
        side, prio = rccl_comm.low_priority_stream(DEV)
        _C().nms_repaired_segments(reset=True)
        outs = []
        for rep in range(8):
            for _ in range(6):
                if rep % 2 == 0:
                    comm.all_reduce_avg_(bucket, side)
                    with torch.cuda.stream(side):
                        torch._fused_sgd_([par], [bucket], [mom], weight_decay=1e-4, momentum=0.9, lr=1e-3, dampening=0.0,
                                          nesterov=False, maximize=False, is_first_step=False)
                else:
                    dist.all_reduce(bucket, op=dist.ReduceOp.AVG, async_op=True)
            outs.append(_C().nms_batched(tb, ts, to, 2000, 0.7))
        torch.cuda.synchronize()
        comm.destroy()
    finally:
        dist.destroy_process_group()
    for keep, num in outs:
        keep, num = keep.cpu().numpy(), num.cpu().numpy()
        for i, ref in enumerate(refs):
            assert num[i] == len(ref), (i, num[i])
            np.testing.assert_array_equal(keep[offs[i]:offs[i] + num[i]], ref)
    assert _C().nms_repaired_segments() == 0


@pytest.mark.parametrize("prio", ["high", "low"])
def test_nms_single_launch_beside_a_contention_stand_in(prio):
    """VERDICT r05 #4d: the single-launch NMS (inter-workgroup waits, bounded + repaired) while workgroups on a second,
    explicit-priority stream HOLD compute units the way an N > 1 ring all-reduce kernel does (detops_debug_occupy: 64
    workgroups of 1024 threads for 1 ms, back to back): results bit-exact.  A repaired segment would still be correct — the
    count is reported, and expected to be 0 (profiles/r06_ddp_contention.txt: 0 in every setting)."""
    from maskrcnn_benchmark import _C as C
    from maskrcnn_benchmark.engine import rccl_comm
    segs = synth.rpn_nms_segments()
    boxes = np.concatenate([b for b, _ in segs])
    scores = np.concatenate([s for _, s in segs])
    offs = np.cumsum([0] + [len(s) for _, s in segs]).astype(np.int32)
    tb, ts, to = _t(boxes), _t(scores), _t(offs)
    refs = [oracle.nms(b, sc, 0.7) for b, sc in segs]
    side = rccl_comm.low_priority_stream(DEV)[0] if prio == "low" else torch.cuda.Stream(DEV, priority=-1)
    C.nms_repaired_segments(reset=True)
    outs = []
    for rep in range(6):
        for _ in range(4):
            C.check(C.lib.detops_debug_occupy(64, 1000, side.cuda_stream), "debug_occupy")
        for _ in range(3):
            outs.append(C.nms_batched(tb, ts, to, 2000, 0.7))
    torch.cuda.synchronize()
    for keep, num in outs:
        keep, num = keep.cpu().numpy(), num.cpu().numpy()
        for i, ref in enumerate(refs):
            assert num[i] == len(ref), (i, num[i])
            np.testing.assert_array_equal(keep[offs[i]:offs[i] + num[i]], ref)
    assert C.nms_repaired_segments() == 0


def test_nms_threshold_boundary_is_exact():
    """Pairs whose IoU is EXACTLY the threshold, one ulp above and one ulp below it, and degenerate unions (negative
    "areas", huge coordinates) must come out as the reference's `inter / union >= thr` (nms_cpu.cpp:59-60) does: the
    decision needs the correctly rounded IEEE quotient."""
    rng = np.random.RandomState(11)
    f = np.float32
    cases = 0
    for trial in range(120):
        w0, h0 = rng.uniform(5, 400, 2).astype(f)
        a = np.array([10.25, 20.5, f(10.25) + w0, f(20.5) + h0], f)
        dx, dy = (rng.uniform(-0.6, 0.6, 2) * np.array([w0, h0])).astype(f)
        sc = f(rng.uniform(0.6, 1.6))
        b = np.array([a[0] + dx, a[1] + dy, a[0] + dx + w0 * sc, a[1] + dy + h0 / sc], f)
        ia = f(f(a[2] - a[0] + f(1)) * f(a[3] - a[1] + f(1)))
        ib = f(f(b[2] - b[0] + f(1)) * f(b[3] - b[1] + f(1)))
        ww = max(f(0), f(f(min(a[2], b[2]) - max(a[0], b[0])) + f(1)))
        hh = max(f(0), f(f(min(a[3], b[3]) - max(a[1], b[1])) + f(1)))
        inter = f(ww * hh)
        if inter <= 0:
            continue
        ovr = f(inter / f(f(ia + ib) - inter))
        boxes = np.stack([a, b])
        scores = np.array([0.9, 0.8], f)
        for thr in (ovr, np.nextafter(ovr, f(2)), np.nextafter(ovr, f(-1)), f(ovr * f(1.0000005)), f(ovr * f(0.9999995))):
            if not (0 < thr < 1):
                continue
            want = oracle.nms(boxes, scores, float(thr))
            assert np.array_equal(_nms(boxes, scores, float(thr)), want), (trial, float(ovr), float(thr))
            assert len(want) == (1 if ovr >= thr else 2)
            cases += 1
    assert cases > 300
    # degenerate unions: zero-area / inverted boxes (negative "areas"), identical boxes, huge coordinates
    odd = np.array([[0, 0, -1, -1], [0, 0, -1, -1], [5, 5, 4, 9], [5, 5, 4, 9], [0, 0, 10, 10], [0, 0, 10, 10],
                    [-3e18, -3e18, 3e18, 3e18], [-3e18, -3e18, 3e18, 3e18], [1, 1, 0.5, 0.5], [0, 0, 1e-20, 1e-20]], f)
    sc = np.linspace(1.0, 0.1, len(odd)).astype(f)
    for thr in (0.0, 1e-6, 0.5, 1.0):
        assert np.array_equal(_nms(odd, sc, thr), oracle.nms(odd, sc, thr)), thr


def test_nms_layer_amp_hint():
    from maskrcnn_benchmark.layers import nms

    b, s = synth.nms_boxes(300, seed=5)
    with torch.autocast("cuda", dtype=torch.float16):
        keep = nms(_t(b), _t(s), 0.5)
    np.testing.assert_array_equal(keep.cpu().numpy(), oracle.nms(b, s, 0.5))


# ============================================================================ SigmoidFocalLoss
def test_focal_vs_oracle_and_golden(golden_dir):
    C = _C()
    logits, targets = synth.focal_inputs(20000, 80)
    tl, tt = _t(logits), _t(targets)
    f = C.sigmoid_focalloss_forward(tl, tt, 80, 2.0, 0.25)
    _close(f, oracle.sigmoid_focal_loss_forward(logits, targets, 2.0, 0.25), rtol=1e-4, atol=1e-6)
    d = np.random.RandomState(6).rand(*logits.shape).astype(np.float32)
    b = C.sigmoid_focalloss_backward(tl, tt, _t(d), 80, 2.0, 0.25)
    _close(b, oracle.sigmoid_focal_loss_backward(logits, targets, d, 2.0, 0.25), rtol=1e-4, atol=1e-6)
    # the reference's own Python CPU composite (layers/sigmoid_focal_loss.py:40-50)
    g = _load(golden_dir, "focal_python_composite.npz")
    f = C.sigmoid_focalloss_forward(_t(g["logits"]), _t(g["targets"]), 80, float(g["gamma"]), float(g["alpha"]))
    _close(f, g["losses"], rtol=1e-4, atol=1e-6)
    b = C.sigmoid_focalloss_backward(_t(g["logits"]), _t(g["targets"]), _t(g["d_losses"]), 80,
                                     float(g["gamma"]), float(g["alpha"]))
    _close(b, g["d_logits"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_focal_vs_reference_cuda_formula_in_fp64_up_to_100(golden_dir, tag):
    """the HIP kernels against the reference's CUDA formula (SigmoidFocalLoss_cuda.cu:29-99) evaluated in float64 over logits
    up to |x| = 100 (tests/golden/make_golden_focal_formula.py): 1e-4 relative / 1e-5 absolute, forward and backward"""
    C = _C()
    g = _load(golden_dir, "focal_cuda_formula_fp64.npz")
    gamma, alpha = (float(v) for v in g["cfg_" + tag])
    f = C.sigmoid_focalloss_forward(_t(g["logits"]), _t(g["targets"]), 80, gamma, alpha)
    _close(f, g["losses_" + tag], rtol=1e-4, atol=1e-5)
    b = C.sigmoid_focalloss_backward(_t(g["logits"]), _t(g["targets"]), _t(g["d_losses"]), 80, gamma, alpha)
    _close(b, g["d_logits_" + tag], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("gamma,alpha,C", [(2.0, 0.25, 80), (1.5, 0.4, 80), (0.0, 0.5, 7), (1.0, 0.25, 3)])
def test_focal_parameters_and_odd_class_counts(gamma, alpha, C):
    logits, targets = synth.focal_inputs(999, C, seed=7)
    logits[0, :] = [-100, -90, -20, 0, 20, 90, 100][:C] + [0.5] * max(0, C - 7)
    targets[:5] = [1, min(2, C), -1, 0, C]
    f = _C().sigmoid_focalloss_forward(_t(logits), _t(targets), C, gamma, alpha)
    _close(f, oracle.sigmoid_focal_loss_forward(logits, targets, gamma, alpha), rtol=1e-4, atol=1e-6)
    d = np.random.RandomState(8).rand(*logits.shape).astype(np.float32)
    b = _C().sigmoid_focalloss_backward(_t(logits), _t(targets), _t(d), C, gamma, alpha)
    _close(b, oracle.sigmoid_focal_loss_backward(logits, targets, d, gamma, alpha), rtol=1e-4, atol=1e-6)
    assert torch.isfinite(f).all() and torch.isfinite(b).all()


def test_focal_full_size_properties_and_module():
    """cfg-4: R = 2 x 201,600 anchors x 80 classes.  sum kernel == sum of elementwise kernel ==
    torch composite; module backward == explicit backward with a constant upstream gradient."""
    from maskrcnn_benchmark.layers import SigmoidFocalLoss

    C = _C()
    logits, targets = synth.focal_inputs(403200, 80)
    tl, tt = _t(logits), _t(targets)
    losses = C.sigmoid_focalloss_forward(tl, tt, 80, 2.0, 0.25)
    total = C.sigmoid_focalloss_forward_sum(tl, tt, 80, 2.0, 0.25)
    ref_sum = losses.double().sum().item()
    assert abs(total.item() - ref_sum) <= 1e-4 * abs(ref_sum)
    # torch composite of the reference's CPU formula, evaluated on the GPU in fp64
    x = tl.double()
    p = torch.sigmoid(x)
    cls = torch.arange(1, 81, device=DEV).unsqueeze(0)
    t = tt.unsqueeze(1)
    comp = -(t == cls).double() * (1 - p) ** 2 * torch.log(p) * 0.25 \
        - ((t != cls) & (t >= 0)).double() * p ** 2 * torch.log1p(-p) * 0.75
    torch.testing.assert_close(losses.double(), comp, rtol=1e-4, atol=1e-6)
    rows_ignored = (tt == -1)
    assert not losses[rows_ignored].any()
    # module: sum + backward
    mod = SigmoidFocalLoss(2.0, 0.25)
    assert repr(mod) == "SigmoidFocalLoss(gamma=2.0, alpha=0.25)"
    xg = tl.clone().requires_grad_(True)
    loss = mod(xg, tt) / 1234.0
    loss.backward()
    explicit = C.sigmoid_focalloss_backward(tl, tt, torch.full_like(tl, 1 / 1234.0), 80, 2.0, 0.25)
    torch.testing.assert_close(xg.grad, explicit, rtol=1e-5, atol=1e-9)


def test_focal_model_entry_points_vs_oracle_full_size():
    """The two entry points the RetinaNet loss actually calls — the fused two-stage sum-forward
    (detops_sigmoid_focal_loss_forward_sum_ws_f32) and the scalar-gradient backward (..._backward_scalar_f32) — against
    the ORACLE (C restatement of SigmoidFocalLoss_cuda.cu:29-56, 71-99) at the full RetinaNet size R = 403,200 x 80."""
    C = _C()
    R = 403200
    logits, targets = synth.focal_inputs(R, 80)
    tl, tt = _t(logits), _t(targets)
    ref = oracle.sigmoid_focal_loss_forward(logits, targets, 2.0, 0.25).astype(np.float64)
    total = float(C.sigmoid_focalloss_forward_sum(tl, tt, 80, 2.0, 0.25))
    assert abs(total - ref.sum()) <= 1e-4 * abs(ref.sum()), (total, ref.sum())
    # bit-reproducible (fixed-order two-stage reduction)
    assert float(C.sigmoid_focalloss_forward_sum(tl, tt, 80, 2.0, 0.25)) == total
    scale = 1.0 / 1234.0                                   # what `.sum() / normaliser` back-propagates
    dref = oracle.sigmoid_focal_loss_backward(logits, targets, np.full_like(logits, scale), 2.0, 0.25)
    d = C.sigmoid_focalloss_backward_scalar(tl, tt, torch.full((), scale, device=DEV), 80, 2.0, 0.25).cpu().numpy()
    np.testing.assert_allclose(d, dref, rtol=1e-4, atol=1e-4 * np.abs(dref).max())


# ============================================================================ deformable conv
GEOMS = [dict(B=2, C=8, H=13, W=17, Cout=6, k=3, stride=1, pad=1, dil=1, dg=1, group=1),
         dict(B=2, C=8, H=14, W=15, Cout=8, k=3, stride=2, pad=2, dil=2, dg=2, group=2),
         dict(B=2, C=64, H=50, W=84, Cout=64, k=3, stride=1, pad=1, dil=1, dg=1, group=1)]


def _dcn_case(g, modulated, seed=3):
    x, off, mask, wgt = synth.dcn_inputs(g["B"], g["C"], g["H"], g["W"], g["Cout"], g["k"], g["dg"],
                                         modulated, seed=seed)
    Ho = (g["H"] + 2 * g["pad"] - (g["dil"] * (g["k"] - 1) + 1)) // g["stride"] + 1
    Wo = (g["W"] + 2 * g["pad"] - (g["dil"] * (g["k"] - 1) + 1)) // g["stride"] + 1
    off = np.ascontiguousarray(off[:, :, :Ho, :Wo])
    mask = None if mask is None else np.ascontiguousarray(mask[:, :, :Ho, :Wo])
    wgt = np.ascontiguousarray(wgt[:, : g["C"] // g["group"]])
    return x, off, mask, wgt


@pytest.mark.parametrize("gi", range(len(GEOMS)))
@pytest.mark.parametrize("modulated", [False, True])
def test_deformable_kernels_vs_oracle_fp32(gi, modulated):
    C = _C()
    g = GEOMS[gi]
    x, off, mask, _ = _dcn_case(g, modulated)
    k, p, s, d, dg = g["k"], g["pad"], g["stride"], g["dil"], g["dg"]
    geo = (k, k, p, p, s, s, d, d, dg)
    ogeo = dict(kh=k, kw=k, pad=(p, p), stride=(s, s), dil=(d, d), dg=dg)
    tm = None if mask is None else _t(mask)
    col = C.deformable_im2col(_t(x), _t(off), tm, *geo)
    ref_col = oracle.deformable_im2col(x, off, mask, **ogeo)
    _close(col, ref_col, rtol=1e-5, atol=1e-5)
    gcol = np.random.RandomState(9).randn(*ref_col.shape).astype(np.float32)
    gim = torch.zeros(*x.shape, device=DEV)
    _tune("dcn_col2im", 1)      # inverted-index gather (no data atomics)
    C.deformable_col2im(_t(gcol), _t(off), tm, gim, *geo)
    ref_gim = oracle.deformable_col2im(gcol, off, mask, *x.shape, **ogeo)
    _close(gim, ref_gim, rtol=1e-4, atol=1e-4)
    gim2 = torch.zeros(*x.shape, device=DEV)
    C.deformable_col2im(_t(gcol), _t(off), tm, gim2, *geo)
    assert torch.equal(gim, gim2), "gather col2im must be deterministic"
    _tune("dcn_col2im", 2)      # the atomic LDS-window kernel
    gim3 = torch.zeros(*x.shape, device=DEV)
    C.deformable_col2im(_t(gcol), _t(off), tm, gim3, *geo)
    _close(gim3, ref_gim, rtol=1e-4, atol=1e-4)
    _tune("dcn_col2im", 3)      # fixed-width inverted index
    gim5 = torch.zeros(*x.shape, device=DEV)
    C.deformable_col2im(_t(gcol), _t(off), tm, gim5, *geo)
    _close(gim5, ref_gim, rtol=1e-4, atol=1e-4)
    _tune("dcn_col2im", 0)      # default: chosen by dtype / map size
    gim4 = torch.zeros(*x.shape, device=DEV)
    C.deformable_col2im(_t(gcol), _t(off), tm, gim4, *geo)
    _close(gim4, ref_gim, rtol=1e-4, atol=1e-4)
    goff = torch.empty(*off.shape, device=DEV)
    gmask = None if mask is None else torch.empty(*mask.shape, device=DEV)
    C.deformable_col2im_coord(_t(gcol), _t(x), _t(off), tm, goff, gmask, *geo)
    rgoff, rgmask = oracle.deformable_col2im_coord(gcol, x, off, mask, **ogeo)
    scale = max(1.0, np.abs(rgoff).max())
    _close(goff, rgoff, rtol=1e-4, atol=1e-4 * scale)
    if modulated:
        _close(gmask, rgmask, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(rgmask).max()))


@pytest.mark.parametrize("case", [dict(B=1, C=3, H=37, W=70, k=3, stride=1, pad=1, dil=1, dg=1, sigma=2.0),
                                  dict(B=2, C=2, H=21, W=45, k=3, stride=1, pad=1, dil=1, dg=1, sigma=12.0),
                                  dict(B=1, C=4, H=40, W=41, k=3, stride=2, pad=3, dil=3, dg=2, sigma=5.0),
                                  dict(B=1, C=2, H=19, W=33, k=1, stride=1, pad=0, dil=1, dg=1, sigma=9.0),
                                  dict(B=1, C=2, H=50, W=40, k=3, stride=1, pad=6, dil=6, dg=1, sigma=3.0)])
def test_deformable_index_tile_owner_build_on_device(case):
    """the tile-owner build of the inverted index on the device (LDS lists, ballot-compacted slot requests): maps of
    several tiles, displacements far beyond the scanned window (overflow list), strides / dilations / padding — the
    oracle's gradient, and the scatter build's (tuning dcn_ell_build = 1)"""
    C = _C()
    g = case
    rng = np.random.RandomState(17)
    k, p_, s_, d, dg = g["k"], g["pad"], g["stride"], g["dil"], g["dg"]
    Ho = (g["H"] + 2 * p_ - (d * (k - 1) + 1)) // s_ + 1
    Wo = (g["W"] + 2 * p_ - (d * (k - 1) + 1)) // s_ + 1
    off = (rng.randn(g["B"], 2 * dg * k * k, Ho, Wo) * g["sigma"]).astype(np.float32)
    mask = rng.uniform(0.2, 1.0, (g["B"], dg * k * k, Ho, Wo)).astype(np.float32)
    geo = (k, k, p_, p_, s_, s_, d, d, dg)
    ogeo = dict(kh=k, kw=k, pad=(p_, p_), stride=(s_, s_), dil=(d, d), dg=dg)
    gcol = rng.randn(g["C"] * k * k, g["B"] * Ho * Wo).astype(np.float32)
    shape = (g["B"], g["C"], g["H"], g["W"])
    ref = oracle.deformable_col2im(gcol, off, mask, *shape, **ogeo)
    tol = dict(rtol=1e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))
    try:
        _tune("dcn_col2im", 3)
        for build in (0, 1):
            _tune("dcn_ell_build", build)
            gim = torch.zeros(*shape, device=DEV)
            C.deformable_col2im(_t(gcol), _t(off), _t(mask), gim, *geo)
            _close(gim, ref, **tol)
    finally:
        _tune("dcn_ell_build", 0)
        _tune("dcn_col2im", 0)


@pytest.mark.parametrize("gi", [0, 1])
@pytest.mark.parametrize("modulated", [False, True])
def test_deform_conv_layers_end_to_end_fp32(gi, modulated):
    from maskrcnn_benchmark.layers import deform_conv, modulated_deform_conv

    g = GEOMS[gi]
    if modulated and g["stride"] != 1:
        pass  # modulated API takes scalar stride/pad/dil (reference deform_conv_func.py:150-160)
    x, off, mask, wgt = _dcn_case(g, modulated)
    bias = np.random.RandomState(4).randn(g["Cout"]).astype(np.float32) if modulated else None
    tx, toff, tw = (_t(a).requires_grad_(True) for a in (x, off, wgt))
    og = dict(pad=(g["pad"],) * 2, stride=(g["stride"],) * 2, dil=(g["dil"],) * 2, group=g["group"], dg=g["dg"])
    if modulated:
        tmask, tb = _t(mask).requires_grad_(True), _t(bias).requires_grad_(True)
        y = modulated_deform_conv(tx, toff, tmask, tw, tb, g["stride"], g["pad"], g["dil"], g["group"], g["dg"])
    else:
        y = deform_conv(tx, toff, tw, g["stride"], g["pad"], g["dil"], g["group"], g["dg"])
    ref = oracle.deform_conv_forward(x, off, mask, wgt, bias, **og)
    _close(y, ref, rtol=1e-4, atol=1e-4)
    go = np.random.RandomState(7).randn(*ref.shape).astype(np.float32)
    y.backward(_t(go))
    gin, goff, gmask, gw, gb = oracle.deform_conv_backward(x, off, mask, wgt, go, bias is not None, **og)
    _close(tx.grad, gin, rtol=1e-4, atol=1e-4)
    _close(toff.grad, goff, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(goff).max()))
    _close(tw.grad, gw, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(gw).max()))
    if modulated:
        _close(tmask.grad, gmask, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(gmask).max()))
        _close(tb.grad, gb, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(gb).max()))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_deform_conv_half_precision(dtype):
    """cfg-5: fp16 / bf16 storage, fp32 sampling arithmetic: <= 2e-2 relative to the fp32 oracle
    evaluated on the same (rounded) inputs."""
    from maskrcnn_benchmark.layers import DFConv2d, deform_conv

    g = GEOMS[2]
    x, off, _, wgt = _dcn_case(g, False)
    hx, hoff, hw = (_t(a).to(dtype) for a in (x, off, wgt))
    ref = oracle.deform_conv_forward(hx.float().cpu().numpy(), hoff.float().cpu().numpy(), None,
                                     hw.float().cpu().numpy(), None, pad=(1, 1), stride=(1, 1),
                                     dil=(1, 1), group=1, dg=1)
    y = deform_conv(hx, hoff, hw, 1, 1, 1, 1, 1)
    assert y.dtype == dtype
    err = (y.float().cpu().numpy() - ref)
    assert np.abs(err).max() <= 2e-2 * np.abs(ref).max()
    layer = DFConv2d(16, 16, with_modulated_dcn=True).to(DEV).to(dtype)
    out = layer(torch.randn(2, 16, 20, 24, device=DEV, dtype=dtype))
    assert out.shape == (2, 16, 20, 24) and torch.isfinite(out).all()
    out.float().sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())


# cfg-5: the three DCN layer shapes of R-101-FPN + DCN at 800x1344, B = 2
CFG5_SHAPES = [(128, 100, 168), (256, 50, 84), (512, 25, 42)]
_OG = dict(pad=(1, 1), stride=(1, 1), dil=(1, 1), group=1, dg=1)
_KG = dict(kh=3, kw=3, pad=(1, 1), stride=(1, 1), dil=(1, 1), dg=1)
_GEO = (3, 3, 1, 1, 1, 1, 1, 1, 1)


@pytest.mark.parametrize("shape", CFG5_SHAPES)
@pytest.mark.parametrize("modulated", [False, True])
def test_deform_conv_cfg5_shapes_fp32(shape, modulated):
    """deform_conv / modulated_deform_conv forward + every gradient at the cfg-5 layer shapes, fp32 <= 1e-4
    (relative to each tensor's scale) against the oracle (reference deform_conv_kernel_cuda.cu:197-472)."""
    from maskrcnn_benchmark.layers import deform_conv, modulated_deform_conv
    Cc, H, W = shape
    x, off, mask, wgt = synth.dcn_inputs(2, Cc, H, W, Cc, 3, 1, modulated, seed=13)
    tx, toff, tw = (_t(a).requires_grad_(True) for a in (x, off, wgt))
    if modulated:
        tmask = _t(mask).requires_grad_(True)
        y = modulated_deform_conv(tx, toff, tmask, tw, None, 1, 1, 1, 1, 1)
    else:
        y = deform_conv(tx, toff, tw, 1, 1, 1, 1, 1)
    ref = oracle.deform_conv_forward(x, off, mask, wgt, None, **_OG)
    _close(y, ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref).max()))
    go = np.random.RandomState(7).randn(*ref.shape).astype(np.float32)
    y.backward(_t(go))
    gin, goff, gmask, gw, _ = oracle.deform_conv_backward(x, off, mask, wgt, go, False, **_OG)
    _close(tx.grad, gin, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(gin).max()))
    _close(toff.grad, goff, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(goff).max()))
    _close(tw.grad, gw, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(gw).max()))
    if modulated:
        _close(tmask.grad, gmask, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(gmask).max()))


@pytest.mark.parametrize("shape", CFG5_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_deformable_kernels_cfg5_shapes_half(shape, dtype):
    """im2col / col2im / col2im_coord with fp16 / bf16 storage at the cfg-5 layer shapes: <= 2e-2 of the
    result's scale against the fp32 oracle evaluated on the same rounded inputs (modulated form)."""
    C = _C()
    Cc, H, W = shape
    x, off, mask, _ = synth.dcn_inputs(2, Cc, H, W, Cc, 3, 1, True, seed=14)
    hx, hoff, hmask = (_t(a).to(dtype) for a in (x, off, mask))
    rx, roff, rmask = (t.float().cpu().numpy() for t in (hx, hoff, hmask))
    col = C.deformable_im2col(hx, hoff, hmask, *_GEO)
    ref_col = oracle.deformable_im2col(rx, roff, rmask, **_KG)
    assert col.dtype == dtype
    assert np.abs(col.float().cpu().numpy() - ref_col).max() <= 2e-2 * np.abs(ref_col).max()
    gcol = _t(np.random.RandomState(9).randn(*ref_col.shape).astype(np.float32)).to(dtype)
    rgcol = gcol.float().cpu().numpy()
    gim = torch.zeros_like(hx)
    C.deformable_col2im(gcol, hoff, hmask, gim, *_GEO)
    ref_gim = oracle.deformable_col2im(rgcol, roff, rmask, *x.shape, **_KG)
    assert np.abs(gim.float().cpu().numpy() - ref_gim).max() <= 2e-2 * np.abs(ref_gim).max()
    goff, gmask = torch.empty_like(hoff), torch.empty_like(hmask)
    C.deformable_col2im_coord(gcol, hx, hoff, hmask, goff, gmask, *_GEO)
    rgoff, rgmask = oracle.deformable_col2im_coord(rgcol, rx, roff, rmask, **_KG)
    assert np.abs(goff.float().cpu().numpy() - rgoff).max() <= 2e-2 * np.abs(rgoff).max()
    assert np.abs(gmask.float().cpu().numpy() - rgmask).max() <= 2e-2 * np.abs(rgmask).max()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_deform_conv_cfg5_layer4_half_fwd_bwd(dtype):
    """the whole layer (im2col + GEMM, and the three backward paths) in half precision at the layer4 shape."""
    from maskrcnn_benchmark.layers import modulated_deform_conv
    Cc, H, W = CFG5_SHAPES[2]
    x, off, mask, wgt = synth.dcn_inputs(2, Cc, H, W, Cc, 3, 1, True, seed=15)
    wgt = wgt * 0.25   # keeps |y| ~ 1 in fp16
    hx, hoff, hmask, hw = (_t(a).to(dtype).requires_grad_(True) for a in (x, off, mask, wgt))
    rx, roff, rmask, rw = (t.detach().float().cpu().numpy() for t in (hx, hoff, hmask, hw))
    y = modulated_deform_conv(hx, hoff, hmask, hw, None, 1, 1, 1, 1, 1)
    ref = oracle.deform_conv_forward(rx, roff, rmask, rw, None, **_OG)
    assert y.dtype == dtype
    assert np.abs(y.float().detach().cpu().numpy() - ref).max() <= 2e-2 * np.abs(ref).max()
    go = _t(np.random.RandomState(7).randn(*ref.shape).astype(np.float32)).to(dtype)
    y.backward(go)
    gin, goff, gmask, gw, _ = oracle.deform_conv_backward(rx, roff, rmask, rw, go.float().cpu().numpy(), False, **_OG)
    for got, want in ((hx.grad, gin), (hoff.grad, goff), (hmask.grad, gmask), (hw.grad, gw)):
        assert np.abs(got.float().cpu().numpy() - want).max() <= 2e-2 * np.abs(want).max()


@pytest.mark.parametrize("modulated", [False, True])
def test_deform_conv_backward_with_kept_forward_copies_equals_rebuild(modulated):
    """layers that will be differentiated keep the channel-fastest input copy and the column matrix of their forward pass
    (`keep=` / `saved=`): the backward pass with them equals the one that rebuilds both"""
    C = _C()
    Cc, H, W = CFG5_SHAPES[1]
    x, off, mask, wgt = synth.dcn_inputs(2, Cc, H, W, Cc, 3, 1, modulated, seed=21)
    dt = torch.float16
    tx, toff, tw = (_t(a).to(dt) for a in (x, off, wgt * 0.25))
    tm = _t(mask).to(dt) if modulated else None
    out = torch.empty(2, Cc, H, W, device=DEV, dtype=dt)
    e = torch.empty(0, device=DEV, dtype=dt)
    keep = []
    if modulated:
        C.modulated_deform_conv_forward(tx, tw, torch.zeros(Cc, device=DEV, dtype=dt), e, toff, tm, out, e, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, False, keep=keep)
    else:
        C.deform_conv_forward(tx, tw, toff, out, e, e, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 2, keep=keep)
    assert len(keep) == 1 and keep[0][1].shape == (2 * H * W, 9 * Cc)       # the channels-last pipeline served it
    go = torch.randn(2, Cc, H, W, device=DEV).to(dt)
    a = C.deform_conv_backward_all(tx, toff, tm, tw, go, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, saved=keep[0])
    b = C.deform_conv_backward_all(tx, toff, tm, tw, go, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
    for p_, q_ in zip(a, b):
        assert (p_ is None) == (q_ is None)
        if p_ is not None:     # (pile-ups beyond the index's 8 slots are added by atomics: not bit-reproducible run to run)
            torch.testing.assert_close(p_.float(), q_.float(), rtol=2e-3, atol=2e-3 * float(q_.float().abs().max()))


# ============================================================================ fused FrozenBN
@pytest.mark.parametrize("shape", [(2, 8, 25, 42), (1, 5, 7, 9), (2, 16, 40, 64), (3, 4, 1, 1)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True), (False, True)])
def test_frozen_bn_fused_matches_reference_composite(shape, relu, res):
    """reference layers/batch_norm.py:19-31 (+ relu / residual of resnet.py:343-366) in plain torch
    fp32 on the CPU vs the fused HIP kernel: forward bit-equal (same mul-then-add roundings),
    backward bit-equal (mask and one multiply)."""
    from maskrcnn_benchmark.layers import FrozenBatchNorm2d
    torch.manual_seed(sum(shape) + relu + 2 * res)
    bn = FrozenBatchNorm2d(shape[1])
    bn.weight.copy_(torch.rand(shape[1]) + 0.5); bn.bias.copy_(torch.randn(shape[1]))
    bn.running_mean.copy_(torch.randn(shape[1])); bn.running_var.copy_(torch.rand(shape[1]) + 0.3)
    bn_d = FrozenBatchNorm2d(shape[1]).to(DEV)
    bn_d.load_state_dict(bn.state_dict())
    # the folded constants (weight * rsqrt(var), ...) are computed on the device; rsqrt differs from the
    # CPU's in the last bit, so the bit-exact reference uses the SAME constants, and the CPU module
    # (its own fold) is compared at 1e-6
    scale, bias = (t.cpu() for t in bn_d.folded())
    x = torch.randn(shape, requires_grad=True)
    r = torch.randn(shape, requires_grad=True) if res else None
    y = x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)
    y_mod = bn(x.detach())
    if res:
        y = y + r
        y_mod = y_mod + r.detach()
    if relu:
        y = torch.relu(y)
        y_mod = torch.relu(y_mod)
    gy = torch.randn(shape)
    y.backward(gy)
    xd = x.detach().to(DEV).requires_grad_(True)
    rd = r.detach().to(DEV).requires_grad_(True) if res else None
    yd = bn_d.fused(xd, relu=relu, residual=rd)
    yd.backward(gy.to(DEV))
    assert torch.equal(yd.detach().cpu(), y.detach())
    torch.testing.assert_close(yd.detach().cpu(), y_mod, rtol=1e-6, atol=1e-6)
    assert torch.equal(xd.grad.cpu(), x.grad)
    if res:
        assert torch.equal(rd.grad.cpu(), r.grad)
    # buffers changed in place -> the cached folded constants are refreshed
    bn_d.running_var.mul_(4.0)
    assert torch.equal(bn_d.fused(xd.detach()), bn_d(xd.detach()))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_frozen_bn_fused_half(dtype):
    from maskrcnn_benchmark.layers import FrozenBatchNorm2d
    torch.manual_seed(3)
    bn = FrozenBatchNorm2d(12).to(DEV)
    bn.weight.copy_(torch.rand(12) + 0.5); bn.running_mean.copy_(torch.randn(12))
    x = torch.randn(2, 12, 50, 84, device=DEV)
    r = torch.randn(2, 12, 50, 84, device=DEV)
    ref = torch.relu(bn(x) + r)
    out = bn.fused(x.to(dtype), relu=True, residual=r.to(dtype)).float()
    assert (out - ref).abs().max() <= 2e-2 * ref.abs().max()


@pytest.mark.parametrize("shape", CFG5_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("modulated", [False, True])
def test_deform_conv_fused_mfma_forward_cfg5_shapes(shape, dtype, modulated):
    """the fused implicit-GEMM forward (v_mfma_f32_32x32x16_{f16,bf16}, columns never written) at the cfg-5 layer
    shapes: <= 2e-2 of the output scale against the fp32 oracle on the same rounded inputs, and within half-precision
    accumulation noise of the unfused im2col + GEMM path."""
    from maskrcnn_benchmark.layers import deform_conv, modulated_deform_conv
    Cc, H, W = shape
    x, off, mask, wgt = synth.dcn_inputs(2, Cc, H, W, Cc, 3, 1, modulated, seed=31)
    wgt = wgt * 0.5
    hx, hoff, hw = (_t(a).to(dtype) for a in (x, off, wgt))
    hmask = _t(mask).to(dtype) if modulated else None
    hb = _t(np.random.RandomState(3).randn(Cc).astype(np.float32)).to(dtype) if modulated else None

    def run():
        if modulated:
            return modulated_deform_conv(hx, hoff, hmask, hw, hb, 1, 1, 1, 1, 1)
        return deform_conv(hx, hoff, hw, 1, 1, 1, 1, 1)

    _tune("dcn_fused", 1)   # also where the dispatch rule prefers im2col + GEMM
    from maskrcnn_benchmark import _C
    timer = _C.KernelTimer()
    _C.KERNEL_TIMER = timer
    y = run()
    _C.KERNEL_TIMER = None
    torch.cuda.synchronize()
    assert any(k.startswith("dcn_fused_fwd") for k in timer.results()), "the fused kernel did not run"
    _tune("dcn_fused", 2)
    y0 = run()
    f = lambda t: None if t is None else t.float().cpu().numpy()  # noqa: E731
    ref = oracle.deform_conv_forward(f(hx), f(hoff), f(hmask), f(hw), f(hb), **_OG)
    scale = np.abs(ref).max()
    assert y.dtype == dtype and np.abs(f(y) - ref).max() <= 2e-2 * scale
    assert np.abs(f(y) - f(y0)).max() <= 2e-2 * scale


# ------------------------------------------------------------------ ROIAlign over a channels-last pyramid (csrc/roi_align_nhwc.hip)
def _nhwc_case(C, K, ph, sr, seed, images=2, shapes=None):
    rng = np.random.RandomState(seed)
    shapes = shapes or synth.fpn_shapes()[:4]
    feats = [rng.randn(images, C, h, w).astype(np.float32) for (h, w) in shapes]
    rois = synth.fpn_rois(seed=seed + 5, per_image=K // images, n_images=images)
    scales = [1.0 / s for s in synth.FPN_STRIDES[:len(shapes)]]
    return feats, rois, scales


@pytest.mark.gpu
@pytest.mark.parametrize("C,K,ph,sr,out_cl", [(256, 1024, 7, 2, False), (256, 256, 14, 2, True), (256, 256, 14, 2, False),
                                               (64, 200, 7, 0, False), (6, 50, 5, 3, True), (80, 100, 7, 2, True)])
def test_roi_align_fpn_forward_channels_last_is_bit_equal_to_the_oracle(C, K, ph, sr, out_cl):
    """ROIAlign forward over a channels-last pyramid (NHWC kernels): same operation order as the reference CPU kernel with FP
    contraction off -> bit-equal to the oracle (csrc/cpu/ROIAlign_cpu.cpp:113-219), at the model's size (1024 x 256 x 7 x 7 over
    the four pyramid levels of the 800 x 1344 input) and for contiguous / channels-last pooled tensors, adaptive sampling,
    channel counts that are not multiples of 4."""
    from maskrcnn_benchmark import _C
    feats, rois, scales = _nhwc_case(C, K, ph, sr, seed=C + ph)
    tf = [_t(f).contiguous(memory_format=torch.channels_last) for f in feats]
    out, lv = _C.roi_align_fpn_forward(tf, _t(rois), scales, ph, ph, sr, 2, 5, out_channels_last=out_cl)
    assert _C.is_channels_last(out) == out_cl
    lvn = lv.cpu().numpy()
    assert np.array_equal(lvn, synth.level_map(rois))
    got = out.contiguous().cpu().numpy()
    for l, (f, s) in enumerate(zip(feats, scales)):
        sel = np.nonzero(lvn == l)[0]
        if sel.size:
            want = oracle.roi_align_forward(f, rois[sel], s, ph, ph, sr)
            assert np.array_equal(got[sel], want), "level %d: max diff %g" % (l, np.abs(got[sel] - want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("C,K,ph,sr,g_cl", [(256, 1024, 7, 2, False), (256, 256, 14, 2, True), (256, 128, 14, 2, False),
                                             (64, 200, 7, 0, False), (6, 50, 5, 3, True), (80, 100, 7, 2, False)])
def test_roi_align_fpn_backward_channels_last_matches_the_oracle_and_is_reproducible(C, K, ph, sr, g_cl):
    """ROIAlign backward into a channels-last pyramid (pixel-owner NHWC kernel, no atomics): <= 1e-4 relative to the
    fp64-accumulated oracle (csrc/cuda/ROIAlign_cuda.cu:125-254 restated), every level, from a contiguous and a channels-last
    pooled gradient; two runs are bit-identical; an `accumulate` into existing maps is not part of the Python surface."""
    from maskrcnn_benchmark import _C
    feats, rois, scales = _nhwc_case(C, K, ph, sr, seed=C + ph + 1)
    rng = np.random.RandomState(K)
    g = rng.randn(K, C, ph, ph).astype(np.float32)
    lvn = synth.level_map(rois)
    tg = _t(g).contiguous(memory_format=torch.channels_last) if g_cl else _t(g)
    shapes = [tuple(f.shape) for f in feats]
    a = _C.roi_align_fpn_backward(tg, _t(rois), _t(lvn), shapes, scales, ph, ph, sr, channels_last=True)
    b = _C.roi_align_fpn_backward(tg, _t(rois), _t(lvn), shapes, scales, ph, ph, sr, channels_last=True)
    for l, (f, s) in enumerate(zip(feats, scales)):
        assert a[l].shape == f.shape and (_C.is_channels_last(a[l]) or a[l].is_contiguous())
        assert torch.equal(a[l], b[l])
        sel = np.nonzero(lvn == l)[0]
        want = oracle.roi_align_backward(g[sel], rois[sel], s, ph, ph, *f.shape, sr, acc64=True)
        got = a[l].contiguous().cpu().numpy()
        assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), "level %d" % l


@pytest.mark.gpu
def test_roi_align_channels_last_adjoint_identity_at_model_size():
    """<fwd(x), g> == <x, bwd(g)> for the NHWC pair at the model's size (size-independent property; fp64 accumulation of the
    two inner products on the host)"""
    from maskrcnn_benchmark import _C
    feats, rois, scales = _nhwc_case(256, 1024, 7, 2, seed=11)
    rng = np.random.RandomState(2)
    g = rng.randn(1024, 256, 7, 7).astype(np.float32)
    tf = [_t(f).contiguous(memory_format=torch.channels_last) for f in feats]
    out, lv = _C.roi_align_fpn_forward(tf, _t(rois), scales, 7, 7, 2, 2, 5)
    gin = _C.roi_align_fpn_backward(_t(g), _t(rois), lv, [tuple(f.shape) for f in feats], scales, 7, 7, 2, channels_last=True)
    lhs = float((out.double() * _t(g).double()).sum())
    rhs = sum(float((a.double() * b.double()).sum()) for a, b in zip(tf, gin))
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs)), (lhs, rhs)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_roi_align_backward_non_finite_gradient_stays_in_the_rois_tiles(layout):
    """A non-finite pooled gradient (a GradScaler overflow step) of ONE bin: the branch-free walks multiply neighbouring rows by
    stored zero weights, so 0 * Inf = NaN may appear around the bin's own footprint — but only inside the 8 x 32 pixel tiles
    the ROI reaches (documented in DESIGN.md section 3.2); every other gradient element, every other level and image stay
    finite and equal to the run without the Inf."""
    from maskrcnn_benchmark import _C
    feats, rois, scales = _nhwc_case(64, 512, 7, 2, seed=77)
    rng = np.random.RandomState(5)
    g = rng.randn(512, 64, 7, 7).astype(np.float32)
    lvn = synth.level_map(rois)
    k = int(np.nonzero(lvn == 1)[0][0])          # a ROI on the stride-8 level
    g2 = g.copy()
    g2[k, 3, 2, 4] = np.inf
    shapes = [tuple(f.shape) for f in feats]
    cl = layout == "nhwc"
    a = _C.roi_align_fpn_backward(_t(g), _t(rois), _t(lvn), shapes, scales, 7, 7, 2, channels_last=cl)
    b = _C.roi_align_fpn_backward(_t(g2), _t(rois), _t(lvn), shapes, scales, 7, 7, 2, channels_last=cl)
    img = int(rois[k, 0])
    x1, y1, x2, y2 = (rois[k, 1:] * scales[1])
    H, W = shapes[1][2:]
    ty0, ty1 = max(int(np.floor(y1)) - 1, 0) // 8 * 8, min((int(np.ceil(y2)) + 1) // 8 * 8 + 8, H)
    tx0, tx1 = max(int(np.floor(x1)) - 1, 0) // 32 * 32, min((int(np.ceil(x2)) + 1) // 32 * 32 + 32, W)
    for l in range(len(feats)):
        ga, gb = a[l].contiguous().cpu().numpy(), b[l].contiguous().cpu().numpy()
        bad = ~np.isfinite(gb)
        if l != 1:
            assert not bad.any() and np.array_equal(ga, gb)
            continue
        assert bad.any()
        allowed = np.zeros_like(bad)
        allowed[img, :, ty0:ty1, tx0:tx1] = True
        assert not (bad & ~allowed).any(), "non-finite values outside the ROI's tiles"
        assert np.array_equal(ga[~allowed], gb[~allowed])


@pytest.mark.gpu
def test_float64_operators_on_the_device():
    """float64 tensors through the reference-named `_C` operators (AT_DISPATCH_FLOATING_TYPES of the reference: float AND double):
    csrc/f64_ops.hip on the device — ROIAlign forward / backward against a float64 torch formulation (1e-12 / 1e-10), ROIPool and
    focal loss against the fp32 oracle on fp32-representable inputs, NMS the same kept set as the oracle."""
    from maskrcnn_benchmark import _C
    from torch_refs import roi_align_torch
    inp, rois, scale = synth.cfg1_roi_align(seed=12, K=200, C=16)
    x64, r64 = _t(inp).double(), _t(rois).double()
    out = _C.roi_align_forward(x64, r64, scale, 7, 7, 2)
    assert out.dtype == torch.float64 and out.is_cuda
    xg = x64.cpu().clone().requires_grad_()                      # the float64 torch formulation runs on the host
    ref = roi_align_torch(xg, r64.cpu(), scale, 7, 7, 2)
    assert torch.allclose(out.cpu(), ref.detach(), rtol=1e-12, atol=1e-12)
    assert np.abs(out.cpu().numpy() - oracle.roi_align_forward(inp, rois, scale, 7, 7, 2)).max() <= 1e-5
    g = torch.randn(out.shape, dtype=torch.float64)
    ref.backward(g)
    gin = _C.roi_align_backward(g.to(DEV), r64, scale, 7, 7, *inp.shape, 2)
    assert torch.allclose(gin.cpu(), xg.grad, rtol=1e-9, atol=1e-11)
    o32, a32 = oracle.roi_pool_forward(inp, rois, scale, 5, 4)
    o64, a64 = _C.roi_pool_forward(x64, r64, scale, 5, 4)
    assert np.array_equal(o64.cpu().numpy().astype(np.float32), o32) and np.array_equal(a64.cpu().numpy(), a32)
    gp = torch.randn(o64.shape, dtype=torch.float64, device=DEV)
    gi = _C.roi_pool_backward(gp, x64, r64, a64, scale, 5, 4, *inp.shape)
    want = oracle.roi_pool_backward(gp.cpu().numpy().astype(np.float32), rois, a32, *inp.shape)
    assert np.abs(gi.cpu().numpy() - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    logits, targets = synth.focal_inputs(3000, 80)
    f = _C.sigmoid_focalloss_forward(_t(logits).double(), _t(targets), 80, 2.0, 0.25)
    np.testing.assert_allclose(f.cpu().numpy(), oracle.sigmoid_focal_loss_forward(logits, targets, 2.0, 0.25), rtol=1e-4, atol=1e-6)
    d = np.random.RandomState(3).rand(*logits.shape).astype(np.float32)
    b = _C.sigmoid_focalloss_backward(_t(logits).double(), _t(targets), _t(d).double(), 80, 2.0, 0.25)
    np.testing.assert_allclose(b.cpu().numpy(), oracle.sigmoid_focal_loss_backward(logits, targets, d, 2.0, 0.25), rtol=1e-4, atol=1e-6)
    bx, sc = synth.nms_boxes(3000, seed=9)
    keep = _C.nms(_t(bx).double(), _t(sc).double(), 0.6)
    assert keep.dtype == torch.int64 and np.array_equal(keep.cpu().numpy(), oracle.nms(bx, sc, 0.6))


# ------------------------------------------------------------------ bias (+ ReLU) behind a channels-last convolution (csrc/bias_act.hip)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.float16, 2e-3), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C,H,W", [(256, 200, 336), (256, 25, 42), (64, 9, 3), (12, 50, 84), (81, 28, 28), (3, 100, 168)])
@pytest.mark.parametrize("relu", [False, True])
def test_bias_act_channels_last_forward_backward_and_bias_gradient(dtype, tol, C, H, W, relu):
    """y = [relu](x + b) on a channels-last activation and its one-pass backward: grad_x is an exact select, the bias gradient
    is the fp32 column sum of the masked gradient (reference: the `+ bias` of the FPN / RPN / mask-head convolutions,
    modeling/backbone/fpn.py:30-40, rpn.py:61-76), at the sizes of the model's P2 lateral, a coarse level, the RPN's 3/12-channel
    outputs and the mask logits (column_sum path); two runs give the same bits."""
    from maskrcnn_benchmark import _C
    g = torch.Generator(device="cpu").manual_seed(C + H)
    x = torch.randn(2, C, H, W, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(2, C, H, W, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, generator=g).to(DEV).requires_grad_()
    assert _C.bias_act_supported(x, b)

    def run():
        xi = x.clone(memory_format=torch.channels_last).requires_grad_()
        b.grad = None
        y = _C.bias_act(xi, b, relu)
        y.backward(gy)
        return y.detach(), xi.grad, b.grad.clone()

    y, gx, gb = run()
    assert _C.is_channels_last(y) and y.dtype == dtype and gb.dtype == torch.float32
    ref = x.float() + b.detach().view(1, -1, 1, 1)
    ref = ref.relu() if relu else ref
    assert (y.float() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    want_gx = gy.float() * (y.float() > 0).float() if relu else gy.float()
    assert torch.equal(gx.float(), want_gx)
    want_gb = want_gx.double().sum((0, 2, 3))
    assert (gb.double() - want_gb).abs().max() <= 1e-5 * max(1.0, float(want_gx.abs().double().sum((0, 2, 3)).max()))
    y2, gx2, gb2 = run()
    assert torch.equal(y2, y) and torch.equal(gx2, gx) and torch.equal(gb2, gb)


@pytest.mark.parametrize("no_presorted", [0, 1])
def test_nms_input_already_in_score_order_skips_the_sort_network(no_presorted):
    """the detector's NMS input is top-k output (scores descending, ties in position order): the fused launch's sort
    workgroups detect ascending keys and skip the network.  Bit-exact against the oracle (reference csrc/cpu/nms_cpu.cpp:37-63)
    with the detection on and off: the model's 10 RPN segments in score order, runs of equal scores, single inversions at the
    end / across a wave / a thread boundary, LDS-sorted sizes (<= 4096), an unsorted segment among sorted ones; three rounds
    on the allocator's dirty workspace."""
    from maskrcnn_benchmark import _lib
    _lib.tuning_set("nms_no_presorted", no_presorted)
    try:
        def in_order(b, sc, ties=False, swap=None):
            o = np.argsort(-sc, kind="stable")
            b, sc = np.ascontiguousarray(b[o]), np.ascontiguousarray(sc[o])
            if ties and len(sc) > 200:
                sc[10:40] = sc[10]
                sc[len(sc) // 2:len(sc) // 2 + 70] = sc[len(sc) // 2]
            if swap is not None:
                sc[[swap, swap + 1]] = sc[[swap + 1, swap]]
            return b, sc
        segs = [in_order(b, s, ties=(i % 3 == 1)) for i, (b, s) in enumerate(synth.rpn_nms_segments())]
        segs += [in_order(*synth.nms_boxes(2000, seed=3), swap=1998), in_order(*synth.nms_boxes(1500, seed=4), swap=255),
                 in_order(*synth.nms_boxes(1500, seed=5), swap=3), in_order(*synth.nms_boxes(4096, seed=6)),
                 in_order(*synth.nms_boxes(3000, seed=7), swap=2998), synth.nms_boxes(700, seed=8), in_order(*synth.nms_boxes(1, seed=9))]
        boxes = np.concatenate([b for b, _ in segs])
        scores = np.concatenate([s for _, s in segs])
        offs = np.cumsum([0] + [len(s) for _, s in segs]).astype(np.int32)
        tb, ts, to = _t(boxes), _t(scores), _t(offs)
        refs = [oracle.nms(b, s, 0.7) for b, s in segs]
        for rep in range(3):
            keep, num = _C().nms_batched(tb, ts, to, 4096, 0.7)
            km, _ = _C().nms_batched_mask(tb, ts, to, 4096, 0.7)
            keep, num, km = keep.cpu().numpy(), num.cpu().numpy(), km.cpu().numpy()
            for i, ref in enumerate(refs):
                assert num[i] == len(ref), (rep, i)
                np.testing.assert_array_equal(keep[offs[i]:offs[i] + num[i]], ref)
                want = np.zeros(offs[i + 1] - offs[i], np.uint8)
                want[ref] = 1
                np.testing.assert_array_equal(km[offs[i]:offs[i + 1]].astype(np.uint8), want)
        assert _C().nms_repaired_segments(torch.device(DEV, 0) if isinstance(DEV, str) else DEV) >= 0
    finally:
        _lib.tuning_set("nms_no_presorted", 0)
