"""Kernel-logic checks WITHOUT a GPU: the HIP sources of the ROIAlign / ROIPool kernels are compiled
as host C++ and executed by the fiber-based emulation in tests/emu/ (workgroups, barriers, wave
ballots, LDS), then compared with the oracle.  This is not the parity gate (that is `-m gpu`, on
the device); it catches indexing / barrier / layout mistakes before GPU minutes are spent.  The
forward comparisons are bit-exact, which also pins the emulation itself."""
import os

import numpy as np
import pytest

import oracle
import synth
import emu


def _edge_rois():
    return np.array([[0, -50, -50, -20, -20], [0, 300, 300, 400, 400], [0, -10, -10, 500, 500],
                     [0, 100, 100, 100, 100], [0, 223, 223, 224, 224], [0, -17, 5, 3, 40]], np.float32)


@pytest.fixture(params=["scan", "atomic"])
def bwd_impl(request):
    emu.tuning_set("roi_bwd_impl", {"scan": 2, "atomic": 3}[request.param])
    return request.param


@pytest.mark.parametrize("ph,pw,sr", [(7, 7, 2), (14, 14, 2), (7, 7, 1), (7, 7, 0), (3, 5, 3), (1, 1, 1)])
def test_emu_roi_align_forward_bit_exact(ph, pw, sr):
    inp, rois, scale = synth.cfg1_roi_align(K=48, C=5)
    rois = np.concatenate([rois, _edge_rois()])
    out = emu.roi_align_forward(inp, rois, scale, ph, pw, sr)
    assert np.array_equal(out, oracle.roi_align_forward(inp, rois, scale, ph, pw, sr))


@pytest.mark.parametrize("ph,pw,sr", [(7, 7, 2), (14, 14, 2), (7, 7, 1), (3, 5, 2), (16, 16, 2)])
def test_emu_roi_align_forward_small_maps_of_two_images_bit_exact(ph, pw, sr):
    """one small map per image, the ROIs of two images interleaved, a ragged last channel chunk (40 channels), edge ROIs
    (outside the map, degenerate, slivers): bit-equal to the oracle.  (Round 6 built a kernel that keeps the planes of a small
    map in LDS for this shape — BASELINE configs[0] — and rejected it: 36 us against 26 us, its 16 scattered 4-byte LDS reads
    per output and channel run into bank conflicts; tools/rejected_kernels/r06_small_map_forward_lds.patch)"""
    rng = np.random.RandomState(31)
    inp, rois, scale = synth.cfg1_roi_align(K=150, C=40)
    inp = np.concatenate([inp, rng.randn(*inp.shape).astype(np.float32)], 0)
    rois = np.concatenate([rois, _edge_rois()]).astype(np.float32)
    rois[:, 0] = rng.randint(0, 2, rois.shape[0])
    out = emu.roi_align_forward(inp, rois, scale, ph, pw, sr)
    assert np.array_equal(out, oracle.roi_align_forward(inp, rois, scale, ph, pw, sr))


@pytest.mark.parametrize("ph,pw,sr", [(7, 7, 2), (14, 14, 2), (7, 7, 0), (3, 5, 3), (20, 20, 2)])
def test_emu_roi_align_backward_small_map(ph, pw, sr, bwd_impl):
    inp, rois, scale = synth.cfg1_roi_align(K=40, C=5)
    rois = np.concatenate([rois, _edge_rois()])
    g = np.random.RandomState(1).randn(rois.shape[0], 5, ph, pw).astype(np.float32)
    ref = oracle.roi_align_backward(g, rois, scale, ph, pw, *inp.shape, sr, acc64=True)
    out = emu.roi_align_backward(g, rois, scale, ph, pw, *inp.shape, sr)  # NaN-prefilled: all written
    assert np.abs(out - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("ct", ["4", "16"])
def test_emu_roi_align_backward_tile_seams_accumulate_and_chunking(ct):
    """multi-tile odd-sized maps, 2 images, channel count not a multiple of the chunk, > 256 ROIs
    (two scan rounds), the accumulate flag, K = 0."""
    emu.tuning_set("roi_bwd_impl", 2)
    emu.tuning_set("roi_bwd_scan_ct", int(ct))
    rng = np.random.RandomState(11)
    N, C, H, W = 2, 6 if ct == "4" else 21, 27, 70
    K = 300
    x1 = rng.uniform(-20, 270, K)
    y1 = rng.uniform(-20, 100, K)
    rois = np.stack([rng.randint(0, N, K), x1, y1, x1 + rng.uniform(2, 120, K), y1 + rng.uniform(2, 60, K)],
                    1).astype(np.float32)
    for (ph, pw, sr) in ((7, 7, 2), (14, 14, 2), (5, 3, 0)):
        g = rng.randn(K, C, ph, pw).astype(np.float32)
        ref = oracle.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr, acc64=True)
        out = emu.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr)
        tol = 1e-5 * max(1.0, np.abs(ref).max())
        assert np.abs(out - ref).max() <= tol
        assert np.array_equal(out, emu.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr))  # deterministic
        base = rng.randn(N, C, H, W).astype(np.float32)
        acc = emu.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr, into=base)
        assert np.abs(acc - (base + ref)).max() <= 2 * tol
    z = emu.roi_align_backward(np.zeros((0, C, 7, 7), np.float32), np.zeros((0, 5), np.float32), 0.25, 7, 7, N, C, H, W, 2)
    assert not z.any()


def test_emu_roi_align_fpn_fused_levels(bwd_impl):
    """the multi-level entry points: device-side LevelMapper, level-ordered work items."""
    rng = np.random.RandomState(5)
    shapes = [(2, 4, 50, 84), (2, 4, 25, 42), (2, 4, 13, 21), (2, 4, 7, 11)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    feats = [rng.randn(*s).astype(np.float32) for s in shapes]
    rois = synth.fpn_rois(seed=4, per_image=40, smin=8, smax=300)
    rois[:, 1:] *= 0.25  # a 200x336 image
    lv = synth.level_map(rois)
    out, levels = emu.roi_align_fpn_forward(feats, rois, scales, 7, 7, 2, 2, 5)
    assert np.array_equal(levels, lv)
    for l in range(4):
        sel = lv == l
        assert np.array_equal(out[sel], oracle.roi_align_forward(feats[l], rois[sel], scales[l], 7, 7, 2))
    g = rng.randn(rois.shape[0], 4, 7, 7).astype(np.float32)
    gins = emu.roi_align_fpn_backward(g, rois, lv, shapes, scales, 7, 7, 2)
    for l in range(4):
        sel = lv == l
        ref = oracle.roi_align_backward(g[sel], rois[sel], scales[l], 7, 7, *shapes[l], 2, acc64=True)
        assert np.abs(gins[l] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("impl", ["dma+order", "dma", "generic"])
def test_emu_roi_align_fpn_forward_visiting_order_and_staging_variants(impl):
    """K >= 384 switches the ROI ranking pre-pass on (poisoned workspace: every order slot must be written, and
    the order must be a permutation — a lost or duplicated ROI leaves NaN rows); the LDS-DMA and the generic
    gather kernels are bit-equal to the oracle, border pieces (replicated column / row) included."""
    if impl == "generic":
        emu.tuning_set("roi_fwd_impl", 1)
    emu.tuning_set("roi_fwd_order", 1 if impl != "dma+order" else 2)   # 2: rank although the maps are tiny
    rng = np.random.RandomState(11)
    shapes = [(2, 3, 50, 84), (2, 3, 25, 42), (2, 3, 13, 21), (2, 3, 7, 11)]
    feats = [rng.randn(*s).astype(np.float32) for s in shapes]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    rois = synth.fpn_rois(seed=5, per_image=200, smin=8, smax=300)
    rois[:, 1:] *= 0.25
    rois[:7, 3] = 335.0                      # touching the right / bottom border: replicated border pieces
    rois[7:14, 4] = 199.0
    rois[14] = rois[15]                      # identical ROIs: equal keys, ranks must stay distinct
    assert emu.lib().detops_roi_align_forward_workspace_bytes(rois.shape[0], 7, 7, 2) > 0
    lv = synth.level_map(rois)
    emu.stats(reset=True)
    out, levels = emu.roi_align_fpn_forward(feats, rois, scales, 7, 7, 2, 2, 5)
    assert emu.stats().get("fwd.ranked_rois", 0) == (rois.shape[0] if impl == "dma+order" else 0)
    assert np.array_equal(levels, lv)
    for l in range(4):
        sel = lv == l
        assert np.array_equal(out[sel], oracle.roi_align_forward(feats[l], rois[sel], scales[l], 7, 7, 2))
    # single-level entry, 14x14 bins
    out1 = emu.roi_align_forward(feats[1], rois, scales[1], 14, 14, 2)
    assert np.array_equal(out1, oracle.roi_align_forward(feats[1], rois, scales[1], 14, 14, 2))


@pytest.mark.parametrize("K", [384, 385, 415, 1025, 4096])
def test_emu_roi_order_prepass_is_a_permutation_at_its_size_limits(K):
    """ranking pre-pass at the smallest / largest ROI counts it serves and at counts that are not a multiple of
    its 32-key LDS padding; duplicate ROIs and ROIs with NaN / huge coordinates (the key only has to be SOME
    number: any permutation is correct, a lost ROI would leave its output rows NaN-poisoned)"""
    emu.tuning_set("roi_fwd_order", 2)
    rng = np.random.RandomState(K)
    feats = [rng.randn(1, 2, 20, 30).astype(np.float32), rng.randn(1, 2, 10, 15).astype(np.float32)]
    scales = [0.25, 0.125]
    rois = synth.fpn_rois(seed=K, per_image=K, n_images=1, smin=8, smax=200)
    rois[:, 1:] *= 0.09
    rois[5] = rois[6]
    rois[7, 1:] = [1e30, 1e30, 1e30, 1e30]       # far outside: all-zero output rows
    emu.stats(reset=True)
    out, levels = emu.roi_align_fpn_forward(feats, rois, scales, 7, 7, 2, 2, 3)
    assert emu.stats().get("fwd.ranked_rois", 0) == K
    lv = np.clip(levels, 0, 1)
    assert not np.isnan(out).any()
    for l in range(2):
        sel = levels == l
        if sel.any():
            assert np.array_equal(out[sel], oracle.roi_align_forward(feats[l], rois[sel], scales[l], 7, 7, 2))
    assert not out[7].any()


# ================================================================================ deformable conv
DCN_GEOMS = [dict(B=2, C=8, H=13, W=17, k=3, stride=1, pad=1, dil=1, dg=1),
             dict(B=2, C=8, H=14, W=15, k=3, stride=2, pad=2, dil=2, dg=2),
             dict(B=1, C=20, H=9, W=33, k=3, stride=1, pad=1, dil=1, dg=1),
             dict(B=1, C=160, H=5, W=7, k=3, stride=1, pad=1, dil=1, dg=1),    # coord kernel: 16 channel slices
             dict(B=2, C=66, H=6, W=5, k=3, stride=1, pad=1, dil=1, dg=2)]     # 4 slices, ragged channel split


def _dcn_case(g, modulated, seed=3):
    x, off, mask, _ = synth.dcn_inputs(g["B"], g["C"], g["H"], g["W"], 4, g["k"], g["dg"], modulated, seed=seed)
    Ho = (g["H"] + 2 * g["pad"] - (g["dil"] * (g["k"] - 1) + 1)) // g["stride"] + 1
    Wo = (g["W"] + 2 * g["pad"] - (g["dil"] * (g["k"] - 1) + 1)) // g["stride"] + 1
    off = np.ascontiguousarray(off[:, :, :Ho, :Wo])
    mask = None if mask is None else np.ascontiguousarray(mask[:, :, :Ho, :Wo])
    return x, off, mask


@pytest.mark.parametrize("gi", range(len(DCN_GEOMS)))
@pytest.mark.parametrize("modulated", [False, True])
def test_emu_deformable_kernels_vs_oracle(gi, modulated):
    """im2col, col2im (atomic scatter AND the inverted-index gather path) and col2im_coord."""
    g = DCN_GEOMS[gi]
    x, off, mask = _dcn_case(g, modulated)
    k, p, s, d, dg = g["k"], g["pad"], g["stride"], g["dil"], g["dg"]
    geo = dict(kh=k, kw=k, pad=(p, p), stride=(s, s), dil=(d, d), dg=dg)
    col = emu.deformable_im2col(x, off, mask, **geo)
    ref_col = oracle.deformable_im2col(x, off, mask, **geo)
    np.testing.assert_allclose(col, ref_col, rtol=1e-5, atol=1e-5)
    gcol = np.random.RandomState(9).randn(*ref_col.shape).astype(np.float32)
    ref = oracle.deformable_col2im(gcol, off, mask, *x.shape, **geo)
    tol = dict(rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(emu.deformable_col2im(gcol, off, mask, *x.shape, gather=False, **geo), ref, **tol)
    out = emu.deformable_col2im(gcol, off, mask, *x.shape, gather=True, **geo)
    np.testing.assert_allclose(out, ref, **tol)
    # accumulate semantics (callers hand in a zeroed or partially filled gradient) + determinism
    base = np.random.RandomState(2).randn(*x.shape).astype(np.float32)
    np.testing.assert_allclose(emu.deformable_col2im(gcol, off, mask, *x.shape, gather=True, into=base, **geo),
                               base + ref, rtol=1e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))
    assert np.array_equal(out, emu.deformable_col2im(gcol, off, mask, *x.shape, gather=True, **geo))
    goff, gmask = emu.deformable_col2im_coord(gcol, x, off, mask, **geo)
    roff, rmask = oracle.deformable_col2im_coord(gcol, x, off, mask, **geo)
    np.testing.assert_allclose(goff, roff, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(roff).max()))
    if modulated:
        np.testing.assert_allclose(gmask, rmask, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(rmask).max()))


# ================================================================================ NMS
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 129, 700])
def test_emu_nms_bit_exact(n):
    b, s = synth.nms_boxes(n, seed=n)
    for thr in (0.3, 0.7):
        assert np.array_equal(emu.nms(b, s, thr), oracle.nms(b, s, thr))


def _nms_chain(n, step):
    """boxes marching along x: each overlaps only its near neighbours, scores descending with the index — the
    greedy choice is a dependency chain (0 kept -> 1 dropped -> 2 kept ...) as long as the row block."""
    x0 = np.arange(n, dtype=np.float32) * step
    b = np.stack([x0, np.zeros(n, np.float32), x0 + 99, np.full(n, 49, np.float32)], 1)
    return b, np.linspace(1.0, 0.1, n).astype(np.float32)


@pytest.mark.parametrize("step,thr", [(10, 0.8), (25, 0.55), (34, 0.45), (50, 0.3)])
def test_emu_nms_dependency_chains(step, thr):
    """worst case of the scan's fixed-point resolve: chains of depth 64 inside a row block, across blocks, and
    the 4097-box case that takes the shared-memory scan"""
    for n in (64, 200, 4097):
        b, s = _nms_chain(n, step)
        ref = oracle.nms(b, s, thr)
        assert 1 < len(ref) < n
        assert np.array_equal(emu.nms(b, s, thr), ref)


def test_emu_nms_ties_large_n_and_batched():
    b, s = synth.nms_boxes(300, seed=4)
    s[::3] = s[0]          # score ties: stable (ascending index) order like the CPU sort
    b[5] = b[4]            # identical boxes: IoU == 1
    assert np.array_equal(emu.nms(b, s, 0.5), oracle.nms(b, s, 0.5))
    b, s = synth.nms_boxes(8300, seed=5)   # > 8192: the radix-sort path (std::sort in emulation)
    assert np.array_equal(emu.nms(b, s, 0.6), oracle.nms(b, s, 0.6))
    segs = [synth.nms_boxes(n, seed=10 + i) for i, n in enumerate((130, 1, 64, 257, 40))]
    boxes = np.concatenate([x for x, _ in segs])
    scores = np.concatenate([y for _, y in segs])
    offs = np.cumsum([0] + [len(y) for _, y in segs]).astype(np.int32)
    keep, num = emu.nms_batched(boxes, scores, offs, 257, 0.7)
    km, num2 = emu.nms_batched(boxes, scores, offs, 257, 0.7, mask=True)
    assert np.array_equal(num, num2)
    for i, (x, y) in enumerate(segs):
        ref = oracle.nms(x, y, 0.7)
        assert num[i] == len(ref)
        assert np.array_equal(keep[offs[i]:offs[i] + num[i]], ref)
        want = np.zeros(len(y), np.uint8)
        want[ref] = 1
        assert np.array_equal(km[offs[i]:offs[i + 1]], want)


@pytest.mark.parametrize("fused", [0, 2])
def test_emu_nms_single_launch_and_three_launch_paths(fused):
    """n <= 4096: sort, mask tiles and scan are ONE launch (workgroup roles + flags between them); `nms_fused=2` keeps the
    three launches.  Segment lengths hit the three scan widths (<= 16 / <= 32 / <= 64 column words: 4 / 2 / 1 mask rows
    per wave-wide load), block boundaries, an empty segment and dependency chains; called twice on the same (dirty)
    workspace contents, since the control block of the fused launch is never cleared by the host."""
    emu.tuning_set("nms_fused", fused)
    sizes = (1000, 0, 1025, 64, 2000, 1, 2049, 4096, 65, 3000)
    segs = [synth.nms_boxes(n, seed=40 + i) if n else (np.zeros((0, 4), np.float32), np.zeros(0, np.float32))
            for i, n in enumerate(sizes)]
    segs[4] = _nms_chain(2000, 25)            # depth-64 chains inside the 2-rows-per-load width
    boxes = np.concatenate([x for x, _ in segs])
    scores = np.concatenate([y for _, y in segs])
    offs = np.cumsum([0] + [len(y) for _, y in segs]).astype(np.int32)
    for thr in (0.7, 0.55):
        keep, num = emu.nms_batched(boxes, scores, offs, 4096, thr)
        km, num2 = emu.nms_batched(boxes, scores, offs, 4096, thr, mask=True)
        assert np.array_equal(num, num2)
        for i, (x, y) in enumerate(segs):
            ref = oracle.nms(x, y, thr) if len(y) else np.zeros(0, np.int64)
            assert num[i] == len(ref), (i, sizes[i])
            assert np.array_equal(keep[offs[i]:offs[i] + num[i]], ref), (i, sizes[i])
            want = np.zeros(len(y), np.uint8)
            want[ref] = 1
            assert np.array_equal(km[offs[i]:offs[i + 1]], want)
    for n in (17 * 64, 33 * 64 - 1, 2048, 1024, 3):     # register sort: full, half-full and nearly empty key sets
        b, sc = synth.nms_boxes(n, seed=n)
        assert np.array_equal(emu.nms(b, sc, 0.6), oracle.nms(b, sc, 0.6))


@pytest.mark.parametrize("no_presorted", [0, 1])
def test_emu_nms_input_already_in_score_order_skips_the_sort_network(no_presorted):
    """The detector hands the segmented NMS top-k output (scores descending, ties in position order): the sort workgroups
    detect ascending keys and skip the bitonic network (csrc/nms.hip, block_all).  Same results as the full network
    (`nms_no_presorted=1`) and as the oracle for: sorted segments of every sort form (register keys <= 2048, LDS keys <= 4096),
    sorted with runs of equal scores, one inversion at the very end / across a wave boundary / across a thread boundary
    (must NOT be taken for sorted), a descending run followed by padding, an unsorted segment beside sorted ones."""
    emu.tuning_set("nms_no_presorted", no_presorted)

    def sorted_case(n, seed, ties=False):
        b, sc = synth.nms_boxes(n, seed=seed)
        order = np.argsort(-sc, kind="stable")
        b, sc = b[order], sc[order]
        if ties:
            sc[10:40] = sc[10]
            sc[n // 2:n // 2 + 70] = sc[n // 2]
        return np.ascontiguousarray(b), np.ascontiguousarray(sc)

    def swapped(n, seed, i):
        b, sc = sorted_case(n, seed)
        sc[[i, i + 1]] = sc[[i + 1, i]]
        return b, sc

    segs = [sorted_case(2000, 1), sorted_case(819, 2, ties=True), swapped(2000, 3, 1998), swapped(1500, 4, 255), swapped(1500, 5, 3),
            sorted_case(4096, 6), swapped(3000, 7, 2998), synth.nms_boxes(700, seed=8), sorted_case(1, 9), sorted_case(65, 10, ties=False)]
    boxes = np.concatenate([x for x, _ in segs])
    scores = np.concatenate([y for _, y in segs])
    offs = np.cumsum([0] + [len(y) for _, y in segs]).astype(np.int32)
    keep, num = emu.nms_batched(boxes, scores, offs, 4096, 0.7)
    km, _ = emu.nms_batched(boxes, scores, offs, 4096, 0.7, mask=True)
    for i, (x, y) in enumerate(segs):
        ref = oracle.nms(x, y, 0.7)
        assert num[i] == len(ref), i
        assert np.array_equal(keep[offs[i]:offs[i] + num[i]], ref), i
        want = np.zeros(len(y), np.uint8)
        want[ref] = 1
        assert np.array_equal(km[offs[i]:offs[i + 1]], want), i
    for (x, y) in (segs[0], segs[2], segs[5]):            # the single-segment entry point
        assert np.array_equal(emu.nms(x, y, 0.5), oracle.nms(x, y, 0.5))


def test_emu_nms_failed_segments_are_redone_by_the_repair_launch():
    """VERDICT r04 "missing" #4: the single launch's failure marker must not turn into "this segment proposes nothing".
    Fault injection (the sort workgroups publish a wrong token, test-sized polling budget -> every consumer wait gives
    up): with the repair launch (the default) every segment comes out exactly as without the fault and the sticky status
    word counts them; with it switched off the marker (num_keep = -1, all-zero mask) is what the caller sees."""
    sizes = (300, 0, 65, 1000, 1, 2049)
    segs = [synth.nms_boxes(n, seed=70 + i) if n else (np.zeros((0, 4), np.float32), np.zeros(0, np.float32))
            for i, n in enumerate(sizes)]
    segs[3] = _nms_chain(1000, 25)
    boxes = np.concatenate([x for x, _ in segs])
    scores = np.concatenate([y for _, y in segs])
    offs = np.cumsum([0] + [len(y) for _, y in segs]).astype(np.int32)
    status = np.zeros(1, np.int32)
    good_keep, good_num = emu.nms_batched(boxes, scores, offs, 2049, 0.7, status=status)
    good_mask, _ = emu.nms_batched(boxes, scores, offs, 2049, 0.7, mask=True, status=status)
    assert status[0] == 0
    try:
        emu.tuning_set("nms_fault", 1)
        emu.tuning_set("nms_spin_budget", 200)
        keep, num = emu.nms_batched(boxes, scores, offs, 2049, 0.7, status=status)
        assert status[0] == len(sizes)
        km, num2 = emu.nms_batched(boxes, scores, offs, 2049, 0.7, mask=True, status=status)
        assert status[0] == 2 * len(sizes)
        assert np.array_equal(num, good_num) and np.array_equal(num2, good_num)
        assert np.array_equal(km, good_mask)
        for i, (x, y) in enumerate(segs):
            ref = oracle.nms(x, y, 0.7) if len(y) else np.zeros(0, np.int64)
            assert num[i] == len(ref)
            assert np.array_equal(keep[offs[i]:offs[i] + num[i]], ref)
        km0, num0 = emu.nms_batched(boxes, scores, offs, 2049, 0.7, mask=True)          # old entry point, NULL status
        assert np.array_equal(km0, good_mask) and np.array_equal(num0, good_num)
        emu.tuning_set("nms_no_repair", 1)
        km, num = emu.nms_batched(boxes, scores, offs, 2049, 0.7, mask=True, status=status)
        assert (num == -1).all() and not km.any() and status[0] == 2 * len(sizes)
    finally:
        for k in ("nms_fault", "nms_spin_budget", "nms_no_repair"):
            emu.tuning_set(k, 0)


def test_emu_nms_threshold_boundary_is_exact():
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
            assert np.array_equal(emu.nms(boxes, scores, float(thr)), want), (trial, float(ovr), float(thr))
            assert len(want) == (1 if ovr >= thr else 2)
            cases += 1
    assert cases > 300
    # degenerate unions: zero-area / inverted boxes (negative "areas"), identical boxes, huge coordinates
    odd = np.array([[0, 0, -1, -1], [0, 0, -1, -1], [5, 5, 4, 9], [5, 5, 4, 9], [0, 0, 10, 10], [0, 0, 10, 10],
                    [-3e18, -3e18, 3e18, 3e18], [-3e18, -3e18, 3e18, 3e18], [1, 1, 0.5, 0.5], [0, 0, 1e-20, 1e-20]], f)
    sc = np.linspace(1.0, 0.1, len(odd)).astype(f)
    for thr in (0.0, 1e-6, 0.5, 1.0):
        assert np.array_equal(emu.nms(odd, sc, thr), oracle.nms(odd, sc, thr)), thr


def test_emu_nms_batched_segments_beyond_the_lds_sort():
    """segments with more than 8192 candidates (the reference's non-FPN PRE_NMS_TOP_N_TRAIN = 12000): per-
    segment radix sort instead of the in-LDS bitonic network; ragged segment lengths."""
    segs = [synth.nms_boxes(n, seed=20 + i) for i, n in enumerate((8300, 700, 9000))]
    boxes = np.concatenate([x for x, _ in segs])
    scores = np.concatenate([y for _, y in segs])
    offs = np.cumsum([0] + [len(y) for _, y in segs]).astype(np.int32)
    km, num = emu.nms_batched(boxes, scores, offs, 9000, 0.7, mask=True)
    for i, (x, y) in enumerate(segs):
        ref = oracle.nms(x, y, 0.7)
        want = np.zeros(len(y), np.uint8)
        want[ref] = 1
        assert num[i] == len(ref) and np.array_equal(km[offs[i]:offs[i + 1]], want)


# ================================================================================ focal loss
@pytest.mark.parametrize("gamma,alpha,C", [(2.0, 0.25, 80), (1.5, 0.4, 7), (0.0, 0.5, 3)])
def test_emu_focal_vs_oracle(gamma, alpha, C):
    logits, targets = synth.focal_inputs(900, C)
    ref = oracle.sigmoid_focal_loss_forward(logits, targets, gamma, alpha)
    out = emu.focal_forward(logits, targets, gamma, alpha)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-6)
    out2, tot = emu.focal_forward(logits, targets, gamma, alpha, with_sum=True)
    assert np.array_equal(out, out2)
    assert abs(tot - float(ref.astype(np.float64).sum())) <= 1e-4 * max(1.0, abs(float(ref.sum())))
    d = np.random.RandomState(1).rand(*logits.shape).astype(np.float32)
    np.testing.assert_allclose(emu.focal_backward(logits, targets, d, gamma, alpha),
                               oracle.sigmoid_focal_loss_backward(logits, targets, d, gamma, alpha), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(emu.focal_backward(logits, targets, np.float32(0.37), gamma, alpha),
                               oracle.sigmoid_focal_loss_backward(logits, targets, np.full_like(logits, 0.37), gamma, alpha),
                               rtol=1e-4, atol=1e-6)


# ================================================================================ fused FrozenBN
@pytest.mark.parametrize("shape", [(2, 8, 25, 42), (1, 5, 7, 9), (3, 4, 1, 1)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True), (False, True)])
def test_emu_frozen_bn(shape, relu, res):
    rng = np.random.RandomState(3)
    x = rng.randn(*shape).astype(np.float32)
    r = rng.randn(*shape).astype(np.float32) if res else None
    scale = (rng.rand(shape[1]) + 0.5).astype(np.float32)
    bias = rng.randn(shape[1]).astype(np.float32)
    y = emu.frozen_bn_forward(x, scale, bias, r, relu)
    want = x * scale[None, :, None, None] + bias[None, :, None, None]
    if res:
        want = want + r
    if relu:
        want = np.maximum(want, 0)
    np.testing.assert_allclose(y, want, rtol=1e-6, atol=1e-6)
    gy = rng.randn(*shape).astype(np.float32)
    gx, gr = emu.frozen_bn_backward(gy, y, scale, relu, res)
    g = np.where(y > 0, gy, 0) if relu else gy
    np.testing.assert_allclose(gx, g * scale[None, :, None, None], rtol=1e-6, atol=1e-6)
    if res:
        np.testing.assert_allclose(gr, g, rtol=0, atol=0)


# ================================================================================ ROIPool
@pytest.mark.parametrize("impl", [1, 3])
def test_emu_roi_pool_backward_owner_and_scatter_forms(impl):
    """csrc/roi_pool.hip: the plane-owner backward (one workgroup sums the (image, channel) plane in LDS and writes it once;
    tuning roi_bwd_impl = 1 forces it for this small launch) and the atomic scatter (= 3) against the oracle — 300 ROIs over
    three images (two rounds of the owner kernel's ROI search, an image with no ROI at all), empty bins, overlapping windows
    that share an argmax pixel, and accumulation onto an existing gradient"""
    rng = np.random.RandomState(21)
    N, C, H, W, K = 3, 5, 19, 27, 300
    x = rng.randn(N, C, H, W).astype(np.float32)
    wh = rng.uniform(4, 300, (K, 2))
    xy = rng.uniform(-20, [W * 16 - 8, H * 16 - 8], (K, 2))
    rois = np.concatenate([rng.randint(0, 2, (K, 1)), xy, xy + wh], 1).astype(np.float32)      # image 2 has no ROI
    ref, ramax = oracle.roi_pool_forward(x, rois, 1.0 / 16, 7, 7)
    out, amax = emu.roi_pool_forward(x, rois, 1.0 / 16, 7, 7)
    assert np.array_equal(out, ref) and np.array_equal(amax, ramax) and (ramax == -1).any()
    g = rng.randn(*ref.shape).astype(np.float32)
    want = oracle.roi_pool_backward(g, rois, ramax, N, C, H, W)
    emu.tuning_set("roi_bwd_impl", impl)
    try:
        got = emu.roi_pool_backward(g, rois, amax, N, C, H, W)
        base = rng.randn(N, C, H, W).astype(np.float32)
        acc = emu.roi_pool_backward(g, rois, amax, N, C, H, W, into=base)
    finally:
        emu.tuning_set("roi_bwd_impl", 0)
    tol = 1e-5 * max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() <= tol and not got[2].any()
    assert np.abs(acc - (want + base)).max() <= 2 * tol


# ================================================================================ deformable PS-ROI pooling
@pytest.mark.parametrize("no_trans,ncls,D,G,P,part,S,std", [
    (True, 1, 8, 3, 3, 3, 4, 0.0), (False, 1, 8, 3, 7, 7, 4, 0.1), (False, 4, 8, 2, 7, 4, 2, 0.1)])
def test_emu_deform_psroi_pool(no_trans, ncls, D, G, P, part, S, std):
    rng = np.random.RandomState(11 + P)
    N, H, W, K = 2, 20, 30, 24
    data = rng.randn(N, D * G * G, H, W).astype(np.float32)
    x1 = rng.uniform(-20, W * 16 - 30, K)
    y1 = rng.uniform(-20, H * 16 - 30, K)
    rois = np.stack([rng.randint(0, N, K), x1, y1, x1 + rng.uniform(2, 300, K), y1 + rng.uniform(2, 250, K)], 1).astype(np.float32)
    trans = None if no_trans else (rng.randn(K, 2 * ncls, part, part) * 1.5).astype(np.float32)
    out, cnt = emu.psroi_forward(data, rois, trans, no_trans, 1 / 16, D, G, P, part, S, std)
    ro, rc = oracle.deform_psroi_pool_forward(data, rois, trans, no_trans, 1 / 16, D, G, P, part, S, std)
    np.testing.assert_allclose(out, ro, rtol=1e-4, atol=1e-5)
    assert np.array_equal(cnt, rc)
    g = rng.randn(*out.shape).astype(np.float32)
    dg, tg = emu.psroi_backward(g, data, rois, trans, cnt, no_trans, 1 / 16, D, G, P, part, S, std)
    rdg, rtg = oracle.deform_psroi_pool_backward(g, data, rois, trans, rc, no_trans, 1 / 16, D, G, P, part, S, std, acc64=True)
    np.testing.assert_allclose(dg, rdg, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(rdg).max()))
    if not no_trans:
        np.testing.assert_allclose(tg, rtg, rtol=1e-3, atol=1e-3 * max(1.0, np.abs(rtg).max()))


@pytest.mark.parametrize("groups", [None, "3", "7"])
def test_emu_roi_align_backward_roi_list_split(groups):
    """small maps split the ROI list over blockIdx.y (partials combined with atomics on the
    pre-zeroed map): automatic (K = 100 on a 14x14 map -> 4 groups) and forced group counts,
    with and without the accumulate flag."""
    emu.tuning_set("roi_bwd_impl", 2)
    if groups:
        emu.tuning_set("roi_bwd_groups", int(groups))
    inp, rois, scale = synth.cfg1_roi_align(K=100, C=6)
    rois = np.concatenate([rois, _edge_rois()])
    for (ph, pw, sr) in ((7, 7, 2), (14, 14, 2), (3, 5, 0)):
        g = np.random.RandomState(1).randn(rois.shape[0], 6, ph, pw).astype(np.float32)
        ref = oracle.roi_align_backward(g, rois, scale, ph, pw, *inp.shape, sr, acc64=True)
        tol = 1e-5 * max(1.0, np.abs(ref).max())
        assert np.abs(emu.roi_align_backward(g, rois, scale, ph, pw, *inp.shape, sr) - ref).max() <= tol
        base = np.random.RandomState(2).randn(*inp.shape).astype(np.float32)
        acc = emu.roi_align_backward(g, rois, scale, ph, pw, *inp.shape, sr, into=base)
        assert np.abs(acc - (base + ref)).max() <= 2 * tol


@pytest.mark.parametrize("ct", ["4", "16"])
def test_emu_roi_align_backward_lane_walk(ct):
    """per-lane bin-range walk of the scan pixel-owner kernel on ragged ROIs / bin shapes, both channel chunkings."""
    emu.tuning_set("roi_bwd_impl", 2)
    emu.tuning_set("roi_bwd_scan_ct", int(ct))
    rng = np.random.RandomState(31)
    N, C, H, W = 2, 9, 27, 70
    K = 90
    x1 = rng.uniform(-20, 270, K)
    y1 = rng.uniform(-20, 100, K)
    rois = np.stack([rng.randint(0, N, K), x1, y1, x1 + rng.uniform(0.2, 120, K), y1 + rng.uniform(0.2, 60, K)],
                    1).astype(np.float32)
    rois = np.concatenate([rois, _edge_rois()])
    for (ph, pw, sr) in ((7, 7, 2), (14, 14, 2), (5, 3, 0), (20, 20, 1)):
        g = rng.randn(rois.shape[0], C, ph, pw).astype(np.float32)
        ref = oracle.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr, acc64=True)
        out = emu.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr)
        assert np.abs(out - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("groups", [0, 1, 5])
def test_emu_roi_align_backward_acc_small_maps(groups):
    """acc backward (one small map with a workspace: the map of an (image, 16-channel chunk) accumulates in LDS, the ROI
    list is split over `groups` workgroups whose partial maps a second launch adds in group order): two images, ROI
    lists longer than one 8-entry round, slivers / outside / degenerate ROIs, a ragged last channel chunk, accumulate
    mode, run-to-run identical."""
    emu.tuning_set("roi_bwd_groups", groups)
    rng = np.random.RandomState(61)
    N, C, H, W = 2, 21, 14, 19
    K = 150
    x1 = rng.uniform(-10, 80, K)
    y1 = rng.uniform(-10, 60, K)
    rois = np.stack([rng.randint(0, N, K), x1, y1, x1 + rng.uniform(0.2, 70, K), y1 + rng.uniform(0.2, 50, K)], 1).astype(np.float32)
    rois = np.concatenate([rois, _edge_rois()])
    rois[:, 0] = np.minimum(rois[:, 0], N - 1)
    for (ph, pw, sr) in ((7, 7, 2), (14, 14, 2), (7, 7, 0)):
        g = rng.randn(rois.shape[0], C, ph, pw).astype(np.float32)
        ref = oracle.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr, acc64=True)
        emu.stats(reset=True)
        out = emu.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr)
        st = emu.stats()
        assert st.get("bwda.units", 0) > 0, "the acc kernel did not run"
        if groups != 1:
            assert st.get("bwda.combines", 0) == 1, st          # the combine launch ran
        tol = 1e-5 * max(1.0, np.abs(ref).max())
        assert np.abs(out - ref).max() <= tol
        assert np.array_equal(out, emu.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr))
        base = rng.randn(N, C, H, W).astype(np.float32)
        acc = emu.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr, into=base)
        assert np.abs(acc - (base + ref)).max() <= 2 * tol
    # a map beyond the plan (33 columns) keeps the other kernels
    g = rng.randn(rois.shape[0], C, 7, 7).astype(np.float32)
    emu.stats(reset=True)
    out = emu.roi_align_backward(g, rois, 0.25, 7, 7, N, C, 14, 33, 2)
    assert emu.stats().get("bwda.units", 0) == 0
    ref = oracle.roi_align_backward(g, rois, 0.25, 7, 7, N, C, 14, 33, 2, acc64=True)
    assert np.abs(out - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("seg,ct", [(None, 0), (8, 0), (9, 0), (None, 16), (8, 16)])
def test_emu_roi_align_backward_ring(seg, ct):
    """ring pixel-owner backward (pre-pass adjoint rows + per-tile hit lists in a poisoned workspace, LDS-DMA ring
    with counted waits — the emulation lands a staged piece only at a covering wait): ragged ROIs, slivers (bin
    ranges longer than 3), channel tails, accumulate mode; `seg` = 8 / 9 splits the crowded tiles into segments
    whose partial sums the last arriver combines; `ct` = 16: the 16-channel units of the 7x7 kernel (default: 32)."""
    emu.tuning_set("roi_bwd_impl", 1)
    emu.tuning_set("roi_bwd_ct", ct)
    if seg:
        emu.tuning_set("roi_bwd_seg", seg)
    rng = np.random.RandomState(51)
    N, C, H, W = 2, 37, 27, 70           # 37 channels: ragged last chunk; 27 x 70: partial edge tiles
    K = 90
    x1 = rng.uniform(-20, 270, K)
    y1 = rng.uniform(-20, 100, K)
    rois = np.stack([rng.randint(0, N, K), x1, y1, x1 + rng.uniform(0.2, 120, K), y1 + rng.uniform(0.2, 60, K)],
                    1).astype(np.float32)
    rois = np.concatenate([rois, _edge_rois()])
    for (ph, pw, sr) in ((7, 7, 2), (14, 14, 2), (7, 7, 0), (14, 14, 1)):
        g = rng.randn(rois.shape[0], C, ph, pw).astype(np.float32)
        ref = oracle.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr, acc64=True)
        emu.stats(reset=True)
        out = emu.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr)
        st = emu.stats()
        assert st.get("bwdr.units", 0) > 0, "the ring kernel did not run"
        if seg:
            assert st.get("bwdr.combines", 0) > 0, "expected split tiles"
        tol = 1e-5 * max(1.0, np.abs(ref).max())
        assert np.abs(out - ref).max() <= tol
        assert np.array_equal(out, emu.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr))   # run-to-run identical
        base = rng.randn(N, C, H, W).astype(np.float32)
        acc = emu.roi_align_backward(g, rois, 0.25, ph, pw, N, C, H, W, sr, into=base)
        assert np.abs(acc - (base + ref)).max() <= 2 * tol


def test_emu_roi_align_backward_ring_crowded_tiles_and_fpn_levels():
    """a tile hit by hundreds of ROIs: the split is capped at 8 segments, each longer than one LDS round of hit
    entries (64); a refused split (more crowded tiles than partial-sum slots) leaves long unsplit lists; the FPN
    entry point with an empty level."""
    rng = np.random.RandomState(52)
    N, C, H, W = 1, 16, 12, 40
    K = 700
    x1 = rng.uniform(0, 30, K)
    y1 = rng.uniform(0, 8, K)
    rois = np.stack([np.zeros(K), x1, y1, x1 + rng.uniform(1, 10, K), y1 + rng.uniform(1, 4, K)], 1).astype(np.float32)
    g = rng.randn(K, C, 7, 7).astype(np.float32)
    ref = oracle.roi_align_backward(g, rois, 1.0, 7, 7, N, C, H, W, 2, acc64=True)
    tol = 2e-5 * max(1.0, np.abs(ref).max())
    emu.tuning_set("roi_bwd_impl", 1)
    emu.tuning_set("roi_bwd_seg", 8)
    emu.stats(reset=True)
    out = emu.roi_align_backward(g, rois, 1.0, 7, 7, N, C, H, W, 2)
    st = emu.stats()
    assert st.get("bwdr.combines", 0) > 0 and st.get("bwdr.rounds", 0) > 0, st
    assert np.abs(out - ref).max() <= tol
    emu.tuning_set("roi_bwd_seg", 0)
    out = emu.roi_align_backward(g, rois, 1.0, 7, 7, N, C, H, W, 2)
    assert np.abs(out - ref).max() <= tol
    emu.tuning_set("roi_bwd_impl", 2)
    emu.tuning_set("roi_bwd_groups", 1)
    scan = emu.roi_align_backward(g, rois, 1.0, 7, 7, N, C, H, W, 2)
    assert np.abs(scan - ref).max() <= tol
    emu.tuning_set("roi_bwd_impl", 1)
    emu.tuning_set("roi_bwd_groups", 0)
    rng = np.random.RandomState(53)
    # multi-level entry, one level without any ROI
    shapes = [(2, 8, 50, 84), (2, 8, 25, 42), (2, 8, 13, 21), (2, 8, 7, 11)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    rois = synth.fpn_rois(seed=4, per_image=40, smin=8, smax=300)
    rois[:, 1:] *= 0.25
    lv = synth.level_map(rois * np.array([1, 4, 4, 4, 4], np.float32))
    lv[lv == 1] = 2
    gg = rng.randn(rois.shape[0], 8, 14, 14).astype(np.float32)
    outs = emu.roi_align_fpn_backward(gg, rois, lv, shapes, scales, 14, 14, 2)
    for l in range(4):
        idx = np.nonzero(lv == l)[0]
        want = oracle.roi_align_backward(gg[idx], rois[idx], scales[l], 14, 14, *shapes[l], 2, acc64=True) \
            if idx.size else np.zeros(shapes[l], np.float32)
        assert np.abs(outs[l] - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("gi", range(len(DCN_GEOMS)))
@pytest.mark.parametrize("modulated", [False, True])
def test_emu_deformable_col2im_ell_variant(gi, modulated):
    """fixed-width (ELL) inverted index (the default for fp32 and large maps)."""
    g = DCN_GEOMS[gi]
    x, off, mask = _dcn_case(g, modulated)
    k, p, s, d, dg = g["k"], g["pad"], g["stride"], g["dil"], g["dg"]
    geo = dict(kh=k, kw=k, pad=(p, p), stride=(s, s), dil=(d, d), dg=dg)
    ncol = oracle.deformable_im2col(x, off, mask, **geo).shape
    gcol = np.random.RandomState(9).randn(*ncol).astype(np.float32)
    ref = oracle.deformable_col2im(gcol, off, mask, *x.shape, **geo)
    out = emu.deformable_col2im(gcol, off, mask, *x.shape, mode="ell", **geo)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))
    base = np.random.RandomState(2).randn(*x.shape).astype(np.float32)
    np.testing.assert_allclose(emu.deformable_col2im(gcol, off, mask, *x.shape, mode="ell", into=base, **geo),
                               base + ref, rtol=1e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))


def test_emu_deformable_col2im_ell_overflow():
    """offsets that pile every sampling point of a tap onto the same pixel: far more than 8 entries per
    (pixel, tap) -> the overflow list + atomic kernel carry the rest."""
    B, C, H, W, k = 1, 5, 9, 11, 3
    geo = dict(kh=k, kw=k, pad=(1, 1), stride=(1, 1), dil=(1, 1), dg=1)
    rng = np.random.RandomState(3)
    off = np.zeros((B, 2 * k * k, H, W), np.float32)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for t in range(k * k):
        i, j = t // k, t % k
        off[0, 2 * t] = 4.3 - (ys - 1 + i)      # every sample of tap t lands at (4.3, 5.6)
        off[0, 2 * t + 1] = 5.6 - (xs - 1 + j)
    off += rng.randn(*off.shape).astype(np.float32) * 0.05
    gcol = rng.randn(C * k * k, B * H * W).astype(np.float32)
    ref = oracle.deformable_col2im(gcol, off, None, B, C, H, W, **geo)
    for mode in ("ell", "gather", "scatter"):
        out = emu.deformable_col2im(gcol, off, None, B, C, H, W, mode=mode, **geo)
        np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref).max()), err_msg=mode)


@pytest.mark.parametrize("case", [dict(B=1, C=3, H=37, W=70, k=3, stride=1, pad=1, dil=1, dg=1, sigma=2.0),      # several 16 x 32 tiles
                                  dict(B=2, C=2, H=21, W=45, k=3, stride=1, pad=1, dil=1, dg=1, sigma=12.0),     # most taps leave the scan window
                                  dict(B=1, C=4, H=40, W=41, k=3, stride=2, pad=3, dil=3, dg=2, sigma=5.0),      # stride, dilation, two offset groups
                                  dict(B=1, C=2, H=19, W=33, k=1, stride=1, pad=0, dil=1, dg=1, sigma=9.0),      # 1 x 1 kernel
                                  dict(B=1, C=2, H=50, W=40, k=3, stride=1, pad=6, dil=6, dg=1, sigma=3.0)])     # padding beyond the margin
def test_emu_deformable_index_tile_owner_build(case):
    """the tile-owner build of the inverted index (LDS lists, no global atomics): maps of several tiles, displacements
    far beyond the scanned window (those corners go through the overflow list), strides / dilations / padding — same
    gradient as the oracle and as the scatter build it replaces, run-to-run identical when nothing overflows"""
    g = case
    rng = np.random.RandomState(17)
    k, p_, s_, d, dg = g["k"], g["pad"], g["stride"], g["dil"], g["dg"]
    Ho = (g["H"] + 2 * p_ - (d * (k - 1) + 1)) // s_ + 1
    Wo = (g["W"] + 2 * p_ - (d * (k - 1) + 1)) // s_ + 1
    off = (rng.randn(g["B"], 2 * dg * k * k, Ho, Wo) * g["sigma"]).astype(np.float32)
    mask = rng.uniform(0.2, 1.0, (g["B"], dg * k * k, Ho, Wo)).astype(np.float32)
    geo = dict(kh=k, kw=k, pad=(p_, p_), stride=(s_, s_), dil=(d, d), dg=dg)
    gcol = rng.randn(g["C"] * k * k, g["B"] * Ho * Wo).astype(np.float32)
    shape = (g["B"], g["C"], g["H"], g["W"])
    ref = oracle.deformable_col2im(gcol, off, mask, *shape, **geo)
    tol = dict(rtol=1e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))
    out = emu.deformable_col2im(gcol, off, mask, *shape, mode="ell", **geo)
    np.testing.assert_allclose(out, ref, **tol)
    try:
        emu.tuning_set("dcn_ell_build", 1)
        np.testing.assert_allclose(emu.deformable_col2im(gcol, off, mask, *shape, mode="ell", **geo), ref, **tol)
    finally:
        emu.tuning_set("dcn_ell_build", 0)
    if g["sigma"] <= 3.0:
        assert np.array_equal(out, emu.deformable_col2im(gcol, off, mask, *shape, mode="ell", **geo))


def test_emu_deformable_channels_last_input_gradient_with_overflowing_pixels():
    """every sampling point of every tap lands on the same four pixels: 64 contributions per (pixel, tap) against the 8 slots
    of the inverted index — the rest goes through the overflow list (packed atomics behind the gather); both forms of the
    input gradient (col2im gather of the column gradient / transposed sampling + GEMM) against the oracle"""
    rng = np.random.RandomState(5)
    B, C, H, W, Cout, k = 1, 64, 8, 8, 64, 3
    x = rng.randn(B, C, H, W).astype(np.float32)
    wgt = (rng.randn(Cout, C, k, k) * 0.1).astype(np.float32)
    off = np.zeros((B, 2 * k * k, H, W), np.float32)
    for i in range(k):
        for j in range(k):
            hh, ww = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
            off[0, 2 * (i * k + j)] = 3.3 - (hh - 1 + i)          # sampling row  = ho - pad + i + off_h = 3.3
            off[0, 2 * (i * k + j) + 1] = 4.6 - (ww - 1 + j)      # sampling col  = 4.6
    mask = rng.uniform(0.2, 1.0, (B, k * k, H, W)).astype(np.float32)
    go = rng.randn(B, Cout, H, W).astype(np.float32)
    rin = oracle.deform_conv_backward(x, off, mask, wgt, go, False, (1, 1), (1, 1), (1, 1), 1, 1)[0]
    assert np.count_nonzero(np.abs(rin).sum(1)) == 4
    for form in ("col2im", "transposed"):
        gin = emu.deformable_nhwc(x, off, mask, wgt, go, k, k, (1, 1), (1, 1), (1, 1), input_grad=form)[1]
        np.testing.assert_allclose(gin, rin, rtol=1e-4, atol=2e-4 * np.abs(rin).max(), err_msg=form)


# ================================================================================ deformable conv, channels-last pipeline
@pytest.mark.parametrize("geom", [dict(B=2, C=64, H=9, W=11, Cout=128, k=3, pad=1, stride=1, dil=1),
                                  dict(B=1, C=128, H=12, W=10, Cout=64, k=3, pad=2, stride=2, dil=2),
                                  dict(B=1, C=512, H=5, W=6, Cout=64, k=3, pad=1, stride=1, dil=1),      # two vectors per lane
                                  dict(B=2, C=64, H=7, W=8, Cout=256, k=1, pad=0, stride=1, dil=1)])
@pytest.mark.parametrize("modulated", [False, True])
def test_emu_deformable_channels_last_pipeline_vs_oracle(geom, modulated):
    """NHWC im2col, coordinate / mask gradients with the in-wave channel reduction, and the input gradient through the
    TRANSPOSED sampling operator (fixed-width inverted index) + GEMM — against the oracle's reference-order forward and
    backward (fp32)."""
    g = geom
    rng = np.random.RandomState(17)
    x, off, mask, wgt = synth.dcn_inputs(g["B"], g["C"], g["H"], g["W"], g["Cout"], g["k"], 1, modulated, seed=3)
    pad, stride, dil = (g["pad"],) * 2, (g["stride"],) * 2, (g["dil"],) * 2
    Ho = (g["H"] + 2 * g["pad"] - (g["dil"] * (g["k"] - 1) + 1)) // g["stride"] + 1
    Wo = (g["W"] + 2 * g["pad"] - (g["dil"] * (g["k"] - 1) + 1)) // g["stride"] + 1
    off = np.ascontiguousarray(off[:, :, :Ho, :Wo])
    mask = None if mask is None else np.ascontiguousarray(mask[:, :, :Ho, :Wo])
    go = rng.randn(g["B"], g["Cout"], Ho, Wo).astype(np.float32)
    out, gin, goff, gmask, gw = emu.deformable_nhwc(x, off, mask, wgt, go, g["k"], g["k"], pad, stride, dil)
    ref_out = oracle.deform_conv_forward(x, off, mask, wgt, None, pad, stride, dil, 1, 1)
    np.testing.assert_allclose(out, ref_out, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref_out).max()))
    rin, roff, rmask, rw, _ = oracle.deform_conv_backward(x, off, mask, wgt, go, False, pad, stride, dil, 1, 1)
    tol = lambda r: dict(rtol=1e-4, atol=2e-4 * max(1.0, np.abs(r).max()))  # noqa: E731
    np.testing.assert_allclose(gin, rin, **tol(rin))            # col2im gather of the column gradient (round 6, the default)
    gin_t = emu.deformable_nhwc(x, off, mask, wgt, go, g["k"], g["k"], pad, stride, dil, input_grad="transposed")[1]
    np.testing.assert_allclose(gin_t, rin, **tol(rin))          # transposed sampling of the output gradient + GEMM (rounds 3-5)
    np.testing.assert_allclose(goff, roff, **tol(roff))
    np.testing.assert_allclose(gw, rw, **tol(rw))
    if modulated:
        np.testing.assert_allclose(gmask, rmask, **tol(rmask))


def test_emu_deformable_transposed_sample_overflow_and_fp16():
    """offsets that pile every sampling point of a tap onto one pixel (more than 8 entries per (pixel, tap): the
    overflow list + atomic kernel), and fp16 storage of every intermediate."""
    B, C, H, W, Cout, k = 1, 128, 9, 11, 128, 3
    rng = np.random.RandomState(3)
    off = np.zeros((B, 2 * k * k, H, W), np.float32)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for t in range(k * k):
        i, j = t // k, t % k
        off[0, 2 * t] = 4.3 - (ys - 1 + i)      # every sample of tap t lands at (4.3, 5.6)
        off[0, 2 * t + 1] = 5.6 - (xs - 1 + j)
    off += rng.randn(*off.shape).astype(np.float32) * 0.05
    x = rng.randn(B, C, H, W).astype(np.float32)
    wgt = (rng.randn(Cout, C, k, k) / np.sqrt(C * k * k)).astype(np.float32)
    go = rng.randn(B, Cout, H, W).astype(np.float32)
    rin, roff, _, rw, _ = oracle.deform_conv_backward(x, off, None, wgt, go, False, (1, 1), (1, 1), (1, 1), 1, 1)
    for dt, rt in ((np.float32, 1e-4), (np.float16, 2e-2)):
        out, gin, goff, _, gw = emu.deformable_nhwc(x.astype(dt), off.astype(dt), None, wgt.astype(dt), go.astype(dt), k, k,
                                                    (1, 1), (1, 1), (1, 1))
        if dt == np.float16:      # the oracle on the fp16-rounded operands
            rin, roff, _, rw, _ = oracle.deform_conv_backward(x.astype(dt).astype(np.float32), off.astype(dt).astype(np.float32), None,
                                                              wgt.astype(dt).astype(np.float32), go.astype(dt).astype(np.float32),
                                                              False, (1, 1), (1, 1), (1, 1), 1, 1)
        np.testing.assert_allclose(gin, rin, rtol=rt, atol=rt * max(1.0, np.abs(rin).max()))
        np.testing.assert_allclose(goff.astype(np.float32), roff, rtol=rt, atol=2 * rt * max(1.0, np.abs(roff).max()))
        np.testing.assert_allclose(gw, rw, rtol=rt, atol=rt * max(1.0, np.abs(rw).max()))


# ================================================================================ fused deformable conv (MFMA)
@pytest.mark.parametrize("geom", [dict(B=2, C=32, H=9, W=11, Cout=40, k=3, pad=1, stride=1, dil=1, dg=1),
                                  dict(B=1, C=64, H=12, W=10, Cout=130, k=3, pad=2, stride=2, dil=2, dg=2),
                                  dict(B=1, C=64, H=7, W=9, Cout=32, k=3, pad=1, stride=1, dil=1, dg=1),      # 64-channel K-steps
                                  dict(B=1, C=256, H=6, W=7, Cout=64, k=3, pad=1, stride=1, dil=1, dg=1)])    # 128-channel K-steps
@pytest.mark.parametrize("modulated", [False, True])
def test_emu_deform_conv_forward_fused_mfma(geom, modulated):
    """implicit-GEMM forward (B tile interpolated in LDS, v_mfma_f32_32x32x16_f16 fragment arithmetic emulated lane
    by lane): ragged Cout / pixel tiles, two deformable groups, stride / dilation, bias; vs the oracle's
    im2col + GEMM on the same fp16-rounded inputs."""
    g = geom
    x, off, mask, wgt = synth.dcn_inputs(g["B"], g["C"], g["H"], g["W"], g["Cout"], g["k"], g["dg"], modulated, seed=21)
    Ho = (g["H"] + 2 * g["pad"] - (g["dil"] * (g["k"] - 1) + 1)) // g["stride"] + 1
    Wo = (g["W"] + 2 * g["pad"] - (g["dil"] * (g["k"] - 1) + 1)) // g["stride"] + 1
    off = np.ascontiguousarray(off[:, :, :Ho, :Wo])
    mask = None if mask is None else np.ascontiguousarray(mask[:, :, :Ho, :Wo])
    bias = np.random.RandomState(2).randn(g["Cout"]).astype(np.float32) if modulated else None
    h = lambda a: None if a is None else a.astype(np.float16)  # noqa: E731
    geo = dict(pad=(g["pad"],) * 2, stride=(g["stride"],) * 2, dil=(g["dil"],) * 2)
    out = emu.deform_conv_forward_fused(h(x), h(wgt), h(off), h(mask), h(bias), dg=g["dg"], **geo)
    assert out is not None and out.shape == (g["B"], g["Cout"], Ho, Wo)
    f = lambda a: None if a is None else a.astype(np.float16).astype(np.float32)  # noqa: E731
    ref = oracle.deform_conv_forward(f(x), f(off), f(mask), f(wgt), f(bias), group=1, dg=g["dg"], **geo)
    assert np.abs(out.astype(np.float32) - ref).max() <= 2e-2 * np.abs(ref).max()


def test_emu_roi_align_backward_prepare_then_prepared_equals_one_call():
    """the ring backward in two calls (pre-pass at forward time into a kept workspace, main kernel alone at backward time)
    gives the bits of the one-call entry point; shapes the ring plan does not serve answer DETOPS_EUNSUPPORTED"""
    rng = np.random.RandomState(71)
    shapes = [(2, 19, 50, 84), (2, 19, 25, 42), (2, 19, 13, 21), (2, 19, 7, 11)]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    rois = synth.fpn_rois(seed=9, per_image=120, smin=8, smax=300)
    rois[:, 1:] *= 0.25
    lv = synth.level_map(rois)
    emu.tuning_set("roi_bwd_impl", 1)      # the ring also for these small test maps
    for ph in (7, 14):
        g = rng.randn(rois.shape[0], 19, ph, ph).astype(np.float32)
        one = emu.roi_align_fpn_backward(g, rois, lv, shapes, scales, ph, ph, 2)
        two = emu.roi_align_fpn_backward_two_calls(g, rois, lv, shapes, scales, ph, ph, 2)
        assert two is not None
        for a, b in zip(one, two):
            assert np.array_equal(a, b)
    emu.tuning_set("roi_bwd_impl", 2)
    assert emu.roi_align_fpn_backward_two_calls(g, rois, lv, shapes, scales, 14, 14, 2) is None
    emu.tuning_set("roi_bwd_impl", 0)      # auto: these maps are an under-filled launch -> the one-call kernels
    assert emu.roi_align_fpn_backward_two_calls(g, rois, lv, shapes, scales, 14, 14, 2) is None
