"""Host-logic parity of the model surface against fixtures produced by the REFERENCE's own Python
modules (tests/golden/make_golden_model.py): anchors, Matcher, BoxList ops, IoU, LevelMapper, target
assignment (RPN / RetinaNet / Fast R-CNN), RPN loss, mask targets, proposal selection, LR schedule.
Operators that only exist as HIP kernels are replaced by the oracle through tests/cpu_shim.py."""
import os

import numpy as np
import pytest
import torch

import cpu_shim
from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler
from maskrcnn_benchmark.modeling.box_coder import BoxCoder
from maskrcnn_benchmark.modeling.matcher import Matcher
from maskrcnn_benchmark.modeling.poolers import LevelMapper
from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator, generate_anchors
from maskrcnn_benchmark.structures.bounding_box import BoxList
from maskrcnn_benchmark.structures.boxlist_ops import boxlist_iou
from maskrcnn_benchmark.structures.image_list import ImageList, to_image_list

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def T(a, dev="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


class _Dev(str):
    """a device string that also names the cpu_shim backend serving the HIP-only operators"""
    backend = "oracle"


def _emu(backend):
    d = _Dev("cpu")
    d.backend = backend
    return d


@pytest.fixture(params=["cpu", pytest.param(_emu("emu-device"), id="cpu-device-branches"),
                        pytest.param(_emu("emu-lib"), id="cpu-product-wrappers"), pytest.param("cuda", marks=pytest.mark.gpu)])
def dev(request):
    """The reference-generated fixtures are checked four times: on the CPU (host logic; HIP-only operators
    replaced by the oracle through cpu_shim); on the CPU with the model taking its DEVICE branches (fused label /
    sampler / sampled-slot / decode launches, served by the HIP sources under the host emulation: cpu_shim backend
    "emu-device"); on the CPU through the product's own `_C` wrappers over the emulation library (backend "emu-lib":
    nothing of `_C` replaced); and, in the `-m gpu` suite, on the device through the real HIP kernels (no shim)."""
    backend = getattr(request.param, "backend", None)
    if backend in ("emu-device", "emu-lib"):
        with cpu_shim.install(backend):          # for the whole test: target assignment runs outside `_shim` blocks
            yield request.param
    else:
        yield request.param


def _shim(dev):
    import contextlib
    return cpu_shim.install(getattr(dev, "backend", "oracle")) if dev == "cpu" else contextlib.nullcontext()


def N_(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ config
def test_config_files_load_and_reject_unknown_keys(tmp_path):
    from maskrcnn_benchmark.config import cfg
    from maskrcnn_benchmark.engine.bench_step import CONFIG_DIR, load_cfg
    for f in ("e2e_mask_rcnn_R_50_FPN_1x.yaml", "e2e_faster_rcnn_R_50_FPN_1x.yaml", "e2e_mask_rcnn_R_101_FPN_1x.yaml",
              "retinanet/retinanet_R-50-FPN_1x.yaml"):
        c = load_cfg(f)
        assert c.MODEL.RPN.ANCHOR_STRIDE == (4, 8, 16, 32, 64)
        assert c.is_frozen()
    c = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml", ["MODEL.RESNETS.STAGE_WITH_DCN", "(False,True,True,True)",
                                                   "SOLVER.BASE_LR", "0.01", "DTYPE", "float16"])
    assert c.MODEL.RESNETS.STAGE_WITH_DCN == (False, True, True, True) and c.SOLVER.BASE_LR == 0.01
    assert c.MODEL.ROI_MASK_HEAD.RESOLUTION == 28 and c.MODEL.MASK_ON is True
    assert cfg.MODEL.MASK_ON is False  # the global default tree is untouched
    bad = tmp_path / "bad.yaml"
    bad.write_text("MODEL:\n  NO_SUCH_KEY: 1\n")
    with pytest.raises(KeyError):
        load_cfg(str(bad))
    with pytest.raises(AttributeError):
        c.SOLVER.BASE_LR = 1.0
    assert os.path.isdir(CONFIG_DIR)


# ------------------------------------------------------------------ anchors
def test_anchors_match_reference():
    g = load("model_anchors.npz")
    for i, (stride, size) in enumerate(zip((4, 8, 16, 32, 64), (32, 64, 128, 256, 512))):
        np.testing.assert_array_equal(generate_anchors(stride, (size,), (0.5, 1.0, 2.0)).float().numpy(), g["rpn_cell_%d" % i])
    for i, (stride, size) in enumerate(zip((8, 16, 32, 64, 128), (32, 64, 128, 256, 512))):
        sizes = tuple(size * 2 ** (k / 3.0) for k in range(3))
        np.testing.assert_array_equal(generate_anchors(stride, sizes, (0.5, 1.0, 2.0)).float().numpy(), g["retina_cell_%d" % i])
    ag = AnchorGenerator(sizes=((32,), (64,), (128,)), anchor_strides=(4, 8, 16), straddle_thresh=0)
    feats = [torch.zeros(1, 1, h, w) for h, w in g["grids"]]
    il = ImageList(torch.zeros(2, 3, 20, 28), [tuple(s) for s in g["image_sizes"]])
    anchors = ag(il, feats)
    for l in range(3):
        np.testing.assert_array_equal(anchors[0][l].bbox.numpy(), g["grid_%d" % l])
        for i in range(2):
            np.testing.assert_array_equal(anchors[i][l].get_field("visibility").numpy(), g["vis_%d_%d" % (i, l)])


# ------------------------------------------------------------------ matcher
def test_matcher_matches_reference_including_ties_and_batched(dev):
    g = load("model_matcher.npz")
    for c, (hi, lo, lq) in enumerate(g["cfgs"]):
        m = Matcher(float(hi), float(lo), allow_low_quality_matches=bool(lq))
        for k in range(4):
            q = T(g["q_%d_%d" % (c, k)], dev)
            np.testing.assert_array_equal(N_(m(q)), g["m_%d_%d" % (c, k)])
            # batched + padded rows give the same answer
            pad = torch.full((3, q.shape[1]), -1.0, device=dev)
            qb = torch.stack([torch.cat([q, pad]), torch.cat([q, pad])])
            rv = torch.zeros(2, q.shape[0] + 3, dtype=torch.bool, device=dev)
            rv[:, :q.shape[0]] = True
            out = m(qb, rv)
            np.testing.assert_array_equal(N_(out[0]), g["m_%d_%d" % (c, k)])
            np.testing.assert_array_equal(N_(out[1]), g["m_%d_%d" % (c, k)])
    with pytest.raises(ValueError):
        Matcher(0.5, 0.5)(torch.zeros(0, 4))


# ------------------------------------------------------------------ BoxList
def test_boxlist_ops_match_reference():
    g = load("model_boxlist.npz")
    W, H = (int(v) for v in g["size"])
    bl = BoxList(T(g["boxes"]), (W, H), "xyxy")
    tol = dict(rtol=0, atol=1e-4)
    np.testing.assert_allclose(bl.convert("xywh").bbox.numpy(), g["xywh"], **tol)
    np.testing.assert_allclose(bl.convert("xywh").convert("xyxy").bbox.numpy(), g["xywh_back"], **tol)
    np.testing.assert_allclose(bl.resize((W * 2, H * 2)).bbox.numpy(), g["resize_same"], **tol)
    np.testing.assert_allclose(bl.resize((480, 250)).bbox.numpy(), g["resize_diff"], **tol)
    np.testing.assert_allclose(bl.transpose(0).bbox.numpy(), g["flip_lr"], **tol)
    np.testing.assert_allclose(bl.transpose(1).bbox.numpy(), g["flip_tb"], **tol)
    np.testing.assert_allclose(bl.crop((30, 20, 250, 160)).bbox.numpy(), g["crop"], **tol)
    np.testing.assert_allclose(bl.area().numpy(), g["area"], rtol=1e-6)
    np.testing.assert_allclose(bl.convert("xywh").area().numpy(), g["area_xywh"], rtol=1e-5)
    np.testing.assert_allclose(BoxList(T(g["clip_in"]).clone(), (W, H)).clip_to_image(False).bbox.numpy(), g["clip_out"], **tol)
    np.testing.assert_allclose(BoxList(T(g["clip_in"]).clone(), (W, H)).clip_to_image(True).bbox.numpy(), g["clip_kept"], **tol)
    np.testing.assert_allclose(boxlist_iou(bl, BoxList(T(g["boxes2"]), (W, H))).numpy(), g["iou"], rtol=1e-5, atol=1e-7)
    lv = LevelMapper(2, 5)([BoxList(T(g["lm_boxes"]), (1344, 800))])
    np.testing.assert_array_equal(lv.numpy(), g["lm_levels"])
    with pytest.raises(ValueError):
        BoxList(torch.zeros(3, 5), (10, 10))
    f = bl.copy_with_fields([])
    f.add_field("labels", torch.arange(len(f)))
    assert f[torch.tensor([3, 1])].get_field("labels").tolist() == [3, 1]


def test_to_image_list_pads_to_divisor():
    il = to_image_list([torch.ones(3, 30, 50), torch.ones(3, 41, 33)], 32)
    assert tuple(il.tensors.shape) == (2, 3, 64, 64)
    assert [tuple(s) for s in il.image_sizes] == [(30, 50), (41, 33)]
    assert float(il.tensors[0, :, 30:].abs().sum()) == 0 and float(il.tensors[1, :, :, 33:].abs().sum()) == 0


# ------------------------------------------------------------------ target assignment + RPN loss
def _targets_setup(g, dev="cpu"):
    H, W = (int(v) for v in g["canvas"])
    ag = AnchorGenerator(sizes=((32,), (64,), (128,)), anchor_strides=(8, 16, 32), straddle_thresh=0).to(dev)
    feats = [torch.zeros(2, 1, H // s, W // s, device=dev) for s in (8, 16, 32)]
    il = ImageList(torch.zeros(2, 3, H, W, device=dev), [tuple(int(v) for v in s) for s in g["image_sizes"]])
    anchors = ag(il, feats)
    targets = []
    for i, (h, w) in enumerate(il.image_sizes):
        t = BoxList(T(g["gt_%d" % i], dev), (w, h))
        t.add_field("labels", T(g["gt_labels_%d" % i], dev))
        targets.append(t)
    return anchors, targets


class _FixedSampler(object):
    def __init__(self, n_pos, n_neg):
        self.n_pos, self.n_neg = n_pos, n_neg

    def _masks(self, labels):
        pos, neg = labels >= 1, labels == 0
        return pos & (pos.cumsum(-1) <= self.n_pos), neg & (neg.cumsum(-1) <= self.n_neg)


def test_rpn_and_retinanet_target_assignment_and_rpn_loss_match_reference(dev):
    from maskrcnn_benchmark.modeling.rpn.loss import RPNLossComputation, generate_rpn_labels
    from maskrcnn_benchmark.modeling.rpn.retinanet.loss import generate_retinanet_labels
    g = load("model_targets.npz")
    anchors, targets = _targets_setup(g, dev)
    rpn = RPNLossComputation(Matcher(0.7, 0.3, True), _FixedSampler(16, 48), BoxCoder((1., 1., 1., 1.)), generate_rpn_labels)
    lab, reg = rpn.prepare_targets(anchors, targets)
    for i in range(2):
        np.testing.assert_array_equal(N_(lab[i]), g["rpn_labels_%d" % i])
        sel = g["rpn_labels_%d" % i] > 0  # regression targets only matter on positives; all are compared anyway
        np.testing.assert_allclose(N_(reg[i]), g["rpn_reg_%d" % i], rtol=1e-4, atol=1e-5)
        assert sel.sum() > 0
    obj = [T(g["objectness_%d" % l], dev) for l in range(3)]
    breg = [T(g["box_reg_%d" % l], dev) for l in range(3)]
    lo, lb = rpn(anchors, obj, breg, targets)
    np.testing.assert_allclose([float(lo), float(lb)], g["rpn_loss"], rtol=1e-5)
    ret = RPNLossComputation(Matcher(0.5, 0.4, True), None, BoxCoder((10., 10., 5., 5.)), generate_retinanet_labels)
    ret.copied_fields, ret.discard_cases = ["labels"], ["between_thresholds"]
    lab, reg = ret.prepare_targets(anchors, targets)
    for i in range(2):
        np.testing.assert_array_equal(N_(lab[i]), g["ret_labels_%d" % i])
        np.testing.assert_allclose(N_(reg[i]), g["ret_reg_%d" % i], rtol=1e-4, atol=1e-5)


def test_fast_rcnn_target_assignment_matches_reference(dev):
    from maskrcnn_benchmark.modeling.roi_heads.box_head.loss import FastRCNNLossComputation, stack_proposals
    g = load("model_targets.npz")
    _, targets = _targets_setup(g, dev)
    sizes = [(int(w), int(h)) for h, w in g["image_sizes"]]
    props = [BoxList(T(g["props_%d" % i], dev), sizes[i]) for i in range(2)]
    frc = FastRCNNLossComputation(Matcher(0.5, 0.5, False), BalancedPositiveNegativeSampler(32, 0.25), BoxCoder((10., 10., 5., 5.)))
    boxes, valid = stack_proposals(props)
    lab, reg, _ = frc.prepare_targets(boxes, valid, targets)
    for i in range(2):
        n = len(props[i])
        np.testing.assert_array_equal(N_(lab[i, :n]), g["frc_labels_%d" % i])
        np.testing.assert_allclose(N_(reg[i, :n]), g["frc_reg_%d" % i], rtol=1e-4, atol=1e-5)
        assert (lab[i, n:] == -1).all()
    # fixed-length sampling: positives first, <= 25 % positives, invalid slots flagged
    out = frc.subsample(props, targets)
    for i, s in enumerate(out):
        l, v = s.get_field("labels"), s.get_field("valid")
        assert len(s) == 32 and int((l > 0).sum()) <= 8
        npos = int((l > 0).sum())
        assert (l[:npos] > 0).all() and (l[npos:] <= 0).all()
        assert (l[~v] == -1).all() and (l[v] >= 0).all()
        # every sampled row is a real proposal with the label the assignment gave it
        for j in torch.nonzero(v).flatten().tolist():
            d = (boxes[i] - s.bbox[j]).abs().sum(1)
            k = int(d.argmin())
            assert float(d[k]) == 0 and int(lab[i, k]) == int(l[j])


# ------------------------------------------------------------------ samplers
def test_balanced_sampler_counts_and_uniformity():
    torch.manual_seed(0)
    s = BalancedPositiveNegativeSampler(64, 0.25)
    labels = torch.zeros(2, 500, dtype=torch.int64)
    labels[0, :40] = 3
    labels[0, 400:] = -1
    labels[1, :5] = 1
    pos, neg = s([labels[0], labels[1]])
    assert int(pos[0].sum()) == 16 and int(neg[0].sum()) == 48
    assert int(pos[1].sum()) == 5 and int(neg[1].sum()) == 59
    assert not (pos[0] & (labels[0] < 1)).any() and not (neg[0] & (labels[0] != 0)).any()
    hits = torch.zeros(40)
    for _ in range(300):
        p, _ = s([labels[0]])
        hits += p[0][:40].float()
    assert hits.min() > 60 and hits.max() < 180  # each positive chosen ~ 300*16/40 = 120 times
    idx, valid = s.sample_fixed(labels)
    assert idx.shape == (2, 64) and valid.all()
    few = torch.full((1, 100), -1, dtype=torch.int64)
    few[0, :10] = 0
    idx, valid = s.sample_fixed(few)
    assert int(valid.sum()) == 10 and (few[0][idx[0][valid[0]]] == 0).all()


def test_balanced_sampler_topk_threshold_variant(monkeypatch):
    """experimental DETOPS_SAMPLER=topk selection: same counts, candidates only, uniform."""
    import maskrcnn_benchmark.modeling.balanced_positive_negative_sampler as bs
    monkeypatch.setattr(bs, "_THRESHOLD_SELECT", True)
    torch.manual_seed(0)
    s = bs.BalancedPositiveNegativeSampler(64, 0.25)
    labels = torch.zeros(2, 500, dtype=torch.int64)
    labels[0, :40] = 3
    labels[0, 400:] = -1
    labels[1, :5] = 1
    pos, neg = s._masks(labels)
    assert int(pos[0].sum()) == 16 and int(neg[0].sum()) == 48
    assert int(pos[1].sum()) == 5 and int(neg[1].sum()) == 59
    assert not (pos & (labels < 1)).any() and not (neg & (labels != 0)).any()
    none = torch.full((1, 200), -1, dtype=torch.int64)
    none[0, :100] = 0                     # no positives at all
    p, q = s._masks(none)
    assert int(p.sum()) == 0 and int(q.sum()) == 64
    hits = torch.zeros(40)
    for _ in range(300):
        p, _ = s._masks(labels[:1])
        hits += p[0][:40].float()
    assert hits.min() > 60 and hits.max() < 180
    idx, valid = s.sample_fixed(labels)
    assert idx.shape == (2, 64) and valid.all()


# ------------------------------------------------------------------ mask targets
def test_mask_targets_match_reference_binary_mask_path(dev):
    from maskrcnn_benchmark.modeling.roi_heads.mask_head.loss import project_masks_on_boxes
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask
    g = load("model_masks.npz")
    masks, props, which = T(g["masks"], dev), T(g["props"], dev), T(g["which"], dev)
    out_u8 = N_(project_masks_on_boxes(masks, which, props, 28))
    out_f32 = N_(project_masks_on_boxes(masks.float(), which, props, 28))
    np.testing.assert_allclose(out_f32, g["targets_f32"], atol=2e-6)
    # integer masks truncate the interpolated value (a sum one ulp below 1.0 becomes 0): the
    # projection follows the CPU kernel operation by operation, so this is exact
    np.testing.assert_array_equal(out_u8, g["targets_u8"])
    # and the structures-level crop+resize (what the reference's loss calls per ROI) agrees too
    W, H = (int(v) for v in g["size"])
    seg = SegmentationMask(masks.float(), (W, H))
    for j in (0, 7, 39):
        r = N_(seg[int(which[j])].crop(props[j]).resize((28, 28)).get_mask_tensor())
        np.testing.assert_allclose(r, g["targets_f32"][j], atol=2e-6)


# ------------------------------------------------------------------ proposal selection
@pytest.mark.parametrize("tag", ["train", "train_perimg", "test"])
def test_rpn_proposal_selection_matches_reference(tag, dev):
    from maskrcnn_benchmark.modeling.rpn.inference import RPNPostProcessor
    g = load("model_proposals.npz")
    H, W = (int(v) for v in g["canvas"])
    ag = AnchorGenerator(sizes=((32,), (64,), (128,)), anchor_strides=(8, 16, 32), straddle_thresh=0).to(dev)
    feats = [torch.zeros(2, 1, H // s, W // s, device=dev) for s in (8, 16, 32)]
    il = ImageList(torch.zeros(2, 3, H, W, device=dev), [tuple(int(v) for v in s) for s in g["image_sizes"]])
    anchors = ag(il, feats)
    obj = [T(g["objectness_%d" % l], dev) for l in range(3)]
    breg = [T(g["box_reg_%d" % l], dev) for l in range(3)]
    pre, post, min_size, fpn_post, per_batch = (int(v) for v in g["%s_cfg" % tag])
    pp = RPNPostProcessor(pre, post, 0.7, min_size, BoxCoder((1., 1., 1., 1.)), fpn_post, bool(per_batch))
    pp.train(tag.startswith("train"))
    with _shim(dev):
        res = pp(anchors, obj, breg, None)
    for i, r in enumerate(res):
        s = r.get_field("objectness")
        b = r.bbox
        if r.has_field("valid"):
            v = r.get_field("valid")
            s, b = s[v], b[v]
        order = torch.argsort(s, descending=True, stable=True)
        np.testing.assert_allclose(N_(s[order]), g["%s_scores_%d" % (tag, i)], rtol=1e-6)
        np.testing.assert_allclose(N_(b[order]), g["%s_boxes_%d" % (tag, i)], rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------ evaluation post-processing
def _canon(r):
    s, l = N_(r.get_field("scores")), N_(r.get_field("labels"))
    order = np.lexsort((-s.astype(np.float64), l))
    return N_(r.bbox)[order], s[order], l[order]


@pytest.mark.parametrize("tag", ["a", "b"])
def test_box_head_postprocessor_matches_reference(tag, dev):
    """PostProcessor.forward (reference roi_heads/box_head/inference.py:42-149): the per-class NMS problems run
    as one segmented launch; detections equal the reference's per-class loop (set-wise, scores to 1e-6)."""
    from maskrcnn_benchmark.modeling.roi_heads.box_head.inference import PostProcessor
    g = load("model_postprocess.npz")
    thr, nms, dets = g["box_%s_cfg" % tag]
    pp = PostProcessor(float(thr), float(nms), int(dets), BoxCoder((10., 10., 5., 5.)))
    sizes = [tuple(int(v) for v in s) for s in g["box_sizes"]]
    props = [BoxList(T(g["box_props_%d" % i], dev), sizes[i], mode="xyxy") for i in range(2)]
    with _shim(dev):
        res = pp((T(g["box_logits"], dev), T(g["box_reg"], dev)), props)
    for i, r in enumerate(res):
        b, s, l = _canon(r)
        np.testing.assert_array_equal(l, g["box_%s_labels_%d" % (tag, i)])
        np.testing.assert_allclose(s, g["box_%s_scores_%d" % (tag, i)], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(b, g["box_%s_boxes_%d" % (tag, i)], rtol=1e-5, atol=1e-3)
    empty = pp.filter_results(torch.zeros(0, 36, device=dev), torch.zeros(0, 9, device=dev), (32, 32))
    assert len(empty) == 0 and empty.has_field("scores") and empty.has_field("labels")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_retinanet_postprocessor_matches_reference(tag, dev):
    """RetinaNetPostProcessor.forward (reference rpn/retinanet/inference.py:64-173)."""
    from maskrcnn_benchmark.modeling.rpn.retinanet.inference import RetinaNetPostProcessor
    g = load("model_postprocess.npz")
    H, W = (int(v) for v in g["ret_canvas"])
    sz = tuple(tuple(s * 2 ** (k / 3.0) for k in range(3)) for s in (32, 64, 128))
    ag = AnchorGenerator(sizes=sz, aspect_ratios=(0.5, 1.0, 2.0), anchor_strides=(8, 16, 32), straddle_thresh=-1).to(dev)
    feats = [torch.zeros(2, 1, H // s, W // s, device=dev) for s in (8, 16, 32)]
    il = ImageList(torch.zeros(2, 3, H, W, device=dev), [tuple(int(v) for v in s) for s in g["ret_image_sizes"]])
    anchors = ag(il, feats)
    thr, topn, nms, post = g["ret_%s_cfg" % tag]
    pp = RetinaNetPostProcessor(float(thr), int(topn), float(nms), int(post), 0, 6, BoxCoder((10., 10., 5., 5.)))
    cls = [T(g["ret_cls_%d" % l], dev) for l in range(3)]
    reg = [T(g["ret_reg_%d" % l], dev) for l in range(3)]
    with _shim(dev):
        res = pp(anchors, cls, reg)
    for i, r in enumerate(res):
        b, s, l = _canon(r)
        np.testing.assert_array_equal(l, g["ret_%s_labels_%d" % (tag, i)])
        np.testing.assert_allclose(s, g["ret_%s_scores_%d" % (tag, i)], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(b, g["ret_%s_boxes_%d" % (tag, i)], rtol=1e-5, atol=1e-3)


# ------------------------------------------------------------------ solver
def test_lr_schedule_and_smooth_l1_match_reference():
    from maskrcnn_benchmark.layers import smooth_l1_loss
    from maskrcnn_benchmark.solver import WarmupMultiStepLR
    g = load("model_solver.npz")
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=0.02)
    sched = WarmupMultiStepLR(opt, (30, 40), 0.1, warmup_factor=1.0 / 3, warmup_iters=10, warmup_method="linear")
    lrs = []
    for _ in range(50):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)
    a, b = T(g["sl1_a"]), T(g["sl1_b"])
    got = [float(smooth_l1_loss(a, b, beta=1.0 / 9, size_average=False)), float(smooth_l1_loss(a, b, beta=1.0, size_average=True)),
           float(smooth_l1_loss(a, b, beta=0.11, size_average=False))]
    np.testing.assert_allclose(got, g["sl1"], rtol=1e-6)
    with pytest.raises(ValueError):
        WarmupMultiStepLR(opt, (40, 30))


# ------------------------------------------------------------------ whole model on the CPU shim
@pytest.mark.parametrize("config", ["e2e_mask_rcnn_R_50_FPN_1x.yaml", "retinanet/retinanet_R-50-FPN_1x.yaml"])
def test_tiny_model_trains_on_cpu_shim(config, monkeypatch):
    import maskrcnn_benchmark.layers.sigmoid_focal_loss as sfl
    from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    from maskrcnn_benchmark.engine.ddp_step import TrainStep, make_overlapped_sgd
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = load_cfg(config, ["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150,
                            "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32, "MODEL.RESNETS.RES2_OUT_CHANNELS", 16,
                            "MODEL.RESNETS.WIDTH_PER_GROUP", 4, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16,
                            "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32, "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16),
                            "SOLVER.BASE_LR", 0.002])
    torch.manual_seed(0)
    model = build_detection_model(cfg).train()
    ds = SyntheticCOCODataset(length=2, height=96, width=128, with_masks=cfg.MODEL.MASK_ON, min_objects=2, max_objects=4)
    images, targets, _ = BatchCollator(32)([ds[0], ds[1]])
    step = TrainStep(model, make_overlapped_sgd(cfg, model), None, "float32", "cpu")
    monkeypatch.setattr(sfl.SigmoidFocalLoss, "forward",
                        lambda self, l, t: sfl.sigmoid_focal_loss_sum(l.float(), t, self.gamma, self.alpha))
    with cpu_shim.install():
        first = {k: float(v) for k, v in step(images, list(targets)).items()}
        for _ in range(3):
            last = {k: float(v) for k, v in step(images, list(targets)).items()}
        assert all(np.isfinite(v) for v in last.values())
        assert sum(last.values()) < sum(first.values())
        model.eval()
        with torch.no_grad():
            det = model(images)
        assert len(det) == 2 and det[0].has_field("scores") and det[0].has_field("labels")


def test_box_coder_reference_known_answers(golden_dir):
    """the reference's own tests/test_box_coder.py:11-105 vectors (captured by tests/golden/make_golden.py) + an encode
    fixture from the reference's BoxCoder: decode within the reference test's atol, encode exact"""
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    g = np.load(os.path.join(golden_dir, "box_coder_reference_tests.npz"))
    coder = BoxCoder(tuple(float(x) for x in g["weights"]))
    out = coder.decode(torch.from_numpy(g["rel_codes"]), torch.from_numpy(g["boxes"])).numpy()
    np.testing.assert_allclose(out, g["expected"], atol=float(g["atol"]))
    np.testing.assert_allclose(out, g["decoded"], rtol=1e-6, atol=1e-5)
    enc = BoxCoder(tuple(float(x) for x in g["enc_weights"]))
    codes = enc.encode(torch.from_numpy(g["enc_reference_boxes"]), torch.from_numpy(g["enc_proposals"])).numpy()
    np.testing.assert_allclose(codes, g["enc_codes"], rtol=1e-6, atol=1e-6)
    back = enc.decode(torch.from_numpy(codes), torch.from_numpy(g["enc_proposals"])).numpy()
    np.testing.assert_allclose(back[:, :2], g["enc_reference_boxes"][:, :2], atol=2e-3)


def test_metric_logger_reference_known_answers():
    """the reference's tests/test_metric_logger.py:9-28"""
    from maskrcnn_benchmark.utils.metric_logger import MetricLogger
    meter = MetricLogger()
    for i in range(10):
        meter.update(metric=float(i))
    m = meter.meters["metric"]
    assert m.count == 10 and m.total == 45 and m.median == 4 and m.avg == 4.5
    meter.update(loss=torch.tensor(2.0))          # device tensors are accepted and read back lazily
    assert meter.loss.global_avg == 2.0 and "loss: 2.0000 (2.0000)" in str(meter)
    assert meter.delimiter == "\t"
    with pytest.raises(AttributeError):
        meter.not_existent


def test_do_train_loop_checkpoints_and_resumes(tmp_path, caplog):
    """engine.trainer.do_train over the synthetic loader (reference engine/trainer.py:43-150): logs,
    saves model_final + last_checkpoint, and a second run resumes from the saved iteration."""
    import logging
    from maskrcnn_benchmark.data import make_data_loader
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg
    from maskrcnn_benchmark.engine.trainer import do_train
    from maskrcnn_benchmark.utils.checkpoint import DetectronCheckpointer
    opts = ["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 60, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 80,
            "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 8, "MODEL.RESNETS.WIDTH_PER_GROUP", 2,
            "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 8, "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 16, "SOLVER.BASE_LR", 0.001,
            "SOLVER.MAX_ITER", 3, "SOLVER.IMS_PER_BATCH", 1, "INPUT.MIN_SIZE_TRAIN", (64,), "INPUT.MAX_SIZE_TRAIN", 96,
            "OUTPUT_DIR", str(tmp_path)]
    cfg = load_cfg("e2e_faster_rcnn_R_50_FPN_1x.yaml", opts)
    torch.manual_seed(0)
    model, optimizer, scheduler, _ = build_training(cfg, torch.device("cpu"))
    ckpt = DetectronCheckpointer(cfg, model, optimizer, scheduler, str(tmp_path))
    args = {"iteration": 0}
    args.update(ckpt.load(None))
    loader = make_data_loader(cfg, is_train=True, length=3)
    with cpu_shim.install(), caplog.at_level(logging.INFO, logger="maskrcnn_benchmark.trainer"):
        do_train(cfg, model, loader, optimizer, scheduler, ckpt, "cpu", 2, args, log_period=1)
    assert args["iteration"] == 3
    assert (tmp_path / "model_final.pth").exists() and (tmp_path / "model_0000002.pth").exists()
    assert any("loss_objectness" in r.getMessage() and "lr:" in r.getMessage() for r in caplog.records)
    # resume
    model2, opt2, sched2, _ = build_training(cfg, torch.device("cpu"))
    ckpt2 = DetectronCheckpointer(cfg, model2, opt2, sched2, str(tmp_path))
    extra = ckpt2.load(None)
    assert extra["iteration"] == 3 and sched2.last_epoch == scheduler.last_epoch
    for a, b in zip(model.parameters(), model2.parameters()):
        assert torch.equal(a, b)


@pytest.mark.parametrize("config", ["e2e_mask_rcnn_R_50_FPN_1x.yaml", "retinanet/retinanet_R-50-FPN_1x.yaml"])
def test_tiny_model_through_emulated_hip_kernels_matches_oracle_backend(config, monkeypatch):
    """The detector's forward + backward with the detection-head operators served (a) by the oracle and
    (b) by the HIP kernel sources under the host emulation: same losses and same parameter gradients.
    Exercises the kernels with the model's own argument patterns (padded proposal sets, all FPN levels in
    one launch, segmented NMS masks, fused FrozenBN) — without a GPU."""
    import maskrcnn_benchmark.layers.sigmoid_focal_loss as sfl
    from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = load_cfg(config, ["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150,
                            "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32, "MODEL.RESNETS.RES2_OUT_CHANNELS", 16,
                            "MODEL.RESNETS.WIDTH_PER_GROUP", 4, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16,
                            "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32, "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16)])
    monkeypatch.setattr(sfl.SigmoidFocalLoss, "forward",
                        lambda self, l, t: sfl.sigmoid_focal_loss_sum(l.float(), t, self.gamma, self.alpha))
    ds = SyntheticCOCODataset(length=2, height=96, width=128, with_masks=cfg.MODEL.MASK_ON, min_objects=2, max_objects=4)
    images, targets, _ = BatchCollator(32)([ds[0], ds[1]])
    results = {}
    for backend in ("oracle", "emu"):
        torch.manual_seed(0)
        model = build_detection_model(cfg).train()
        with cpu_shim.install(backend):
            torch.manual_seed(1)  # the samplers draw random subsets
            losses = model(images, list(targets))
            sum(losses.values()).backward()
        results[backend] = ({k: float(v) for k, v in losses.items()},
                            [p.grad.clone() for p in model.parameters() if p.grad is not None])
    lo, le = results["oracle"][0], results["emu"][0]
    assert lo.keys() == le.keys()
    for k in lo:
        assert abs(lo[k] - le[k]) <= 1e-4 * max(1.0, abs(lo[k])), (k, lo[k], le[k])
    assert len(results["oracle"][1]) == len(results["emu"][1]) > 10
    for a, b in zip(results["oracle"][1], results["emu"][1]):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-4 * max(1.0, float(a.abs().max())))


# ------------------------------------------------------------------ mask head on shared box features
def test_mask_head_shared_box_features_uses_global_rows_for_every_image():
    """SHARE_BOX_FEATURE_EXTRACTOR (the C4 configs): the mask head indexes the box head's pooled features of
    ALL images; image i's slots must be offset by the proposals of images < i (reference mask_head.py:60-63)."""
    from types import SimpleNamespace
    from maskrcnn_benchmark.modeling.roi_heads.mask_head.mask_head import ROIMaskHead
    head = ROIMaskHead.__new__(ROIMaskHead)
    torch.nn.Module.__init__(head)
    head.cfg = SimpleNamespace(MODEL=SimpleNamespace(ROI_MASK_HEAD=SimpleNamespace(SHARE_BOX_FEATURE_EXTRACTOR=True)))
    head.max_positives = 3
    head.predictor = lambda x: x
    seen = {}
    head.loss_evaluator = lambda props, logits, targets: seen.setdefault("x", logits).sum() * 0
    head.train()
    props = []
    for n in (5, 4):
        b = BoxList(torch.zeros(n, 4), (10, 10))
        lab = torch.zeros(n, dtype=torch.int64)
        lab[:2] = 1
        b.add_field("labels", lab)
        b.add_field("valid", torch.ones(n, dtype=torch.bool))
        props.append(b)
    feats = torch.arange(9, dtype=torch.float32).view(9, 1)   # row id of the box head's feature tensor
    head(feats, props, None)
    assert seen["x"].flatten().tolist() == [0, 1, 2, 5, 6, 7]
    # variable-length path (no "valid" field): nonzero() indices per image
    for b in props:
        b.extra_fields.pop("valid")
    seen.clear()
    head(feats, props, None)
    assert seen["x"].flatten().tolist() == [0, 1, 5, 6]


def test_frozen_bn_folded_cache_follows_every_replacement_path():
    """the cached folded scale / bias must notice in-place writes (version counters), `.data = x` and direct
    `_buffers[...]` replacement (storage addresses); in-place writes through the `.data` alias are the documented case
    for invalidate()."""
    from maskrcnn_benchmark.layers.batch_norm import FrozenBatchNorm2d
    bn = FrozenBatchNorm2d(4)

    def expect():
        scale = bn.weight * bn.running_var.rsqrt()
        return scale, bn.bias - bn.running_mean * scale

    def check():
        s, b = bn.folded()
        es, eb = expect()
        assert torch.equal(s, es) and torch.equal(b, eb)

    check()
    bn.weight.mul_(2.0); check()                                  # in-place: version counter
    bn.running_var.data = torch.full((4,), 4.0); check()          # .data = x: storage address
    bn._buffers["bias"] = torch.full((4,), 0.5); check()          # direct replacement: storage address
    bn.running_mean = torch.full((4,), 0.25); check()             # assignment: __setattr__
    bn.load_state_dict({k: v * 3 for k, v in bn.state_dict().items()}); check()
    bn.running_mean.data.copy_(torch.full((4,), 7.0))             # invisible to any key ...
    bn.invalidate(); check()                                      # ... hence invalidate()


def test_half_weights_single_cast_equals_autocast_per_layer_casts(monkeypatch):
    """layers/half_weights.py: under autocast every weight's half copy comes from ONE multi-tensor cast through one autograd
    node (and the half gradients go back to the fp32 masters the same way) — the same losses and the same fp32 gradients as
    autocast's per-layer casts; the modules' `weight` attribute is the fp32 parameter again after the forward (also when the
    forward raises), parameters / state_dict never see the copies, the data-parallel wrapper switches the mechanism off."""
    from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml",
                   ["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150,
                    "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32, "MODEL.RESNETS.RES2_OUT_CHANNELS", 16,
                    "MODEL.RESNETS.WIDTH_PER_GROUP", 4, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16,
                    "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32, "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16)])
    ds = SyntheticCOCODataset(length=2, height=96, width=128, with_masks=True, min_objects=2, max_objects=4)
    images, targets, _ = BatchCollator(32)([ds[0], ds[1]])

    def run(enabled):
        torch.manual_seed(0)
        model = build_detection_model(cfg).train()
        model.half_weights.enabled = enabled
        keys = list(model.state_dict().keys())
        arrival = []
        for n, p in model.named_parameters():
            if n in ("roi_heads.mask.feature_extractor.mask_fcn1.weight", "backbone.body.layer2.0.conv1.weight", "rpn.head.conv.weight"):
                p.register_post_accumulate_grad_hook(lambda q, _n=n: arrival.append(_n))
        with cpu_shim.install():
            torch.manual_seed(1)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                losses = model(images, list(targets))
            total = sum(v.float() for v in losses.values())
            total.backward()
        assert list(model.state_dict().keys()) == keys
        for m in model.modules():
            assert "weight" not in m.__dict__
        model._arrival = arrival
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        return {k: float(v.detach()) for k, v in losses.items()}, grads, model

    la, ga, ma = run(False)
    lb, gb, mb = run(True)
    assert mb.half_weights.entries and len(mb.half_weights.entries) > 40 and ma.half_weights.entries is None
    # one cast node per STAGE, created right before the stage runs: a stage's fp32 gradients arrive when its backward is over
    # (heads before the RPN head before the backbone) — not all at once at the end of the backward pass in parameter order
    assert sorted(mb.half_weights.groups) == ["backbone.body.layer1", "backbone.body.layer2", "backbone.body.layer3",
                                              "backbone.body.layer4", "backbone.body.stem", "backbone.fpn", "roi_heads.box",
                                              "roi_heads.mask", "rpn.head"], sorted(mb.half_weights.groups)
    assert mb._arrival == ma._arrival == ["roi_heads.mask.feature_extractor.mask_fcn1.weight", "rpn.head.conv.weight",
                                          "backbone.body.layer2.0.conv1.weight"], (ma._arrival, mb._arrival)
    import copy
    mc = copy.deepcopy(mb)
    assert mc.half_weights is not mb.half_weights and mc.half_weights.model is mc
    assert all(h.owner is mc.half_weights for m in mc.modules() for h in m._forward_pre_hooks.values() if hasattr(h, "owner"))
    assert la.keys() == lb.keys() and ga.keys() == gb.keys()
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-6 * max(1.0, abs(la[k])), (k, la[k], lb[k])
    for n in ga:
        assert gb[n].dtype == torch.float32 and gb[n].shape == ga[n].shape
        assert torch.allclose(ga[n], gb[n], rtol=1e-5, atol=1e-7), n
    # fp32 forward: not used; a forward that raises still restores the attributes
    with pytest.raises(ValueError):
        with torch.autocast("cpu", dtype=torch.bfloat16):
            mb(images, None)
    assert all("weight" not in m.__dict__ for m in mb.modules())
    with cpu_shim.install():
        mb(images, list(targets))
    assert not mb.half_weights.installed


def test_mask_head_dynamic_slots_equal_the_fixed_quota(monkeypatch):
    """roi_heads/mask_head/mask_head.py: the mask head on the first n = ceil_g(positives) slots (granule g = 16) of every image ("dynamic": the
    reference's workload, which keeps only the positive boxes) gives the losses and gradients of the fixed quota of
    BATCH_SIZE_PER_IMAGE * POSITIVE_FRACTION slots per image — the extra slots are masked out of the loss either way —, the
    counts are requested by the box head right after its sampler, forced slot counts ("<n>") clamp to the quota."""
    from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.modeling.roi_heads.mask_head import mask_head as MH
    cfg = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml",
                   ["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150,
                    "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 256, "MODEL.RESNETS.RES2_OUT_CHANNELS", 16,
                    "MODEL.RESNETS.WIDTH_PER_GROUP", 4, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16,
                    "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32, "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16)])
    ds = SyntheticCOCODataset(length=2, height=96, width=128, with_masks=True, min_objects=2, max_objects=4)
    images, targets, _ = BatchCollator(32)([ds[0], ds[1]])

    def run(mode):
        monkeypatch.setattr(MH, "SLOT_MODE", mode)
        torch.manual_seed(0)
        model = build_detection_model(cfg).train()
        with cpu_shim.install():
            torch.manual_seed(1)
            losses = model(images, list(targets))
            sum(losses.values()).backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        return {k: float(v.detach()) for k, v in losses.items()}, grads, list(model.roi_heads.mask.last_slots)

    lf, gf, sf = run("fixed")
    ld, gd, sd = run("dynamic")
    assert sf == [64, 64]                                  # the quota: 256 x 0.25
    assert all(s % MH.SLOT_GRANULE == 0 and MH.SLOT_GRANULE <= s <= 64 for s in sd) and sd != sf, sd
    for k in lf:
        assert abs(lf[k] - ld[k]) <= 1e-6 * max(1.0, abs(lf[k])), (k, lf[k], ld[k])
    assert gf.keys() == gd.keys()
    for n in gf:
        assert torch.allclose(gf[n], gd[n], rtol=1e-4, atol=1e-7), n
    _, _, forced = run("16,500")
    assert forced == [16, 64]
