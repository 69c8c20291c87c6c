"""GPU parity of csrc/targets.hip (fused IoU + Matcher, sampler, mask targets) at the BASELINE sizes: 268,569
RPN anchors / 201,600 RetinaNet anchors per image, 512-ROI box-head sampling, 28 x 28 mask targets — against the
torch CPU implementations that tests/test_model_cpu.py pins to the reference-generated fixtures."""
import numpy as np
import pytest
import torch

import synth
from maskrcnn_benchmark.modeling.matcher import Matcher
from maskrcnn_benchmark.structures.boxlist_ops import box_iou_matrix

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gt(rng, N, M, W=1344, H=800):
    cx, cy = rng.uniform(0, W, (N, M)), rng.uniform(0, H, (N, M))
    w, h = rng.uniform(8, 500, (N, M)), rng.uniform(8, 400, (N, M))
    return np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1).astype(np.float32)


@pytest.mark.parametrize("name,hi,lo,lq,strides,sizes,per", [
    ("rpn", 0.7, 0.3, True, (4, 8, 16, 32, 64), ((32,), (64,), (128,), (256,), (512,)), 3),
    ("retinanet", 0.5, 0.4, True, (8, 16, 32, 64, 128), None, 9)])
def test_match_boxes_full_anchor_set_equals_torch_matcher(name, hi, lo, lq, strides, sizes, per):
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator
    from maskrcnn_benchmark.structures.image_list import ImageList
    if sizes is None:
        sizes = tuple(tuple(s * 2 ** (k / 3.0) for k in range(3)) for s in (32, 64, 128, 256, 512))
    ag = AnchorGenerator(sizes=sizes, anchor_strides=strides, straddle_thresh=-1)
    feats = [torch.zeros(1, 1, -(-800 // s), -(-1344 // s)) for s in strides]
    anchors = torch.cat([b.bbox for b in ag(ImageList(torch.zeros(1, 3, 800, 1344), [(800, 1344)]), feats)[0]])
    K = anchors.shape[0]
    assert K == {"rpn": 268569, "retinanet": 201600}[name]
    rng = np.random.RandomState(5)
    N, M = 2, 23
    gt = _gt(rng, N, M)
    valid = np.ones((N, M), bool)
    valid[1, 9:] = False
    gt[~valid] = [-1e5, -1e5, -1e5 + 1, -1e5 + 1]
    gt[0, 4] = anchors[12345].numpy()       # a ground truth identical to an anchor (IoU exactly 1)
    out = _C.match_boxes(torch.from_numpy(gt).to(DEV), torch.from_numpy(valid).to(DEV), anchors.to(DEV), hi, lo, lq).cpu()
    iou = box_iou_matrix(torch.from_numpy(gt), anchors.unsqueeze(0).expand(N, -1, -1))
    iou = torch.where(torch.from_numpy(valid)[:, :, None], iou, iou.new_full((), -1.0))
    ref = Matcher(hi, lo, allow_low_quality_matches=lq)(iou, torch.from_numpy(valid))
    assert torch.equal(out, ref)
    assert int((out >= 0).sum()) > 50 and int((out == -2).sum()) > 50


def test_match_boxes_box_head_batched_proposals():
    from maskrcnn_benchmark import _C
    rng = np.random.RandomState(6)
    N, M, K = 2, 40, 2040
    gt = _gt(rng, N, M)
    valid = np.ones((N, M), bool)
    boxes = _gt(rng, N, K)
    boxes[:, :M] = gt                        # ground truth appended to the proposals (add_gt_proposals)
    out = _C.match_boxes(torch.from_numpy(gt).to(DEV), torch.from_numpy(valid).to(DEV), torch.from_numpy(boxes).to(DEV),
                         0.5, 0.5, False).cpu()
    iou = box_iou_matrix(torch.from_numpy(gt), torch.from_numpy(boxes))
    ref = Matcher(0.5, 0.5, allow_low_quality_matches=False)(iou, torch.from_numpy(valid))
    assert torch.equal(out, ref)


def test_sample_labels_rpn_and_box_head_sizes():
    from maskrcnn_benchmark import _C
    g = torch.Generator().manual_seed(1)
    N, n = 2, 268569
    labels = torch.zeros(N, n)
    labels[torch.rand(N, n, generator=g) < 0.3] = -1
    labels[0, torch.randperm(n, generator=g)[:57]] = 1
    labels[1, torch.randperm(n, generator=g)[:900]] = 1
    lab = labels.to(DEV)
    pos, neg = _C.sample_labels(lab, 256, 128, seed=7)
    for i, kp in enumerate((57, 128)):
        assert int(pos[i].sum()) == kp and int(neg[i].sum()) == 256 - kp
    assert not (pos & ~(lab >= 1)).any() and not (neg & ~(lab == 0)).any()
    p2, n2 = _C.sample_labels(lab, 256, 128, seed=7)
    assert torch.equal(pos, p2) and torch.equal(neg, n2)
    p3, n3 = _C.sample_labels(lab, 256, 128)            # default seed: advances per call
    p4, n4 = _C.sample_labels(lab, 256, 128)
    assert not torch.equal(n3, n4)
    # each of the 900 positives of row 1 is drawn ~128/900 of the time
    hits = torch.zeros(n, device=DEV)
    for s in range(300):
        hits += _C.sample_labels(lab[1:], 256, 128, seed=100 + s)[0][0]
    h = hits[lab[1] >= 1]
    assert abs(float(h.mean()) - 300 * 128 / 900) < 1e-3 and float(h.min()) >= 12 and float(h.max()) <= 80
    # box head: int64 class labels, fixed-length list, positives first
    cls = torch.zeros(2, 2040, dtype=torch.int64)
    cls[0, :300] = torch.randint(1, 81, (300,), generator=g)
    cls[1, :20] = 5
    cls[1, 1500:] = -1
    pos, neg, idx, val = _C.sample_labels(cls.to(DEV), 512, 128, with_list=True, seed=3)
    idx, val = idx.cpu(), val.cpu()
    for i, kp in enumerate((128, 20)):
        assert int(pos[i].sum()) == kp and int(neg[i].sum()) == 512 - kp and bool(val[i].all())
        assert (cls[i][idx[i][:kp]] >= 1).all() and (cls[i][idx[i][kp:]] == 0).all()
        assert idx[i].unique().numel() == 512


def test_mask_targets_bit_equal_to_the_torch_cpu_path_at_model_size():
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.modeling.roi_heads.mask_head.loss import project_masks_on_boxes
    rng = np.random.RandomState(8)
    G, H, W, P, M = 12, 800, 1344, 256, 28
    yy, xx = np.mgrid[:H, :W]
    masks = np.stack([((yy - rng.uniform(100, 700)) ** 2 / rng.uniform(900, 40000) +
                       (xx - rng.uniform(100, 1200)) ** 2 / rng.uniform(900, 90000)) < 1 for _ in range(G)]).astype(np.uint8)
    rois = synth.fpn_rois(seed=11, per_image=P, n_images=1)[:, 1:]
    which = rng.randint(0, G, P).astype(np.int64)
    # uint8 / float / bool, and other integer dtypes (int64, int16): the CPU composite's `.type_as(masks)` truncates for
    # every non-floating mask, so they must yield 0 / 1 targets too
    for m in (torch.from_numpy(masks), torch.from_numpy(masks).float(), torch.from_numpy(masks).bool(),
              torch.from_numpy(masks).long(), torch.from_numpy(masks).short()):
        ref = project_masks_on_boxes(m, torch.from_numpy(which), torch.from_numpy(rois), M)
        out = _C.mask_targets(m.to(DEV), torch.from_numpy(which).to(DEV), torch.from_numpy(rois).to(DEV), M).cpu()
        assert torch.equal(out, ref), m.dtype


def test_rpn_loss_fused_full_size_equals_torch_composite_and_autograd():
    """BASELINE shape: 2 images x 268,569 anchors over 5 levels.  Losses within 1e-5 rel, gradients within 1e-5 of
    what autograd derives from the reference-order composite; deterministic run to run; weights / upstream honoured."""
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.modeling.rpn.loss import smooth_l1_elementwise
    from maskrcnn_benchmark.modeling.rpn.utils import concat_box_prediction_layers

    rng = np.random.RandomState(3)
    N, A, M = 2, 3, 14
    shapes = synth.fpn_shapes()[:5]
    T = A * sum(h * w for h, w in shapes)
    assert T == 268569
    t = lambda a: torch.from_numpy(a).to(DEV)
    obj = [t(rng.randn(N, A, h, w).astype(np.float32) * 3).requires_grad_() for h, w in shapes]
    box = [t(rng.randn(N, 4 * A, h, w).astype(np.float32) * 0.3).requires_grad_() for h, w in shapes]
    x1 = rng.uniform(0, 1300, T); y1 = rng.uniform(0, 760, T)
    anchors = t(np.stack([x1, y1, x1 + rng.uniform(8, 500, T), y1 + rng.uniform(8, 500, T)], 1).astype(np.float32))
    gt = t(_gt(rng, N, M))
    matched = rng.randint(-2, M, (N, T)).astype(np.int64)
    pos = t((matched >= 0) & (rng.rand(N, T) < 0.002))
    neg = t((matched == -1) & (rng.rand(N, T) < 0.003))
    matched = t(matched)
    for weights in ((1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)):
        beta = 1.0 / 9
        o, b = concat_box_prediction_layers(obj, box, keep_batch=True)
        o = o.squeeze(-1)
        mg = torch.gather(gt, 1, matched.clamp(min=0)[:, :, None].expand(-1, -1, 4))
        tg = BoxCoder(weights).encode(mg, anchors.unsqueeze(0))
        ns = (pos | neg).sum().clamp(min=1).float()
        bl = smooth_l1_elementwise(b, tg, beta).sum(-1)
        box_loss = torch.where(pos, bl, torch.zeros_like(bl)).sum() / ns
        bce = torch.nn.functional.binary_cross_entropy_with_logits(o, pos.float(), reduction="none")
        obj_loss = torch.where(pos | neg, bce, torch.zeros_like(bce)).sum() / ns
        (0.7 * obj_loss + 1.3 * box_loss).backward()
        ref = [p.grad.clone() for p in obj + box]
        for p in obj + box:
            p.grad = None
        lo, lb = _C.rpn_loss(obj, box, anchors, matched, pos, neg, gt, beta, weights)
        (0.7 * lo + 1.3 * lb).backward()
        got = [p.grad.clone() for p in obj + box]
        for p in obj + box:
            p.grad = None
        assert abs(float(lo) - float(obj_loss)) <= 1e-5 * max(1.0, abs(float(obj_loss)))
        assert abs(float(lb) - float(box_loss)) <= 1e-5 * max(1.0, abs(float(box_loss)))
        for a, r in zip(got, ref):
            torch.testing.assert_close(a, r, rtol=1e-5, atol=1e-8)
        lo2, lb2 = _C.rpn_loss(obj, box, anchors, matched, pos, neg, gt, beta, weights)
        assert float(lo2) == float(lo) and float(lb2) == float(lb)   # fixed-order sums


@pytest.mark.parametrize("training,min_size", [(True, 0), (False, 16)])
def test_rpn_proposal_selection_with_the_fused_decode_is_bit_equal_to_the_aten_composition(training, min_size):
    """RPNPostProcessor._select at the BASELINE shape (5 FPN levels of an 800 x 1344 batch, 2000 / 1000 pre-NMS
    candidates per level): the one-launch-per-level decode (_C.rpn_decode) against the per-level ATen composition
    (gather, BoxCoder.decode, clip, min-size mask, concatenations) — boxes, scores and the post-NMS validity mask
    all identical (tests/test_emu_targets.py holds the same kernel against the CPU composite)."""
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator
    from maskrcnn_benchmark.modeling.rpn.inference import RPNPostProcessor
    strides = (4, 8, 16, 32, 64)
    ag = AnchorGenerator(sizes=((32,), (64,), (128,), (256,), (512,)), anchor_strides=strides)
    shapes = [(-(-800 // s), -(-1344 // s)) for s in strides]
    anchors = [a.to(DEV) for a in ag.grid_anchors(shapes)]
    g = torch.Generator().manual_seed(3)
    N, A = 2, 3
    obj = [torch.randn(N, A, h, w, generator=g).to(DEV) for h, w in shapes]
    reg = [(torch.randn(N, 4 * A, h, w, generator=g) * 0.5).to(DEV) for h, w in shapes]
    reg[2][0, 2::4] += 5.0
    sizes = [(800, 1344), (771, 1203)]
    top = 2000 if training else 1000
    post = RPNPostProcessor(top, top, 0.7, min_size, BoxCoder((1.0, 1.0, 1.0, 1.0)), fpn_post_nms_top_n=top)
    out = {}
    for fused in (True, False):
        post.fused_decode = fused
        out[fused] = post._select(anchors, obj, reg, sizes, training)
    for a, b in zip(out[True], out[False]):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)
    boxes, scores, valid = out[True]
    assert 0 < int(valid.sum()) <= (top if training else N * top)


def test_match_labels_and_sampled_slot_targets_equal_the_aten_composition_on_the_device():
    """_C.match_labels / _C.roi_head_targets at the box-head size (2 x 2020 proposals, 512 slots) against the ATen
    composition on the same device: labels, boxes, matched indices and objectness identical, encoded targets to the
    last place (division / log in one kernel vs separate ATen kernels)."""
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    rng = np.random.RandomState(31)
    N, K, M, B = 2, 2020, 20, 512
    weights = (10.0, 10.0, 5.0, 5.0)
    gt = torch.from_numpy(_gt(rng, N, M)).to(DEV)
    boxes = torch.from_numpy(_gt(rng, N, K)).to(DEV)
    boxes[:, :M * 20] = gt.repeat(1, 20, 1) + torch.from_numpy(rng.uniform(-8, 8, (N, M * 20, 4)).astype(np.float32)).to(DEV)
    boxes[..., 2:] = torch.maximum(boxes[..., 2:], boxes[..., :2] + 1)
    gl = torch.from_numpy(rng.randint(1, 81, (N, M)).astype(np.int64)).to(DEV)
    gvalid = torch.ones((N, M), dtype=torch.bool, device=DEV)
    valid = torch.from_numpy(rng.rand(N, K) < 0.9).to(DEV)
    obj = torch.from_numpy(rng.rand(N, K).astype(np.float32)).to(DEV)
    matched = _C.match_boxes(gt, gvalid, boxes, 0.5, 0.5, False)
    matched[0, 5:9] = Matcher.BETWEEN_THRESHOLDS
    ref = torch.gather(gl, 1, matched.clamp(min=0))
    ref = torch.where(matched == Matcher.BELOW_LOW_THRESHOLD, torch.zeros_like(ref), ref)
    ref = torch.where(matched == Matcher.BETWEEN_THRESHOLDS, torch.full_like(ref, -1), ref)
    ref = torch.where(valid, ref, torch.full_like(ref, -1))
    labels = _C.match_labels(matched, gl, valid, torch.int64)
    assert torch.equal(labels, ref) and int((labels > 0).sum()) > 100
    rpn = torch.where(valid & (matched != Matcher.BETWEEN_THRESHOLDS), (matched >= 0).float(), torch.full((), -1.0, device=DEV))
    assert torch.equal(_C.match_labels(matched, None, valid, torch.float32), rpn)
    _, _, idx, slot_valid = _C.sample_labels(labels, B, 128, with_list=True, seed=4)
    ob, ol, oreg, om, oo = _C.roi_head_targets(boxes, matched, gt, gl, valid, idx, slot_valid, obj, weights)
    reg = BoxCoder(weights).encode(torch.gather(gt, 1, matched.clamp(min=0)[:, :, None].expand(-1, -1, 4)), boxes)
    for i in range(N):
        sel = idx[i]
        assert torch.equal(ob[i], boxes[i][sel]) and torch.equal(om[i], matched[i][sel]) and torch.equal(oo[i], obj[i][sel])
        assert torch.equal(ol[i], torch.where(slot_valid[i], labels[i][sel], torch.full_like(sel, -1)))
        assert torch.allclose(oreg[i], reg[i][sel], rtol=2e-6, atol=2e-6)
    assert int((ol > 0).sum()) > 50 and int((ol == 0).sum()) > 50
    assert _C.roi_head_targets(boxes, matched, gt, gl, None, idx, slot_valid, None, weights)[4] is None


def test_rpn_to_box_head_hand_over_on_the_device_equals_the_per_image_composition(monkeypatch):
    """training RPNPostProcessor.forward + FastRCNNLossComputation.subsample on the device (fused decode, batched
    ground-truth hand-over, fused labels / sampled-slot targets) against the same modules with every device branch
    switched off (ATen compositions on the same tensors), with the device generator re-seeded so both draw the same
    sampled subsets."""
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.modeling.roi_heads.box_head import loss as box_loss
    from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator
    from maskrcnn_benchmark.modeling.rpn.inference import RPNPostProcessor
    from maskrcnn_benchmark.modeling.rpn.loss import begin_step
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.image_list import ImageList
    strides = (4, 8, 16, 32, 64)
    sizes = [(800, 1344), (771, 1203)]
    ag = AnchorGenerator(sizes=((32,), (64,), (128,), (256,), (512,)), anchor_strides=strides).to(DEV)
    feats = [torch.zeros(2, 1, -(-800 // s), -(-1344 // s), device=DEV) for s in strides]
    anchors = ag(ImageList(torch.zeros(2, 3, 800, 1344, device=DEV), sizes), feats)
    g = torch.Generator().manual_seed(8)
    obj = [torch.randn(2, 3, f.shape[2], f.shape[3], generator=g).to(DEV) for f in feats]
    reg = [(torch.randn(2, 12, f.shape[2], f.shape[3], generator=g) * 0.3).to(DEV) for f in feats]
    rng = np.random.RandomState(2)
    targets = []
    for (h, w), m in zip(sizes, (7, 12)):
        t = BoxList(torch.from_numpy(_gt(rng, 1, m, w, h)[0]).clamp(min=0).to(DEV), (w, h), mode="xyxy")
        t.add_field("labels", torch.from_numpy(rng.randint(1, 81, m).astype(np.int64)).to(DEV))
        targets.append(t)
    post = RPNPostProcessor(2000, 2000, 0.7, 0, BoxCoder((1.0, 1.0, 1.0, 1.0)), fpn_post_nms_top_n=2000).train()
    ev = box_loss.make_roi_box_loss_evaluator(load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml"))

    def run():
        begin_step()
        props = post(anchors, obj, reg, targets)
        boxes, valid = box_loss.stack_proposals(props)
        torch.cuda.manual_seed(77)
        return props, boxes, valid, ev.subsample(props, targets)
    props, boxes, valid, out = run()
    assert boxes is props[0].batch_rows[0]["boxes"] and boxes.shape == (2, 4 * 2000 + 819 + 12, 4)
    post.fused_decode = False
    monkeypatch.setattr(box_loss, "_FUSED", False)
    on = _C.on_device
    monkeypatch.setattr(_C, "on_device", lambda t: False)
    rprops = post(anchors, obj, reg, targets)
    monkeypatch.setattr(_C, "on_device", on)                    # matcher + sampler kernels again; the rest stays composite
    assert all(getattr(p, "batch_rows", None) is None for p in rprops)
    rboxes, rvalid = box_loss.stack_proposals(rprops)
    torch.cuda.manual_seed(77)
    ref = ev.subsample(rprops, targets)
    assert torch.equal(valid, rvalid) and 19 < int(valid.sum()) <= 2000 + 19
    assert torch.equal(boxes[valid], rboxes[valid])
    for o, r in zip(out, ref):
        assert set(o.fields()) == set(r.fields())
        for f in ("labels", "matched_idxs", "valid", "objectness"):
            assert torch.equal(o.get_field(f), r.get_field(f)), f
        assert torch.equal(o.bbox, r.bbox)
        assert torch.allclose(o.get_field("regression_targets"), r.get_field("regression_targets"), rtol=2e-6, atol=2e-6)
        assert int((o.get_field("labels") > 0).sum()) >= 7


def test_training_proposals_are_identical_with_an_injected_nms_timeout():
    """VERDICT r04 "missing" #4: the proposal selection discards num_keep — so the library itself must never drop a
    segment.  Same RPN outputs through RPNPostProcessor twice: normally, and with every wait of the single-launch NMS
    timing out (fault injection); the repair launch redoes the 10 segments -> identical proposals, and the status word
    the trainer reads at its logging interval says 10."""
    from maskrcnn_benchmark import _C, _lib
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.modeling.roi_heads.box_head import loss as box_loss
    from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator
    from maskrcnn_benchmark.modeling.rpn.inference import RPNPostProcessor
    from maskrcnn_benchmark.modeling.rpn.loss import begin_step
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.image_list import ImageList
    strides = (4, 8, 16, 32, 64)
    sizes = [(800, 1344), (771, 1203)]
    ag = AnchorGenerator(sizes=((32,), (64,), (128,), (256,), (512,)), anchor_strides=strides).to(DEV)
    feats = [torch.zeros(2, 1, -(-800 // s), -(-1344 // s), device=DEV) for s in strides]
    anchors = ag(ImageList(torch.zeros(2, 3, 800, 1344, device=DEV), sizes), feats)
    g = torch.Generator().manual_seed(9)
    obj = [torch.randn(2, 3, f.shape[2], f.shape[3], generator=g).to(DEV) for f in feats]
    reg = [(torch.randn(2, 12, f.shape[2], f.shape[3], generator=g) * 0.3).to(DEV) for f in feats]
    rng = np.random.RandomState(3)
    targets = []
    for (h, w), m in zip(sizes, (5, 9)):
        t = BoxList(torch.from_numpy(_gt(rng, 1, m, w, h)[0]).clamp(min=0).to(DEV), (w, h), mode="xyxy")
        t.add_field("labels", torch.from_numpy(rng.randint(1, 81, m).astype(np.int64)).to(DEV))
        targets.append(t)
    post = RPNPostProcessor(2000, 2000, 0.7, 0, BoxCoder((1.0, 1.0, 1.0, 1.0)), fpn_post_nms_top_n=2000).train()

    def run():
        begin_step()
        boxes, valid = box_loss.stack_proposals(post(anchors, obj, reg, targets))
        return boxes.clone(), valid.clone()
    boxes, valid = run()
    assert 100 < int(valid.sum())
    _C.nms_repaired_segments(reset=True)
    try:
        _lib.tuning_set("nms_fault", 1)
        _lib.tuning_set("nms_spin_budget", 200)
        fboxes, fvalid = run()
    finally:
        _lib.tuning_set("nms_fault", 0)
        _lib.tuning_set("nms_spin_budget", 0)
    assert torch.equal(valid, fvalid) and torch.equal(boxes[valid], fboxes[fvalid])
    assert _C.nms_repaired_segments(DEV, reset=True) == 10


def test_fused_head_losses_equal_the_aten_compositions_on_the_device():
    """_C.fastrcnn_loss / _C.mask_loss (csrc/head_loss.hip, the default since round 5) at the model's sizes — 1024 sampled ROIs x 81 classes,
    256 mask ROIs x 81 x 28 x 28 — against the ATen compositions and their autograd on the same device"""
    from torch.nn import functional as F
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.modeling.rpn.loss import smooth_l1_elementwise
    g = torch.Generator().manual_seed(5)
    R, C = 1024, 81
    logits = (torch.randn(R, C, generator=g) * 3).to(DEV)
    box = (torch.randn(R, 4 * C, generator=g) * 0.8).to(DEV)
    tgt = (torch.randn(R, 4, generator=g) * 0.8).to(DEV)
    labels = torch.randint(-1, C, (R,), generator=g).to(DEV)
    labels[:600] = 0
    tl, tb = logits.clone().requires_grad_(), box.clone().requires_grad_()
    n = (labels >= 0).sum().clamp(min=1).float()
    cls = F.cross_entropy(tl, labels, ignore_index=-1, reduction="sum") / n
    cols = 4 * labels.clamp(min=0)[:, None] + torch.arange(4, device=DEV)
    l1 = smooth_l1_elementwise(torch.gather(tb, 1, cols), tgt, beta=1.0).sum(dim=1)
    reg = torch.where(labels > 0, l1, torch.zeros_like(l1)).sum() / n
    (1.7 * cls + 0.6 * reg).backward()
    fl, fb = logits.clone().requires_grad_(), box.clone().requires_grad_()
    lc, lb = _C.fastrcnn_loss(fl, fb, labels, tgt, False, 1.0)
    (1.7 * lc + 0.6 * lb).backward()
    assert torch.allclose(lc, cls, rtol=1e-5) and torch.allclose(lb, reg, rtol=1e-5)
    assert torch.allclose(fl.grad, tl.grad, rtol=1e-4, atol=1e-8) and torch.allclose(fb.grad, tb.grad, rtol=1e-4, atol=1e-8)
    P, M = 256, 28
    ml = (torch.randn(P, C, M, M, generator=g) * 2).to(DEV)
    mt = (torch.rand(P, M, M, generator=g) < 0.4).float().to(DEV)
    lab = torch.randint(-1, C, (P,), generator=g).to(DEV)
    a = ml.clone().requires_grad_()
    pos = lab > 0
    plane = a.gather(1, lab.clamp(min=0)[:, None, None, None].expand(-1, 1, M, M)).squeeze(1)
    bce = F.binary_cross_entropy_with_logits(plane, mt, reduction="none")
    ref = torch.where(pos[:, None, None], bce, torch.zeros_like(bce)).sum() / (pos.sum() * (M * M)).clamp(min=1).float()
    (0.8 * ref).backward()
    b = ml.clone().requires_grad_()
    out = _C.mask_loss(b, lab, mt)
    (0.8 * out).backward()
    assert torch.allclose(out, ref, rtol=1e-5) and torch.allclose(b.grad, a.grad, rtol=1e-4, atol=1e-10)
    assert float(_C.mask_loss(ml, torch.zeros_like(lab), mt)) == 0.0
