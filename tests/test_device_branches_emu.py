"""The branches the model only takes on the GPU (`_C.on_device`): fused label / sampled-slot / proposal-decode launches
and the batched hand-over of the proposals from the RPN to the box head — driven WITHOUT a GPU by serving the kernels of
csrc/targets.hip from the host emulation (cpu_shim backend "emu-device") and compared with the ATen compositions the
CPU path runs (which tests/test_model_cpu.py pins to the reference-generated fixtures)."""
import numpy as np
import pytest
import torch

import cpu_shim
from maskrcnn_benchmark import _C
from maskrcnn_benchmark.modeling.box_coder import BoxCoder
from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator
from maskrcnn_benchmark.modeling.rpn.inference import RPNPostProcessor
from maskrcnn_benchmark.modeling.rpn.loss import begin_step, make_rpn_loss_evaluator, pad_targets
from maskrcnn_benchmark.modeling.roi_heads.box_head.loss import make_roi_box_loss_evaluator, stack_proposals
from maskrcnn_benchmark.structures.bounding_box import BoxList
from maskrcnn_benchmark.structures.image_list import ImageList


def _cfg(extra=()):
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    return load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml", ["MODEL.DEVICE", "cpu"] + list(extra))


def _targets(rng, sizes, counts):
    out = []
    for (h, w), m in zip(sizes, counts):
        cx, cy = rng.uniform(0, w, m), rng.uniform(0, h, m)
        bw, bh = rng.uniform(10, w / 2, m), rng.uniform(10, h / 2, m)
        b = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).clip(0, [w - 1, h - 1, w - 1, h - 1])
        t = BoxList(torch.from_numpy(b.astype(np.float32)), (w, h), mode="xyxy")
        t.add_field("labels", torch.from_numpy(rng.randint(1, 81, m).astype(np.int64)))
        out.append(t)
    return out


def _rpn_inputs(rng, sizes, strides=(4, 8, 16, 32, 64), A=3):
    H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
    H, W = -(-H // 64) * 64, -(-W // 64) * 64
    ag = AnchorGenerator(sizes=tuple((8 * s,) for s in strides), anchor_strides=strides)
    feats = [torch.zeros(len(sizes), 1, H // s, W // s) for s in strides]
    anchors = ag(ImageList(torch.zeros(len(sizes), 3, H, W), list(sizes)), feats)
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    obj = [torch.randn(len(sizes), A, f.shape[2], f.shape[3], generator=g) for f in feats]
    reg = [torch.randn(len(sizes), 4 * A, f.shape[2], f.shape[3], generator=g) * 0.4 for f in feats]
    return anchors, obj, reg


def test_pad_targets_is_shared_within_a_step_and_rebuilt_after_begin_step_or_a_change():
    rng = np.random.RandomState(0)
    targets = _targets(rng, [(96, 128), (80, 120)], [3, 5])
    begin_step()
    a = pad_targets(targets, torch.device("cpu"))
    assert pad_targets(list(targets), torch.device("cpu"))[0] is a[0]          # same BoxLists: shared
    c = pad_targets(targets, torch.device("cpu"), ("labels",))
    assert c[0] is not a[0] and torch.equal(c[0], a[0]) and c[2]["labels"].shape == (2, 5)
    targets[0].bbox[0, 0] += 1.0                                             # an in-place edit is seen
    d = pad_targets(targets, torch.device("cpu"))
    assert d[0] is not a[0] and d[0][0, 0, 0] == a[0][0, 0, 0] + 1
    begin_step()
    assert pad_targets(targets, torch.device("cpu"))[0] is not d[0]            # a new step pads again
    assert a[1].sum() == 8 and not a[1][0, 3:].any()


@pytest.mark.parametrize("min_size", [0, 12])
def test_rpn_proposals_device_branch_equals_the_composite(min_size):
    """RPNPostProcessor.forward in training: fused decode + batched ground-truth hand-over vs the per-level ATen
    composition + per-image concatenations, compared the way the box head sees them (stack_proposals)."""
    rng = np.random.RandomState(3)
    sizes = [(160, 192), (130, 200)]
    anchors, obj, reg = _rpn_inputs(rng, sizes)
    targets = _targets(rng, sizes, [2, 6])
    post = RPNPostProcessor(300, 300, 0.7, min_size, BoxCoder((1.0, 1.0, 1.0, 1.0)), fpn_post_nms_top_n=400).train()
    with cpu_shim.install("emu"):
        ref = post(anchors, obj, reg, targets)
    begin_step()
    with cpu_shim.install("emu-device"):
        out = post(anchors, obj, reg, targets)
        boxes, valid = stack_proposals(out)
        assert boxes is out[0].batch_rows[0]["boxes"]                         # handed over as the batch, no copy
    rb, rv = stack_proposals(ref)
    assert boxes.shape == rb.shape and torch.equal(valid, rv) and int(valid.sum()) > 50
    # glibc expf vs ATen's exp in the decode: last place on the boxes (tests/test_targets_gpu.py: bit equality on the GPU)
    assert torch.allclose(boxes[valid], rb[valid], rtol=1e-6, atol=4e-3)
    for o, r in zip(out, ref):
        n = len(r)
        assert torch.equal(o.get_field("objectness")[:n], r.get_field("objectness"))
        assert not o.get_field("valid")[n:].any()                            # padded ground-truth rows are invalid
    # the appended ground truth is there, valid, with objectness 1
    K = len(ref[0]) - len(targets[0])
    assert torch.equal(out[1].bbox[K:K + 6], targets[1].bbox) and out[1].get_field("valid")[K:K + 6].all()
    assert (out[1].get_field("objectness")[K:K + 6] == 1).all()


def test_box_head_subsample_device_branch_equals_the_composite(monkeypatch):
    rng = np.random.RandomState(5)
    sizes = [(160, 192), (130, 200)]
    cfg = _cfg(["MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64])
    ev = make_roi_box_loss_evaluator(cfg)
    targets = _targets(rng, sizes, [3, 5])
    props = []
    for (h, w), t, k in zip(sizes, targets, (200, 170)):       # different lengths: the shorter list is padded
        jit = t.bbox[rng.randint(0, len(t), k // 2)] + torch.from_numpy(rng.uniform(-6, 6, (k // 2, 4)).astype(np.float32))
        rnd = _targets(rng, [(h, w)], [k - k // 2])[0].bbox
        p = BoxList(torch.cat([jit, rnd]), (w, h), mode="xyxy")
        p.add_field("objectness", torch.from_numpy(rng.rand(k).astype(np.float32)))
        p.add_field("valid", torch.from_numpy(rng.rand(k) < 0.9))
        props.append(p)
    # the same sampled slots on both sides: the emulated sampler kernel with a fixed seed
    import emu

    def sample_fixed(labels):
        _, _, idx, val = emu.sample_labels(labels.numpy(), 64, 16, seed=9)
        return torch.from_numpy(idx), torch.from_numpy(val)
    monkeypatch.setattr(ev.fg_bg_sampler, "sample_fixed", sample_fixed)
    with cpu_shim.install("emu"):
        ref = ev.subsample(props, targets)
    begin_step()
    with cpu_shim.install("emu-device"):
        out = ev.subsample(props, targets)
    assert len(out) == len(ref) == 2
    for o, r in zip(out, ref):
        assert set(o.fields()) == set(r.fields()) == {"labels", "regression_targets", "matched_idxs", "valid", "objectness"}
        assert o.size == r.size and o.mode == r.mode
        for f in ("labels", "matched_idxs", "valid", "objectness"):
            assert torch.equal(o.get_field(f), r.get_field(f)), f
        assert torch.equal(o.bbox, r.bbox)
        assert torch.allclose(o.get_field("regression_targets"), r.get_field("regression_targets"), rtol=1e-6, atol=1e-6)
        assert (o.get_field("labels") > 0).any() and (o.get_field("labels") == 0).any()


def test_rpn_labels_device_branch_equals_the_composite():
    rng = np.random.RandomState(7)
    sizes = [(160, 192), (130, 200)]
    anchors, _, _ = _rpn_inputs(rng, sizes)
    targets = _targets(rng, sizes, [4, 1])
    ev = make_rpn_loss_evaluator(_cfg(), BoxCoder((1.0, 1.0, 1.0, 1.0)))
    with cpu_shim.install("emu"):
        ref = ev._match(anchors, targets)
    begin_step()
    with cpu_shim.install("emu-device"):
        out = ev._match(anchors, targets)
    assert out[0].dtype == ref[0].dtype == torch.float32
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    assert set(np.unique(out[0].numpy())) == {-1.0, 0.0, 1.0}


def test_tiny_mask_rcnn_trains_through_the_device_branches(monkeypatch):
    """forward + backward of the detector with every `_C.on_device` branch taken: finite losses of the usual size,
    gradients on the parameters, and each fused entry point actually called"""
    from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = _cfg(["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150,
                "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32, "MODEL.RESNETS.RES2_OUT_CHANNELS", 16,
                "MODEL.RESNETS.WIDTH_PER_GROUP", 4, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16,
                "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32, "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16)])
    ds = SyntheticCOCODataset(length=2, height=96, width=128, with_masks=True, min_objects=2, max_objects=4)
    images, targets, _ = BatchCollator(32)([ds[0], ds[1]])
    torch.manual_seed(0)
    model = build_detection_model(cfg).train()
    calls = {}
    # (the counting wrappers are undone INSIDE the install block: undone after it they would re-install the shims they wrap)
    with cpu_shim.install("emu-device"), monkeypatch.context() as mp:
        for name in ("match_boxes", "sample_labels", "match_labels", "roi_head_targets", "rpn_decode", "mask_targets"):
            fn = getattr(_C, name)

            def counted(*a, _fn=fn, _name=name, **k):
                calls[_name] = calls.get(_name, 0) + 1
                return _fn(*a, **k)
            mp.setattr(_C, name, counted)
        losses = model(images, list(targets))
        sum(losses.values()).backward()
    assert calls == {"match_boxes": 2, "sample_labels": 2, "match_labels": 2, "roi_head_targets": 1, "rpn_decode": 5,
                     "mask_targets": 2}, calls
    assert set(losses) == {"loss_classifier", "loss_box_reg", "loss_mask", "loss_objectness", "loss_rpn_box_reg"}
    for k, v in losses.items():
        assert torch.isfinite(v) and 0 <= float(v.detach()) < 10, (k, float(v.detach()))
    assert sum(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters()) > 10


@pytest.mark.parametrize("backend", ["emu-device", "emu-lib"])
def test_tiny_mask_rcnn_with_the_fused_head_losses_equals_the_aten_losses(backend, monkeypatch):
    """DETOPS_HEAD_LOSS=fused (opt-in): the value + gradient kernels of csrc/head_loss.hip in the detector — same losses
    and same parameter gradients as the ATen compositions (same weights, same batch, same sampler draws).
    "emu-lib": through the product's own `_C.fastrcnn_loss` / `_C.mask_loss` autograd functions (and every other `_C`
    wrapper of the model) over the emulation library."""
    from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.modeling.roi_heads.box_head import loss as box_loss
    cfg = _cfg(["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150,
                "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32, "MODEL.RESNETS.RES2_OUT_CHANNELS", 16,
                "MODEL.RESNETS.WIDTH_PER_GROUP", 4, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16,
                "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32, "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16)])
    ds = SyntheticCOCODataset(length=2, height=96, width=128, with_masks=True, min_objects=2, max_objects=4)
    images, targets, _ = BatchCollator(32)([ds[0], ds[1]])
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(box_loss, "_FUSED_LOSS", fused)
        torch.manual_seed(0)
        model = build_detection_model(cfg).train()
        calls = []
        _C._SAMPLER_CALLS[0] = 0      # without a device generator the sampler seeds come from this per-process counter
        with cpu_shim.install(backend), monkeypatch.context() as mp:
            for name in ("fastrcnn_loss", "mask_loss"):
                mp.setattr(_C, name, (lambda *a, _fn=getattr(_C, name), _n=name, **k: (calls.append(_n), _fn(*a, **k))[1]))
            # the emulated sampler's seed counter restarts with every install(): both runs draw the same subsets
            losses = model(images, list(targets))
            sum(v * w for v, w in zip(losses.values(), (1.0, 0.7, 1.3, 0.9, 1.1))).backward()
        assert sorted(calls) == (["fastrcnn_loss", "mask_loss"] if fused else [])
        res[fused] = ({k: v.item() for k, v in losses.items()}, [p.grad.clone() for p in model.parameters() if p.grad is not None])
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) <= 1e-5 * max(1.0, abs(v)), (k, res[True][0][k], v)
    assert len(res[True][1]) == len(res[False][1]) > 10
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * max(1.0, float(b.abs().max())))


def test_stack_proposals_takes_the_handed_over_batch_only_while_the_lists_are_its_rows():
    boxes = torch.rand(2, 5, 4)
    batch = {"boxes": boxes, "valid": torch.ones(2, 5, dtype=torch.bool), "objectness": torch.rand(2, 5)}
    props = []
    for i in range(2):
        p = BoxList(boxes[i], (10, 10), mode="xyxy")
        p.add_field("valid", batch["valid"][i])
        p.batch_rows = (batch, i)
        props.append(p)
    assert stack_proposals(props)[0] is boxes
    assert stack_proposals(props[::-1])[0] is not boxes                      # not in batch order
    props[1].bbox = boxes[1].clone()                                         # boxes replaced after the hand-over
    b, v = stack_proposals(props)
    assert b is not boxes and torch.equal(b, boxes) and torch.equal(v, batch["valid"])
    props[1].bbox = boxes[1]
    props[1].extra_fields["valid"] = torch.zeros(5, dtype=torch.bool)       # validity replaced
    assert not stack_proposals(props)[1][1].any()


@pytest.mark.parametrize("seed,sizes,counts,pre,post_n", [
    (0, [(96, 128)], [3], 60, 80),                                   # one image
    (1, [(96, 128), (80, 120), (64, 64)], [2, 0, 5], 50, 90),         # an image WITHOUT ground truth in the batch
    (2, [(130, 200), (160, 192)], [9, 1], 300, 250),                  # per-batch top-n below the candidates
    (3, [(64, 96), (64, 96)], [1, 1], 1000, 2000)])                   # k = all anchors of every level
def test_proposals_and_box_head_sampling_fuzz_device_branches_vs_compositions(seed, sizes, counts, pre, post_n, monkeypatch):
    """RPNPostProcessor.forward (training, ground truth appended) + FastRCNNLossComputation.subsample over random
    batches, through the product's `_C` wrappers on the emulation library: every device branch on, against every device
    branch off (per-level ATen decode, per-image concatenation and re-stacking, ATen label chain + encode + indexing)"""
    from maskrcnn_benchmark.modeling.roi_heads.box_head import loss as box_loss
    rng = np.random.RandomState(100 + seed)
    anchors, obj, reg = _rpn_inputs(rng, sizes)
    targets = _targets(rng, sizes, counts)
    post = RPNPostProcessor(pre, pre, 0.7, 0, BoxCoder((1.0, 1.0, 1.0, 1.0)), fpn_post_nms_top_n=post_n).train()
    ev = make_roi_box_loss_evaluator(_cfg(["MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64]))
    with cpu_shim.install("emu-lib"):
        begin_step()
        props = post(anchors, obj, reg, targets)
        boxes, valid = stack_proposals(props)
        assert boxes is props[0].batch_rows[0]["boxes"]
        _C._SAMPLER_CALLS[0] = 0
        out = ev.subsample(props, targets)
        # ---- the same modules with the device branches off
        post.fused_decode = False
        monkeypatch.setattr(box_loss, "_FUSED", False)
        with monkeypatch.context() as mp:
            mp.setattr(_C, "on_device", lambda t: False)
            begin_step()
            rprops = post(anchors, obj, reg, targets)
        assert all(getattr(p, "batch_rows", None) is None for p in rprops)
        rboxes, rvalid = stack_proposals(rprops)
        _C._SAMPLER_CALLS[0] = 0
        ref = ev.subsample(rprops, targets)
    assert torch.equal(valid, rvalid) and int(valid.sum()) >= sum(counts)
    assert torch.allclose(boxes[valid], rboxes[valid], rtol=1e-6, atol=4e-3)      # glibc expf vs ATen's exp: last place
    for o, r, m in zip(out, ref, counts):
        assert set(o.fields()) == set(r.fields())
        for f in ("labels", "matched_idxs", "valid", "objectness"):
            assert torch.equal(o.get_field(f), r.get_field(f)), f
        v = o.get_field("valid")
        assert torch.allclose(o.bbox[v], r.bbox[v], rtol=1e-6, atol=4e-3)
        pos = o.get_field("labels") > 0
        assert torch.allclose(o.get_field("regression_targets")[pos], r.get_field("regression_targets")[pos], rtol=1e-4, atol=1e-4)
        assert int(pos.sum()) >= min(m, 16) and (m > 0 or not pos.any())
        assert torch.isfinite(o.get_field("regression_targets")).all()
