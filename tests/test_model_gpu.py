"""GPU tests of the model-surface pieces that call the HIP operators: dense-mask segmented NMS vs
the oracle (bit-exact), fused FPN pooler vs per-level oracle ROIAlign, and a finite training step of
each BASELINE architecture on a small image."""
import numpy as np
import pytest
import torch

import oracle
import synth

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def test_nms_batched_mask_matches_oracle():
    from maskrcnn_benchmark import _C
    segs = [synth.nms_boxes(n, seed=7 + i) for i, n in enumerate((819, 2000, 1, 64, 1337, 2000))]
    boxes = np.concatenate([b for b, _ in segs]).astype(np.float32)
    scores = np.concatenate([s for _, s in segs]).astype(np.float32)
    offs = np.cumsum([0] + [len(b) for b, _ in segs]).astype(np.int32)
    mask, num = _C.nms_batched_mask(torch.from_numpy(boxes).to(_dev()), torch.from_numpy(scores).to(_dev()),
                                    torch.from_numpy(offs).to(_dev()), 2000, 0.7)
    mask, num = mask.cpu().numpy(), num.cpu().numpy()
    for i, (b, s) in enumerate(segs):
        keep = oracle.nms(b, s, 0.7)
        want = np.zeros(len(b), bool)
        want[keep] = True
        assert np.array_equal(mask[offs[i]:offs[i + 1]], want), "segment %d" % i
        assert num[i] == len(keep)


def test_pooler_matches_per_level_oracle():
    from maskrcnn_benchmark.modeling.poolers import Pooler
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    rng = np.random.RandomState(0)
    feats = [rng.randn(2, 16, h, w).astype(np.float32) for (h, w) in synth.fpn_shapes()[:5]]
    rois = synth.fpn_rois(seed=5, per_image=40, n_images=2)
    boxes = [BoxList(torch.from_numpy(rois[rois[:, 0] == b][:, 1:]).to(_dev()), (synth.IMG_W, synth.IMG_H)) for b in (0, 1)]
    pooler = Pooler((7, 7), (0.25, 0.125, 0.0625, 0.03125), 2)
    x = [torch.from_numpy(f).to(_dev()).requires_grad_(True) for f in feats]
    out = pooler(x, boxes)
    lv = synth.level_map(rois)
    ref = np.zeros(out.shape, np.float32)
    for l in range(4):
        sel = np.nonzero(lv == l)[0]
        if sel.size:
            ref[sel] = oracle.roi_align_forward(feats[l], rois[sel], 1.0 / (4 << l), 7, 7, 2)
    assert np.abs(out.detach().cpu().numpy() - ref).max() <= 1e-4
    g = rng.randn(*out.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(_dev()))
    for l in range(4):
        sel = np.nonzero(lv == l)[0]
        want = oracle.roi_align_backward(g[sel], rois[sel], 1.0 / (4 << l), 7, 7, *feats[l].shape, 2, acc64=True) \
            if sel.size else np.zeros_like(feats[l])
        got = x[l].grad.cpu().numpy()
        assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    assert x[4].grad is None  # P6 is not pooled from


@pytest.mark.parametrize("config", ["e2e_mask_rcnn_R_50_FPN_1x.yaml", "e2e_faster_rcnn_R_50_FPN_1x.yaml",
                                    "retinanet/retinanet_R-50-FPN_1x.yaml"])
def test_train_step_finite(config):
    from maskrcnn_benchmark.engine.bench_step import smoke_train_step
    vals = smoke_train_step(_dev(), config, steps=3)
    assert all(np.isfinite(v) for v in vals.values())
    assert len(vals) >= 2


def test_train_step_dcn_bf16():
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches
    cfg = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml",
                   ["MODEL.RESNETS.STAGE_WITH_DCN", "(False, True, True, True)", "DTYPE", "bfloat16",
                    "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 300, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64])
    torch.manual_seed(0)
    model, opt, sched, step = build_training(cfg, _dev())
    (images, targets), = make_device_batches(cfg, _dev(), images_per_gpu=1, num_batches=1, height=192, width=256)
    for _ in range(2):
        losses = step(images, targets)
    assert all(torch.isfinite(v) for v in losses.values())
