"""Generate tests/golden/model_*.npz from the REFERENCE's own Python modules (build container only).

The reference package cannot be imported wholesale here (apex / yacs / torchvision / pycocotools /
cv2 are absent), but the modules on the hot path's caller side only need `torch`: they are imported
from /root/reference with inert stubs for `maskrcnn_benchmark.layers` (its `nms` is the reference's
own compiled CPU kernel from oracle/_ref, `smooth_l1_loss` is loaded from the reference file), for
`cv2` / `pycocotools` (never called: binary masks only) and with `np.float = float` (removed from
numpy >= 1.24, used at modeling/rpn/anchor_generator.py:229-238).

Captured (inputs + the reference's outputs), all seeded:
  model_anchors.npz      generate_anchors / grid_anchors / visibility   (rpn/anchor_generator.py)
  model_matcher.npz      Matcher on random IoU matrices with ties        (modeling/matcher.py)
  model_boxlist.npz      BoxList convert/resize/transpose/crop/clip/area, boxlist_iou, LevelMapper
  model_targets.npz      RPNLossComputation.prepare_targets (RPN + RetinaNet flavours),
                         FastRCNNLossComputation.prepare_targets, RPN loss value with a fixed sampler
  model_masks.npz        project_masks_on_boxes on BinaryMaskList targets (mask_head/loss.py:11-42)
  model_proposals.npz    RPNPostProcessor.forward (train + test settings) on random head outputs
  model_solver.npz       WarmupMultiStepLR factors, smooth_l1_loss values
  model_postprocess.npz  evaluation post-processing: PostProcessor.forward (roi_heads/box_head/inference.py:
                         softmax, decode, clip, score threshold, per-class NMS, detections-per-image kthvalue)
                         and RetinaNetPostProcessor.forward (rpn/retinanet/inference.py: per-level threshold +
                         top-n, decode, per-class NMS over levels, detections-per-image)

Run:  python tests/golden/make_golden_model.py      (this script never runs on the GPU box)
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

np.float = float  # noqa: shim for the reference's anchor generator

sys.path.insert(0, ROOT)
import oracle  # noqa: E402

ref_C = oracle.ref()
assert ref_C is not None, "build oracle/_ref first (python -c 'from oracle import build_ref; build_ref.build()')"
sys.path.remove(ROOT)
for m in [k for k in sys.modules if k.startswith("oracle")]:
    pass  # oracle stays importable through sys.modules; the product package is never imported here


def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_reference():
    sys.path.insert(0, REF)
    for name in ("cv2", "pycocotools", "pycocotools.mask", "apex", "apex.amp"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    layers = types.ModuleType("maskrcnn_benchmark.layers")
    layers.__path__ = []
    sl1 = _load_file("maskrcnn_benchmark.layers.smooth_l1_loss", os.path.join(REF, "maskrcnn_benchmark/layers/smooth_l1_loss.py"))
    layers.smooth_l1_loss = sl1.smooth_l1_loss
    layers.nms = lambda dets, scores, thr: ref_C.nms(dets, scores, thr)
    layers.interpolate = torch.nn.functional.interpolate
    for n in ("ROIAlign", "ROIPool", "SigmoidFocalLoss", "Conv2d", "ConvTranspose2d", "FrozenBatchNorm2d"):
        setattr(layers, n, type(n, (torch.nn.Module,), {}))
    misc = types.ModuleType("maskrcnn_benchmark.layers.misc")
    misc.interpolate = torch.nn.functional.interpolate
    sys.modules["maskrcnn_benchmark.layers.misc"] = misc
    import maskrcnn_benchmark  # the reference's (empty) package __init__
    assert maskrcnn_benchmark.__file__.startswith(REF)
    sys.modules["maskrcnn_benchmark.layers"] = layers
    maskrcnn_benchmark.layers = layers


install_reference()
from maskrcnn_benchmark.modeling.balanced_positive_negative_sampler import BalancedPositiveNegativeSampler  # noqa: E402,F401
from maskrcnn_benchmark.modeling.box_coder import BoxCoder  # noqa: E402
from maskrcnn_benchmark.modeling.matcher import Matcher  # noqa: E402
from maskrcnn_benchmark.modeling.poolers import LevelMapper  # noqa: E402
from maskrcnn_benchmark.modeling.roi_heads.box_head.loss import FastRCNNLossComputation  # noqa: E402
from maskrcnn_benchmark.modeling.roi_heads.mask_head.loss import project_masks_on_boxes  # noqa: E402
from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator, generate_anchors  # noqa: E402
from maskrcnn_benchmark.modeling.rpn.inference import RPNPostProcessor  # noqa: E402
from maskrcnn_benchmark.modeling.rpn.loss import RPNLossComputation, generate_rpn_labels  # noqa: E402
from maskrcnn_benchmark.modeling.rpn.retinanet.loss import generate_retinanet_labels  # noqa: E402
from maskrcnn_benchmark.solver.lr_scheduler import WarmupMultiStepLR  # noqa: E402
from maskrcnn_benchmark.structures.bounding_box import BoxList  # noqa: E402
from maskrcnn_benchmark.structures.boxlist_ops import boxlist_iou  # noqa: E402
from maskrcnn_benchmark.structures.image_list import ImageList  # noqa: E402
from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask  # noqa: E402


def rand_boxes(rng, n, W, H, smin=8, smax=None):
    smax = smax or min(W, H)
    w = rng.uniform(smin, smax, n)
    h = rng.uniform(smin, smax, n)
    x1 = rng.uniform(0, W - 2, n)
    y1 = rng.uniform(0, H - 2, n)
    return np.stack([x1, y1, np.minimum(x1 + w, W - 1), np.minimum(y1 + h, H - 1)], 1).astype(np.float32)


def save(name, **arrays):
    np.savez(os.path.join(HERE, name), **arrays)
    print(name, {k: np.asarray(v).shape for k, v in arrays.items()})


def t2n(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ anchors
def gen_anchors():
    out = {}
    for i, (stride, size) in enumerate(zip((4, 8, 16, 32, 64), (32, 64, 128, 256, 512))):
        out["rpn_cell_%d" % i] = t2n(generate_anchors(stride, (size,), (0.5, 1.0, 2.0)).float())
    for i, (stride, size) in enumerate(zip((8, 16, 32, 64, 128), (32, 64, 128, 256, 512))):
        sizes = tuple(size * 2 ** (k / 3.0) for k in range(3))
        out["retina_cell_%d" % i] = t2n(generate_anchors(stride, sizes, (0.5, 1.0, 2.0)).float())
    ag = AnchorGenerator(sizes=((32,), (64,), (128,)), anchor_strides=(4, 8, 16), straddle_thresh=0)
    grids = [(5, 7), (3, 4), (2, 2)]
    feats = [torch.zeros(1, 1, h, w) for h, w in grids]
    il = ImageList(torch.zeros(2, 3, 20, 28), [(20, 28), (17, 23)])
    anchors = ag(il, feats)
    for l in range(3):
        out["grid_%d" % l] = t2n(anchors[0][l].bbox)
        for i in range(2):
            out["vis_%d_%d" % (i, l)] = t2n(anchors[i][l].get_field("visibility"))
    out["grids"] = np.array(grids)
    out["image_sizes"] = np.array([(20, 28), (17, 23)])
    save("model_anchors.npz", **out)


# ------------------------------------------------------------------ matcher
def gen_matcher():
    rng = np.random.RandomState(0)
    out = {}
    cfgs = [(0.7, 0.3, True), (0.5, 0.5, False), (0.5, 0.4, True)]
    for c, (hi, lo, lq) in enumerate(cfgs):
        for k in range(4):
            M, N = rng.randint(1, 9), rng.randint(5, 60)
            q = np.round(rng.uniform(0, 1, (M, N)), 1 if k % 2 else 3).astype(np.float32)  # k odd: many ties
            m = Matcher(hi, lo, allow_low_quality_matches=lq)(torch.from_numpy(q.copy()))
            out["q_%d_%d" % (c, k)] = q
            out["m_%d_%d" % (c, k)] = t2n(m)
    out["cfgs"] = np.array([(a, b, float(c)) for a, b, c in cfgs], np.float32)
    save("model_matcher.npz", **out)


# ------------------------------------------------------------------ BoxList / IoU / LevelMapper
def gen_boxlist():
    rng = np.random.RandomState(1)
    W, H = 320, 200
    b = rand_boxes(rng, 40, W, H)
    bl = BoxList(torch.from_numpy(b.copy()), (W, H), "xyxy")
    out = {"boxes": b, "size": np.array([W, H])}
    out["xywh"] = t2n(bl.convert("xywh").bbox)
    out["xywh_back"] = t2n(bl.convert("xywh").convert("xyxy").bbox)
    out["resize_same"] = t2n(bl.resize((W * 2, H * 2)).bbox)
    out["resize_diff"] = t2n(bl.resize((480, 250)).bbox)
    out["flip_lr"] = t2n(bl.transpose(0).bbox)
    out["flip_tb"] = t2n(bl.transpose(1).bbox)
    out["crop"] = t2n(bl.crop((30, 20, 250, 160)).bbox)
    out["area"] = t2n(bl.area())
    out["area_xywh"] = t2n(bl.convert("xywh").area())
    big = BoxList(torch.from_numpy(b.copy()) * 1.5 - 40, (W, H), "xyxy")
    out["clip_in"] = t2n(big.bbox).copy()
    out["clip_out"] = t2n(BoxList(big.bbox.clone(), (W, H)).clip_to_image(remove_empty=False).bbox)
    kept = BoxList(big.bbox.clone(), (W, H)).clip_to_image(remove_empty=True)
    out["clip_kept"] = t2n(kept.bbox)
    b2 = rand_boxes(rng, 25, W, H)
    out["boxes2"] = b2
    out["iou"] = t2n(boxlist_iou(bl, BoxList(torch.from_numpy(b2), (W, H))))
    rois = rand_boxes(rng, 200, 1344, 800, smin=4, smax=800)
    out["lm_boxes"] = rois
    out["lm_levels"] = t2n(LevelMapper(2, 5)([BoxList(torch.from_numpy(rois), (1344, 800))]))
    save("model_boxlist.npz", **out)


# ------------------------------------------------------------------ target assignment
class FixedSampler(object):
    """deterministic stand-in: first `k` positives / negatives in index order."""

    def __init__(self, n_pos, n_neg):
        self.n_pos, self.n_neg = n_pos, n_neg

    def __call__(self, matched_idxs):
        pos, neg = [], []
        for l in matched_idxs:
            p = (l >= 1) & ((l >= 1).cumsum(0) <= self.n_pos)
            n = (l == 0) & ((l == 0).cumsum(0) <= self.n_neg)
            pos.append(p)
            neg.append(n)
        return pos, neg


def gen_targets():
    rng = np.random.RandomState(2)
    torch.manual_seed(2)
    W, H = 224, 160
    ag = AnchorGenerator(sizes=((32,), (64,), (128,)), anchor_strides=(8, 16, 32), straddle_thresh=0)
    feats = [torch.zeros(2, 1, H // s, W // s) for s in (8, 16, 32)]
    il = ImageList(torch.zeros(2, 3, H, W), [(H, W), (150, 200)])
    anchors = ag(il, feats)
    gts, labels = [], []
    targets = []
    for i, (h, w) in enumerate(il.image_sizes):
        n = (5, 3)[i]
        g = rand_boxes(rng, n, w, h, smin=20, smax=120)
        lab = rng.randint(1, 81, n)
        t = BoxList(torch.from_numpy(g), (w, h))
        t.add_field("labels", torch.from_numpy(lab))
        targets.append(t)
        gts.append(g)
        labels.append(lab)
    out = {"gt_0": gts[0], "gt_1": gts[1], "gt_labels_0": labels[0], "gt_labels_1": labels[1],
           "image_sizes": np.array(il.image_sizes), "canvas": np.array([H, W])}
    from maskrcnn_benchmark.structures.boxlist_ops import cat_boxlist
    cat_anchors = [cat_boxlist(a) for a in anchors]
    # RPN flavour
    rpn = RPNLossComputation(Matcher(0.7, 0.3, True), FixedSampler(16, 48), BoxCoder((1., 1., 1., 1.)), generate_rpn_labels)
    lab, reg = rpn.prepare_targets(cat_anchors, targets)
    for i in range(2):
        out["rpn_labels_%d" % i] = t2n(lab[i])
        out["rpn_reg_%d" % i] = t2n(reg[i])
    A = 3
    objectness = [torch.randn(2, A, f.shape[2], f.shape[3]) for f in feats]
    box_reg = [torch.randn(2, A * 4, f.shape[2], f.shape[3]) * 0.3 for f in feats]
    lo, lb = rpn(anchors, objectness, box_reg, targets)
    for l in range(3):
        out["objectness_%d" % l] = t2n(objectness[l])
        out["box_reg_%d" % l] = t2n(box_reg[l])
    out["rpn_loss"] = np.array([float(lo), float(lb)], np.float64)
    # RetinaNet flavour (labels = class of the matched gt; only 'between_thresholds' discarded)
    ret = RPNLossComputation(Matcher(0.5, 0.4, True), None, BoxCoder((10., 10., 5., 5.)), generate_retinanet_labels)
    ret.copied_fields = ["labels"]
    ret.discard_cases = ["between_thresholds"]
    lab, reg = ret.prepare_targets(cat_anchors, targets)
    for i in range(2):
        out["ret_labels_%d" % i] = t2n(lab[i])
        out["ret_reg_%d" % i] = t2n(reg[i])
    # Fast R-CNN head flavour
    frc = FastRCNNLossComputation(Matcher(0.5, 0.5, False), FixedSampler(8, 24), BoxCoder((10., 10., 5., 5.)))
    props = []
    for i, (h, w) in enumerate(il.image_sizes):
        p = np.concatenate([rand_boxes(rng, 60, w, h, smin=10, smax=140), gts[i] + rng.uniform(-4, 4, gts[i].shape).astype(np.float32), gts[i]])
        p = np.clip(p, 0, [w - 1, h - 1, w - 1, h - 1]).astype(np.float32)
        props.append(BoxList(torch.from_numpy(p), (w, h)))
        out["props_%d" % i] = p
    lab, reg = frc.prepare_targets(props, targets)
    for i in range(2):
        out["frc_labels_%d" % i] = t2n(lab[i])
        out["frc_reg_%d" % i] = t2n(reg[i])
    save("model_targets.npz", **out)


# ------------------------------------------------------------------ mask targets
def gen_masks():
    rng = np.random.RandomState(3)
    W, H = 200, 144
    n = 6
    g = rand_boxes(rng, n, W, H, smin=24, smax=110)
    yy = np.arange(H, dtype=np.float32)[None, :, None]
    xx = np.arange(W, dtype=np.float32)[None, None, :]
    cx, cy = (g[:, 0] + g[:, 2]) / 2, (g[:, 1] + g[:, 3]) / 2
    rw, rh = (g[:, 2] - g[:, 0]) / 2 + 0.5, (g[:, 3] - g[:, 1]) / 2 + 0.5
    masks = ((((xx - cx[:, None, None]) / rw[:, None, None]) ** 2 + ((yy - cy[:, None, None]) / rh[:, None, None]) ** 2) <= 1).astype(np.uint8)
    P = 40
    which = rng.randint(0, n, P)
    props = g[which] + rng.uniform(-10, 10, (P, 4)).astype(np.float32)
    props[:, 2:] = np.maximum(props[:, 2:], props[:, :2] + 1)
    props[-3:] += np.array([[-60, -60, 80, 80]], np.float32)  # spill outside the image
    out = {"masks": masks, "props": props, "which": which, "size": np.array([W, H])}
    for dtype, tag in ((torch.uint8, "u8"), (torch.float32, "f32")):
        seg = SegmentationMask(torch.from_numpy(masks).to(dtype), (W, H), mode="mask")
        sel = seg[torch.from_numpy(which)]
        t = project_masks_on_boxes(sel, BoxList(torch.from_numpy(props), (W, H)), 28)
        out["targets_" + tag] = t2n(t)
    save("model_masks.npz", **out)


# ------------------------------------------------------------------ proposal selection
def gen_proposals():
    torch.manual_seed(4)
    W, H = 224, 160
    ag = AnchorGenerator(sizes=((32,), (64,), (128,)), anchor_strides=(8, 16, 32), straddle_thresh=0)
    feats = [torch.zeros(2, 1, H // s, W // s) for s in (8, 16, 32)]
    il = ImageList(torch.zeros(2, 3, H, W), [(H, W), (150, 200)])
    anchors = ag(il, feats)
    A = 3
    objectness = [torch.randn(2, A, f.shape[2], f.shape[3]) * 2 for f in feats]
    box_reg = [torch.randn(2, A * 4, f.shape[2], f.shape[3]) * 0.5 for f in feats]
    out = {"image_sizes": np.array(il.image_sizes), "canvas": np.array([H, W])}
    for l in range(3):
        out["objectness_%d" % l] = t2n(objectness[l])
        out["box_reg_%d" % l] = t2n(box_reg[l])
    settings = {"train": dict(pre_nms_top_n=100, post_nms_top_n=60, nms_thresh=0.7, min_size=0, fpn_post_nms_top_n=90,
                              fpn_post_nms_per_batch=True),
                "train_perimg": dict(pre_nms_top_n=100, post_nms_top_n=60, nms_thresh=0.7, min_size=0, fpn_post_nms_top_n=50,
                                     fpn_post_nms_per_batch=False),
                "test": dict(pre_nms_top_n=80, post_nms_top_n=40, nms_thresh=0.7, min_size=4, fpn_post_nms_top_n=50,
                             fpn_post_nms_per_batch=True)}
    for tag, kw in settings.items():
        pp = RPNPostProcessor(box_coder=BoxCoder((1., 1., 1., 1.)), **kw)
        pp.train(tag.startswith("train"))
        res = pp(anchors, objectness, box_reg, None)
        for i, r in enumerate(res):
            o = r.get_field("objectness")
            order = torch.argsort(o, descending=True, stable=True)
            out["%s_boxes_%d" % (tag, i)] = t2n(r.bbox[order])
            out["%s_scores_%d" % (tag, i)] = t2n(o[order])
        out["%s_cfg" % tag] = np.array([kw["pre_nms_top_n"], kw["post_nms_top_n"], kw["min_size"], kw["fpn_post_nms_top_n"],
                                         int(kw["fpn_post_nms_per_batch"])])
    save("model_proposals.npz", **out)


# ------------------------------------------------------------------ evaluation post-processing
def _canon(r):
    """detections of one image in a canonical order: label ascending, score descending"""
    s, l = r.get_field("scores"), r.get_field("labels")
    order = np.lexsort((-t2n(s).astype(np.float64), t2n(l)))
    return t2n(r.bbox)[order], t2n(s)[order], t2n(l)[order]


def gen_postprocess():
    from maskrcnn_benchmark.modeling.roi_heads.box_head.inference import PostProcessor
    from maskrcnn_benchmark.modeling.rpn.retinanet.inference import RetinaNetPostProcessor
    out = {}
    # ---- Fast R-CNN box head: two images, C = 9 classes (8 + background)
    torch.manual_seed(6)
    rng = np.random.RandomState(6)
    C = 9
    sizes = [(320, 240), (300, 200)]                       # (W, H)
    counts = [150, 90]
    props = [torch.from_numpy(rand_boxes(rng, n, w, h, smin=10, smax=150)) for n, (w, h) in zip(counts, sizes)]
    logits = torch.randn(sum(counts), C) * 2.5
    reg = torch.randn(sum(counts), 4 * C) * 0.6
    for tag, kw in {"a": dict(score_thresh=0.05, nms=0.5, detections_per_img=100),
                    "b": dict(score_thresh=0.01, nms=0.3, detections_per_img=25)}.items():
        pp = PostProcessor(box_coder=BoxCoder((10., 10., 5., 5.)), **kw)
        res = pp((logits, reg), [BoxList(p, s, mode="xyxy") for p, s in zip(props, sizes)])
        for i, r in enumerate(res):
            b, sc, lb = _canon(r)
            out["box_%s_boxes_%d" % (tag, i)], out["box_%s_scores_%d" % (tag, i)], out["box_%s_labels_%d" % (tag, i)] = b, sc, lb
        out["box_%s_cfg" % tag] = np.array([kw["score_thresh"], kw["nms"], kw["detections_per_img"]])
    out["box_logits"], out["box_reg"] = t2n(logits), t2n(reg)
    out["box_sizes"], out["box_counts"] = np.array(sizes), np.array(counts)
    for i, p in enumerate(props):
        out["box_props_%d" % i] = t2n(p)
    # ---- RetinaNet: 3 levels, A = 9 anchors, 5 classes (+ background)
    W, H = 224, 160
    nc = 6
    sz = tuple(tuple(s * 2 ** (k / 3.0) for k in range(3)) for s in (32, 64, 128))
    ag = AnchorGenerator(sizes=sz, aspect_ratios=(0.5, 1.0, 2.0), anchor_strides=(8, 16, 32), straddle_thresh=-1)
    feats = [torch.zeros(2, 1, H // s, W // s) for s in (8, 16, 32)]
    il = ImageList(torch.zeros(2, 3, H, W), [(H, W), (150, 200)])
    anchors = ag(il, feats)
    A = 9
    box_cls = [torch.randn(2, A * (nc - 1), f.shape[2], f.shape[3]) * 2 - 3 for f in feats]
    box_reg = [torch.randn(2, A * 4, f.shape[2], f.shape[3]) * 0.4 for f in feats]
    for l in range(3):
        out["ret_cls_%d" % l], out["ret_reg_%d" % l] = t2n(box_cls[l]), t2n(box_reg[l])
    out["ret_image_sizes"], out["ret_canvas"] = np.array(il.image_sizes), np.array([H, W])
    for tag, kw in {"a": dict(pre_nms_thresh=0.05, pre_nms_top_n=60, nms_thresh=0.4, fpn_post_nms_top_n=40, min_size=0),
                    "b": dict(pre_nms_thresh=0.2, pre_nms_top_n=1000, nms_thresh=0.5, fpn_post_nms_top_n=100, min_size=0)}.items():
        pp = RetinaNetPostProcessor(num_classes=nc, box_coder=BoxCoder((10., 10., 5., 5.)), **kw)
        pp.eval()
        res = pp(anchors, box_cls, box_reg)
        for i, r in enumerate(res):
            b, sc, lb = _canon(r)
            out["ret_%s_boxes_%d" % (tag, i)], out["ret_%s_scores_%d" % (tag, i)], out["ret_%s_labels_%d" % (tag, i)] = b, sc, lb
        out["ret_%s_cfg" % tag] = np.array([kw["pre_nms_thresh"], kw["pre_nms_top_n"], kw["nms_thresh"], kw["fpn_post_nms_top_n"]])
    save("model_postprocess.npz", **out)


# ------------------------------------------------------------------ solver bits
def gen_solver():
    from maskrcnn_benchmark.layers import smooth_l1_loss
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=0.02)
    sched = WarmupMultiStepLR(opt, (30, 40), 0.1, warmup_factor=1.0 / 3, warmup_iters=10, warmup_method="linear")
    lrs = []
    for _ in range(50):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    torch.manual_seed(5)
    a, b = torch.randn(50, 4), torch.randn(50, 4)
    save("model_solver.npz", lrs=np.array(lrs), sl1_a=t2n(a), sl1_b=t2n(b),
         sl1=np.array([float(smooth_l1_loss(a, b, beta=1.0 / 9, size_average=False)),
                       float(smooth_l1_loss(a, b, beta=1.0, size_average=True)),
                       float(smooth_l1_loss(a, b, beta=0.11, size_average=False))]))


if __name__ == "__main__":
    gen_anchors()
    gen_matcher()
    gen_boxlist()
    gen_targets()
    gen_masks()
    gen_proposals()
    gen_postprocess()
    gen_solver()
