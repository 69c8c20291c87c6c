"""Seeded synthetic workloads for the detection-head hot path (numpy only).

Shapes follow SURVEY.md §8(d) / BASELINE.md §3: the same generators feed the CPU oracle, the
HIP kernels, the parity tests and bench.py, so every comparison is on identical inputs.
"""
import math

import numpy as np

IMG_H, IMG_W = 800, 1344  # 1333x800 padded to SIZE_DIVISIBILITY=32
FPN_STRIDES = (4, 8, 16, 32, 64)


def fpn_shapes(strides=FPN_STRIDES, img_h=IMG_H, img_w=IMG_W):
    return [(int(math.ceil(img_h / s)), int(math.ceil(img_w / s))) for s in strides]


# ------------------------------------------------------------------------------------------
# cfg-1: ROIAlign micro case (BASELINE.json configs[0])
# ------------------------------------------------------------------------------------------
def cfg1_roi_align(seed=0, K=512, C=256, H=14, W=14):
    rng = np.random.RandomState(seed)
    inp = rng.randn(1, C, H, W).astype(np.float32)
    x1 = rng.uniform(0, 150, K)
    y1 = rng.uniform(0, 150, K)
    w = rng.uniform(1, 101, K)
    h = rng.uniform(1, 101, K)
    rois = np.stack([np.zeros(K), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    return inp, rois, 1.0 / 16


# ------------------------------------------------------------------------------------------
# cfg-2/3: FPN box-head / mask-head ROIs on the 2-image 800x1344 pyramid
# ------------------------------------------------------------------------------------------
def fpn_rois(seed=3, per_image=512, n_images=2, smin=16.0, smax=800.0):
    """[K,5] rois: sqrt(area) log-uniform in [smin,smax], aspect U(0.5,2), centres uniform."""
    rng = np.random.RandomState(seed)
    out = []
    for b in range(n_images):
        s = np.exp(rng.uniform(math.log(smin), math.log(smax), per_image))
        ar = rng.uniform(0.5, 2.0, per_image)
        w = s * np.sqrt(ar)
        h = s / np.sqrt(ar)
        cx = rng.uniform(0, IMG_W, per_image)
        cy = rng.uniform(0, IMG_H, per_image)
        x1 = np.clip(cx - w / 2, 0, IMG_W - 1)
        y1 = np.clip(cy - h / 2, 0, IMG_H - 1)
        x2 = np.clip(cx + w / 2, 0, IMG_W - 1)
        y2 = np.clip(cy + h / 2, 0, IMG_H - 1)
        out.append(np.stack([np.full(per_image, b), x1, y1, x2, y2], 1))
    return np.concatenate(out, 0).astype(np.float32)


def fpn_rois_trained_like(seed=7, per_image=512, n_images=2, fg_fraction=0.25, n_gt=(8, 20)):
    """[K,5] rois shaped like the box head's input late in training: per image 8-20 ground-truth boxes
    (the synthetic COCO-shaped generator's size range), `fg_fraction` of the ROIs are jitters of a
    ground-truth box (IoU >~ 0.5: centre +-10 % of the side, log-size N(0, 0.15)), the rest are
    proposal-like boxes clustered AROUND the objects (centre within 1.5 sides, log-size N(0, 0.5)) —
    what an RPN that has learnt objectness emits — positives first per image, like the sampler."""
    rng = np.random.RandomState(seed)
    out = []
    for b in range(n_images):
        m = rng.randint(n_gt[0], n_gt[1] + 1)
        gs = np.exp(rng.uniform(math.log(32), math.log(480), m))
        gar = rng.uniform(0.5, 2.0, m)
        gw, gh = gs * np.sqrt(gar), gs / np.sqrt(gar)
        gcx = rng.uniform(0.1 * IMG_W, 0.9 * IMG_W, m)
        gcy = rng.uniform(0.1 * IMG_H, 0.9 * IMG_H, m)
        n_fg = int(round(per_image * fg_fraction))
        which = rng.randint(0, m, per_image)
        fg = np.arange(per_image) < n_fg
        cj = np.where(fg, 0.1, 1.5)
        sj = np.where(fg, 0.15, 0.5)
        cx = gcx[which] + rng.uniform(-1, 1, per_image) * cj * gw[which]
        cy = gcy[which] + rng.uniform(-1, 1, per_image) * cj * gh[which]
        w = gw[which] * np.exp(rng.randn(per_image) * sj)
        h = gh[which] * np.exp(rng.randn(per_image) * sj)
        x1 = np.clip(cx - w / 2, 0, IMG_W - 1)
        y1 = np.clip(cy - h / 2, 0, IMG_H - 1)
        x2 = np.clip(cx + w / 2, 0, IMG_W - 1)
        y2 = np.clip(cy + h / 2, 0, IMG_H - 1)
        out.append(np.stack([np.full(per_image, b), x1, y1, np.maximum(x2, x1), np.maximum(y2, y1)], 1))
    return np.concatenate(out, 0).astype(np.float32)


def roi_sets(model_npz=None):
    """name -> {"box": [K,5], "mask": [K',5]} ROI sets for tools/opbench.py: the log-uniform set of SURVEY.md 8d,
    the trained-like set, and (when the file exists) the sets the detector itself produced in a training step
    (tools/dump_model_rois.py, committed under tests/golden/model_rois.npz)."""
    import os
    sets = {"synthetic-loguniform": {"box": fpn_rois(per_image=512), "mask": fpn_rois(per_image=128)},
            "trained-like": {"box": fpn_rois_trained_like(per_image=512),
                             "mask": fpn_rois_trained_like(per_image=128, fg_fraction=1.0)}}
    path = model_npz or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "model_rois.npz")
    if os.path.exists(path):
        z = np.load(path)
        sets["model-random-init"] = {"box": z["box_rois"].astype(np.float32), "mask": z["mask_rois"].astype(np.float32)}
    return sets


def level_map(rois, k_min=2, k_max=5, s0=224.0, lvl0=4.0, eps=1e-6):
    """LevelMapper (reference modeling/poolers.py:33-42) in fp32, 0-based level index."""
    r = rois.astype(np.float32)
    area = (r[:, 3] - r[:, 1] + np.float32(1)) * (r[:, 4] - r[:, 2] + np.float32(1))
    s = np.sqrt(area, dtype=np.float32)
    t = np.floor(np.float32(lvl0) + np.log2(s / np.float32(s0) + np.float32(eps), dtype=np.float32))
    return (np.clip(t, k_min, k_max) - k_min).astype(np.int32)


def fpn_features(seed=4, n_images=2, C=256, levels=4):
    rng = np.random.RandomState(seed)
    return [rng.randn(n_images, C, h, w).astype(np.float32) for (h, w) in fpn_shapes()[:levels]]


# ------------------------------------------------------------------------------------------
# NMS segments (RPN proposals per (image, level)): clustered boxes, distinct scores
# ------------------------------------------------------------------------------------------
def nms_boxes(n, seed=2, clusters=150, jitter=12.0, uniform=False, distinct_scores=True):
    rng = np.random.RandomState(seed)
    if uniform:  # worst case: ~97 % survive
        cx = rng.uniform(0, IMG_W, n)
        cy = rng.uniform(0, IMG_H, n)
        w = rng.uniform(8, 64, n)
        h = rng.uniform(8, 64, n)
    else:
        k = max(1, min(clusters, n))
        ccx = rng.uniform(0, IMG_W, k)
        ccy = rng.uniform(0, IMG_H, k)
        cs = np.exp(rng.uniform(math.log(24), math.log(320), k))
        which = rng.randint(0, k, n)
        cx = ccx[which] + rng.randn(n) * jitter
        cy = ccy[which] + rng.randn(n) * jitter
        w = cs[which] * np.exp(rng.randn(n) * 0.15)
        h = cs[which] * np.exp(rng.randn(n) * 0.15) * rng.uniform(0.6, 1.6, n)
    x1 = np.clip(cx - w / 2, 0, IMG_W - 1)
    y1 = np.clip(cy - h / 2, 0, IMG_H - 1)
    x2 = np.clip(cx + w / 2, 0, IMG_W - 1)
    y2 = np.clip(cy + h / 2, 0, IMG_H - 1)
    boxes = np.stack([x1, y1, x2, y2], 1).astype(np.float32)
    if distinct_scores:
        vals = np.linspace(0.001, 0.999, n, dtype=np.float64).astype(np.float32)
        assert len(np.unique(vals)) == n
        scores = vals[rng.permutation(n)]
    else:
        scores = rng.rand(n).astype(np.float32)
    return boxes, scores


def rpn_nms_segments(seed=2, n_images=2):
    """10 segments (5 levels x 2 images), n = min(2000, 3*H*W) like PRE_NMS_TOP_N_TRAIN."""
    segs = []
    for b in range(n_images):
        for li, (h, w) in enumerate(fpn_shapes()):
            n = min(2000, 3 * h * w)
            segs.append(nms_boxes(n, seed=seed + 17 * b + li))
    return segs


# ------------------------------------------------------------------------------------------
# cfg-4: RetinaNet focal-loss stream
# ------------------------------------------------------------------------------------------
def focal_inputs(R, C=80, seed=5):
    rng = np.random.RandomState(seed)
    logits = np.clip(rng.randn(R, C) * 2.0 - 2.0, -12, 12).astype(np.float32)
    # most logits sit at the prior-probability bias (PRIOR_PROB 0.01 -> -4.6)
    prior = rng.rand(R, C) < 0.7
    logits[prior] = (-4.6 + 0.3 * rng.randn(int(prior.sum()))).astype(np.float32)
    u = rng.rand(R)
    targets = np.zeros(R, np.int32)
    pos = u < 0.002
    targets[pos] = rng.randint(1, C + 1, int(pos.sum()))
    targets[(u >= 0.002) & (u < 0.022)] = -1
    return logits, targets


# ------------------------------------------------------------------------------------------
# cfg-5: deformable conv blocks
# ------------------------------------------------------------------------------------------
def dcn_inputs(B, C, H, W, Cout, k=3, dg=1, modulated=False, seed=6, dtype=np.float32):
    rng = np.random.RandomState(seed)
    x = rng.randn(B, C, H, W).astype(dtype)
    off = rng.randn(B, dg * 2 * k * k, H, W) * 2.0
    tail = rng.rand(B, dg * 2 * k * k, H, W) < 0.05  # heavy tail reaching the borders
    off[tail] = rng.randn(int(tail.sum())) * 10.0
    off = off.astype(dtype)
    mask = None
    if modulated:
        mask = (1.0 / (1.0 + np.exp(-rng.randn(B, dg * k * k, H, W)))).astype(dtype)
    stdv = 1.0 / math.sqrt(C * k * k)
    wgt = rng.uniform(-stdv, stdv, (Cout, C, k, k)).astype(dtype)
    return x, off, mask, wgt
