"""Generate tests/golden/model_targets_fullsize.npz: the REFERENCE's own target assignment at the BASELINE anchor sets
(build container only; never runs on the GPU box).

VERDICT r04 "weak" #8 / next-round #9: the GPU tests of the match / label / sampler kernels at 268,569 (RPN) and 201,600
(RetinaNet) anchors per image compared with this repository's own ATen composition — a shared misreading of
modeling/matcher.py at scale would pass.  Here the reference's modules themselves run at that size
(modeling/rpn/anchor_generator.py, structures/boxlist_ops.py: boxlist_iou, modeling/matcher.py:42-112,
modeling/rpn/loss.py:42-88 prepare_targets, modeling/rpn/retinanet/loss.py, box_coder.py:27-51,
balanced_positive_negative_sampler.py:22-68 for the quotas) and what they produce is stored COMPACTLY:

  per flavour (rpn | retinanet) and image i:
    anchors_sha            sha256 of the concatenated anchor grid (fp32 bytes) + 16 sampled rows
    matched_sha_i          sha256 of Matcher's output (int64 [K] bytes), counts of -1 / -2,
    matched_idx_i / matched_val_i     the anchors with a match >= 0 and their ground-truth index (sparse)
    labels_sha_i           sha256 of prepare_targets' labels as int8 [K]; labels_pos_i the positive anchors,
                           labels_posval_i their label value (RetinaNet: the class), labels_ignored_i the count of -1
    reg_pos_i              the regression targets of the positive anchors, fp32 [P, 4]
    quota_i                (num_pos, num_neg) the reference's sampler would draw (RPN: 256 per image, half positive)

The inputs (two images of different size, 23 / 9 ground-truth boxes, one of them identical to an anchor) are stored too.
Run:  python tests/golden/make_golden_fullsize.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_model as G  # noqa: E402  (installs the reference's modules; generates nothing on import)

from maskrcnn_benchmark.structures.boxlist_ops import cat_boxlist  # noqa: E402  (the reference's)

CANVAS = (800, 1344)
SIZES = [(800, 1344), (771, 1203)]          # (h, w) per image


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gt_boxes(rng, M, W, H):
    cx, cy = rng.uniform(0, W, M), rng.uniform(0, H, M)
    w, h = rng.uniform(8, 500, M), rng.uniform(8, 400, M)
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1)
    return np.clip(b, 0, [W - 1, H - 1, W - 1, H - 1]).astype(np.float32)


def flavour(name, out):
    rng = np.random.RandomState({"rpn": 5, "retinanet": 6}[name])
    if name == "rpn":
        strides, sizes = (4, 8, 16, 32, 64), ((32,), (64,), (128,), (256,), (512,))
        matcher, coder = G.Matcher(0.7, 0.3, allow_low_quality_matches=True), G.BoxCoder((1.0, 1.0, 1.0, 1.0))
        labels_func, copied, discard = G.generate_rpn_labels, [], ["not_visibility", "between_thresholds"]
    else:
        strides = (8, 16, 32, 64, 128)
        sizes = tuple(tuple(s * 2 ** (k / 3.0) for k in range(3)) for s in (32, 64, 128, 256, 512))
        matcher, coder = G.Matcher(0.5, 0.4, allow_low_quality_matches=True), G.BoxCoder((10.0, 10.0, 5.0, 5.0))
        labels_func, copied, discard = G.generate_retinanet_labels, ["labels"], ["between_thresholds"]
    ag = G.AnchorGenerator(sizes=sizes, anchor_strides=strides, straddle_thresh=0)
    feats = [torch.zeros(2, 1, -(-CANVAS[0] // s), -(-CANVAS[1] // s)) for s in strides]
    anchors = ag(G.ImageList(torch.zeros(2, 3, *CANVAS), SIZES), feats)
    cat_anchors = [cat_boxlist(a) for a in anchors]
    K = len(cat_anchors[0])
    assert K == {"rpn": 268569, "retinanet": 201600}[name], K
    grid = G.t2n(cat_anchors[0].bbox).astype(np.float32)
    out[name + "_anchors_sha"] = np.array(sha(grid))
    rows = np.linspace(0, K - 1, 16).astype(np.int64)
    out[name + "_anchor_rows"], out[name + "_anchor_vals"] = rows, grid[rows]
    targets = []
    for i, ((h, w), m) in enumerate(zip(SIZES, (23, 9))):
        g = gt_boxes(rng, m, w, h)
        if i == 0:
            g[4] = grid[12345]                    # a ground truth identical to an anchor: IoU exactly 1
        lab = rng.randint(1, 81, m).astype(np.int64)
        t = G.BoxList(torch.from_numpy(g.copy()), (w, h))
        t.add_field("labels", torch.from_numpy(lab.copy()))
        targets.append(t)
        out["%s_gt_%d" % (name, i)], out["%s_gt_labels_%d" % (name, i)] = g, lab
    ev = G.RPNLossComputation(matcher, None, coder, labels_func)
    ev.copied_fields, ev.discard_cases = copied, discard
    labels, reg = ev.prepare_targets(cat_anchors, targets)
    for i in range(2):
        matched = ev.match_targets_to_anchors(cat_anchors[i], targets[i], copied).get_field("matched_idxs")
        mi = G.t2n(matched).astype(np.int64)
        out["%s_matched_sha_%d" % (name, i)] = np.array(sha(mi))
        out["%s_matched_counts_%d" % (name, i)] = np.array([(mi >= 0).sum(), (mi == -1).sum(), (mi == -2).sum()])
        nz = np.nonzero(mi >= 0)[0]
        out["%s_matched_idx_%d" % (name, i)], out["%s_matched_val_%d" % (name, i)] = nz.astype(np.int32), mi[nz].astype(np.int16)
        lab = G.t2n(labels[i]).astype(np.int8)
        out["%s_labels_sha_%d" % (name, i)] = np.array(sha(lab))
        pos = np.nonzero(lab > 0)[0]
        out["%s_labels_pos_%d" % (name, i)], out["%s_labels_posval_%d" % (name, i)] = pos.astype(np.int32), lab[pos]
        out["%s_labels_ignored_%d" % (name, i)] = np.array(int((lab < 0).sum()))
        out["%s_reg_pos_%d" % (name, i)] = G.t2n(reg[i]).astype(np.float32)[pos]
        if name == "rpn":        # balanced_positive_negative_sampler.py:38-48 with 256 per image, half positive
            n_pos = min(int((lab >= 1).sum()), 128)
            n_neg = min(int((lab == 0).sum()), 256 - n_pos)
            out["%s_quota_%d" % (name, i)] = np.array([n_pos, n_neg])
        print(name, i, "K", K, "matched >= 0:", len(nz), "positives:", len(pos), "ignored:", int((lab < 0).sum()))


if __name__ == "__main__":
    out = {"canvas": np.array(CANVAS), "image_sizes": np.array(SIZES)}
    flavour("rpn", out)
    flavour("retinanet", out)
    G.save("model_targets_fullsize.npz", **out)
    print("bytes:", os.path.getsize(os.path.join(HERE, "model_targets_fullsize.npz")))
