"""Per-operator micro-benchmarks of the HIP detection-head kernels (HIP-event timed on the stream
the kernels are launched on) with the ALGORITHMIC byte counts of SURVEY.md §8(d).

    python tools/opbench.py [--iters 50] [--json out.json] [--only roi_align_fwd,...]

Used by bench.py for the `roofline` object and by the profiling recipes in profiles/README.md.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling


def dev_time_us(fn, iters=50, warmup=5):
    """Average device time of fn() in microseconds, HIP events on the current stream."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) * 1000.0 / iters


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _entry(name, us, alg_bytes, extra=None):
    e = {"op": name, "us": round(us, 2), "alg_bytes": int(alg_bytes),
         "gbs": round(alg_bytes / us / 1e3, 1), "frac_of_8TBs": round(alg_bytes / us / 1e3 / HBM_PEAK_GBS, 4)}
    if extra:
        e.update(extra)
    return e


def tune(key, value):
    """library tuning / test switch (include/detops.h: detops_tuning_set)"""
    from maskrcnn_benchmark import _lib
    _lib.tuning_set(key, value)



def bench_roi_align(C, iters, which=("fwd", "bwd"), fused_only=False, experimental=False):
    """fused_only: just the FPN-fused launches with the model's shapes (the PMC traffic passes use this,
    so every ROIAlign kernel in their trace is the box-head / mask-head launch bench.py times)."""
    out = []
    # cfg-1 (BASELINE configs[0]) ---------------------------------------------------------
    inp, rois, scale = synth.cfg1_roi_align()
    ti, tr = _t(inp), _t(rois)
    for ph, pw, sr in ([] if fused_only else [(7, 7, 2), (14, 14, 2), (7, 7, 0)]):
        K, Cc = rois.shape[0], inp.shape[1]
        alg = 4 * K * Cc * ph * pw + 4 * inp.size + 20 * K
        if "fwd" in which:
            us = dev_time_us(lambda: C.roi_align_forward(ti, tr, scale, ph, pw, sr), iters)
            out.append(_entry(f"roi_align_fwd cfg1 {ph}x{pw} sr{sr}", us, alg))
        if "bwd" in which:
            g = torch.randn(K, Cc, ph, pw, device="cuda")
            us = dev_time_us(lambda: C.roi_align_backward(g, tr, scale, ph, pw, 1, Cc, 14, 14, sr), iters)
            out.append(_entry(f"roi_align_bwd cfg1 {ph}x{pw} sr{sr} (gather, ROI list split over 32 groups)", us, alg))
            for grp in (1, 16):
                tune("roi_bwd_groups", grp)
                us = dev_time_us(lambda: C.roi_align_backward(g, tr, scale, ph, pw, 1, Cc, 14, 14, sr), max(3, iters // 5))
                out.append(_entry(f"roi_align_bwd cfg1 {ph}x{pw} sr{sr} [groups={grp}]", us, alg))
            tune("roi_bwd_groups", 0)
    # cfg-2 box head: 1024 ROIs over P2..P5, 7x7 sr2; cfg-3 mask head: 256 ROIs, 14x14 sr2 ----
    feats = [torch.randn(2, 256, h, w, device="cuda") for (h, w) in synth.fpn_shapes()[:4]]
    scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
    shapes = [tuple(f.shape) for f in feats]
    feat_bytes = sum(f.numel() * 4 for f in feats)
    for tag, K, ph in (("box-head 1024x7x7", 1024, 7), ("mask-head 256x14x14", 256, 14)):
        rois = synth.fpn_rois(per_image=K // 2)
        tr = _t(rois)
        alg = 4 * K * 256 * ph * ph + feat_bytes + 20 * K
        lv = synth.level_map(rois)
        per = [(l, _t(rois[lv == l])) for l in range(4)]

        def per_level_fwd():
            for l, r in per:
                C.roi_align_forward(feats[l], r, scales[l], ph, ph, 2)

        if "fwd" in which:
            us = dev_time_us(lambda: C.roi_align_fpn_forward(feats, tr, scales, ph, ph, 2, 2, 5), iters)
            out.append(_entry(f"roi_align_fwd fpn-fused {tag}", us, alg))
            if not fused_only:
                us = dev_time_us(per_level_fwd, iters)
                out.append(_entry(f"roi_align_fwd fpn-per-level(4 launches) {tag}", us, alg))
        if "fwd" in which and not fused_only:
            base = C.roi_align_fpn_forward(feats, tr, scales, ph, ph, 2, 2, 5)[0]
            for label, key, val in (("LDS-DMA, ROIs in caller order (no ranking pre-pass)", "roi_fwd_order", 1),
                                    ("generic gather kernel", "roi_fwd_impl", 1)):
                tune(key, val)
                us = dev_time_us(lambda: C.roi_align_fpn_forward(feats, tr, scales, ph, ph, 2, 2, 5), iters)
                same = bool(torch.equal(base, C.roi_align_fpn_forward(feats, tr, scales, ph, ph, 2, 2, 5)[0]))
                tune(key, 0)
                out.append(_entry(f"roi_align_fwd fpn-fused {tag} [{label}]", us, alg, {"bit_equal_to_default": same}))
        if "bwd" in which:
            g = torch.randn(K, 256, ph, ph, device="cuda")
            tl = _t(lv)
            us = dev_time_us(lambda: C.roi_align_fpn_backward(g, tr, tl, shapes, scales, ph, ph, 2), iters)
            out.append(_entry(f"roi_align_bwd fpn-fused ring pixel-owner (atomic-free, incl. pre-pass + zero-fill) {tag}", us, alg))
            if fused_only:
                continue
            ref_g = C.roi_align_fpn_backward(g, tr, tl, shapes, scales, ph, ph, 2)
            for label, key, val in (("ring, hit lists never split", "roi_bwd_seg", 1 << 20), ("ring, segments of 16 hits", "roi_bwd_seg", 16),
                                    ("scan pixel-owner (no pre-pass)", "roi_bwd_impl", 2), ("atomic scatter", "roi_bwd_impl", 3)):
                tune(key, val)
                us = dev_time_us(lambda: C.roi_align_fpn_backward(g, tr, tl, shapes, scales, ph, ph, 2), max(3, iters // 5))
                got = C.roi_align_fpn_backward(g, tr, tl, shapes, scales, ph, ph, 2)
                tune(key, 0)
                out.append(_entry(f"roi_align_bwd fpn-fused [{label}] {tag}", us, alg,
                                  {"max_abs_diff_vs_default": max(float((a - b).abs().max()) for a, b in zip(ref_g, got))}))
    return out


def bench_roi_sets(C, iters, model_npz=None, only_sets=None, only_heads=None, which=("fwd", "bwd"), images=2, tag="", layout="nchw"):
    """FPN-fused ROIAlign forward / backward on every ROI set of synth.roi_sets(): the SURVEY 8d log-uniform set, the
    trained-like set and the sets the detector itself produced (tools/dump_model_rois.py) - VERDICT r02 item 1a."""
    out = []
    feats = [torch.randn(images, 256, h, w, device="cuda") for (h, w) in synth.fpn_shapes()[:4]]
    nhwc = layout == "nhwc"    # channels-last pyramid: csrc/roi_align_nhwc.hip (box head: contiguous pooled tensor / gradient,
    if nhwc:                   # mask head: channels-last pooled tensor / gradient, as in the detector)
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
    shapes = [tuple(f.shape) for f in feats]
    feat_bytes = sum(f.numel() * 4 for f in feats)
    for name, sets in synth.roi_sets(model_npz).items():
        if only_sets and name not in only_sets:
            continue
        for head, ph in (("box", 7), ("mask", 14)):
            if only_heads and head not in only_heads:
                continue
            rois = sets[head]
            label = name
            if images != 2:
                # the > L3 variant of SURVEY 8d (4 img/GPU: 365.6 MB of maps against the 256 MiB Infinity Cache): the
                # 2-image ROI set repeated on image pairs (2, 3), ... — same per-image ROI statistics
                reps = []
                for r in range((images + 1) // 2):
                    rr = rois.copy()
                    rr[:, 0] = np.minimum(rr[:, 0] + 2 * r, images - 1)
                    reps.append(rr)
                rois = np.concatenate(reps)[: rois.shape[0] * images // 2]
                label = label + " x%d images" % images
            if tag:
                label = label + " | " + tag
            K = rois.shape[0]
            tr = _t(rois)
            lv = synth.level_map(rois)
            tl = _t(lv)
            alg = 4 * K * 256 * ph * ph + feat_bytes + 20 * K
            g = torch.randn(K, 256, ph, ph, device="cuda")
            ocl = nhwc and head == "mask"
            if ocl:
                g = g.contiguous(memory_format=torch.channels_last)
            if nhwc:
                label = label + " | nhwc"
            extra = {"rois_per_level": np.bincount(lv, minlength=4).tolist()}
            if "fwd" in which:
                us = dev_time_us(lambda: C.roi_align_fpn_forward(feats, tr, scales, ph, ph, 2, 2, 5, out_channels_last=ocl), iters)
                out.append(_entry(f"roi_align_fwd fpn-fused {head}-head K={K} {ph}x{ph} [{label}]", us, alg, extra))
            if "bwd" in which:
                us = dev_time_us(lambda: C.roi_align_fpn_backward(g, tr, tl, shapes, scales, ph, ph, 2, channels_last=nhwc), iters)
                out.append(_entry(f"roi_align_bwd fpn-fused {head}-head K={K} {ph}x{ph} [{label}]", us, alg, extra))
    return out


def bench_roi_pool_psroi(C, iters):
    """The two API-only pooling operators (no reference config instantiates them), once, at the box
    head's size on the stride-16 level: ROIPool 7x7 (reference csrc/cuda/ROIPool_cuda.cu:16-202) and deformable PS-ROI
    pooling (csrc/cuda/deform_pool_kernel_cuda.cu:53-264; 7x7 parts, 4 samples per part, 10 output channels per group)."""
    out = []
    H, W, Cc, K = 50, 84, 256, 512
    feat = torch.randn(2, Cc, H, W, device="cuda")
    rois = synth.fpn_rois(per_image=K // 2)
    rois = _t(rois[:K])
    fbytes = feat.numel() * 4
    alg = 4 * K * Cc * 49 + fbytes + 20 * K
    us = dev_time_us(lambda: C.roi_pool_forward(feat, rois, 1.0 / 16, 7, 7), iters)
    out.append(_entry("roi_pool_fwd 512 x 256 x 7x7 on 2x256x50x84", us, alg + 4 * K * Cc * 49))      # + argmax
    o, am = C.roi_pool_forward(feat, rois, 1.0 / 16, 7, 7)
    g = torch.randn_like(o)
    us = dev_time_us(lambda: C.roi_pool_backward(g, feat, rois, am, 1.0 / 16, 7, 7, 2, Cc, H, W), iters)
    out.append(_entry("roi_pool_bwd 512 x 256 x 7x7 on 2x256x50x84", us, alg + 4 * K * Cc * 49))
    D, P = 10, 7                                                         # C = D * P * P = 490 position-sensitive maps
    pfeat = torch.randn(2, D * P * P, H, W, device="cuda")
    trans = torch.randn(K, 2, P, P, device="cuda") * 0.1
    po = torch.empty(K, D, P, P, device="cuda")
    cnt = torch.empty_like(po)
    palg = 2 * 4 * K * D * P * P + pfeat.numel() * 4 + 20 * K + trans.numel() * 4
    us = dev_time_us(lambda: C.deform_psroi_pooling_forward(pfeat, rois, trans, po, cnt, 0, 1.0 / 16, D, P, P, P, 4, 0.1), iters)
    out.append(_entry("deform_psroi_fwd 512 rois, 10 x 7x7, 4 samples/part", us, palg))
    gi, gt = torch.zeros_like(pfeat), torch.zeros_like(trans)
    go = torch.randn_like(po)
    us = dev_time_us(lambda: C.deform_psroi_pooling_backward(go, pfeat, rois, trans, cnt, gi, gt, 0, 1.0 / 16, D, P, P, P, 4, 0.1), iters)
    out.append(_entry("deform_psroi_bwd (atomic accumulate into pre-zeroed gradients)", us, palg))
    return out


def bench_targets(C, iters):
    """IoU + Matcher and the balanced sampler at the RPN's shape (2 images x 268,569 anchors of the 800 x 1344 pyramid,
    <= 20 ground-truth boxes) and at the box head's (2 x ~2,020 proposals); bytes: boxes + matched indices."""
    out = []
    rng = np.random.RandomState(9)
    anchors = []
    for stride in (4, 8, 16, 32, 64):        # the RPN's anchor grid of a padded 800 x 1344 image: 3 ratios per location
        hh, ww = -(-800 // stride), -(-1344 // stride)
        ys, xs = np.meshgrid(np.arange(hh) * stride, np.arange(ww) * stride, indexing="ij")
        for r in (0.5, 1.0, 2.0):
            w_, h_ = 8 * stride / np.sqrt(r), 8 * stride * np.sqrt(r)
            anchors.append(np.stack([xs - w_ / 2, ys - h_ / 2, xs + w_ / 2, ys + h_ / 2], -1).reshape(hh * ww, 1, 4))
    per_level = [np.concatenate(anchors[3 * l:3 * l + 3], 1).reshape(-1, 4) for l in range(5)]
    rpn_anchors = np.concatenate(per_level).astype(np.float32)
    assert rpn_anchors.shape[0] == 268569
    for tag, K, M, B, lq in (("rpn 2x268569 anchors", 268569, 20, 256, True), ("box head 2x2020 proposals", 2020, 20, 512, False)):
        x1 = rng.uniform(0, 1200, (K,)); y1 = rng.uniform(0, 700, (K,))
        s_ = np.exp(rng.uniform(np.log(16), np.log(512), (K,)))
        boxes = torch.from_numpy(np.stack([x1, y1, x1 + s_, y1 + s_ * rng.uniform(0.5, 2, (K,))], 1).astype(np.float32)).cuda()
        if K == 268569:
            boxes = torch.from_numpy(rpn_anchors).cuda()
        gx = rng.uniform(0, 1000, (2, M)); gy = rng.uniform(0, 600, (2, M)); gs = rng.uniform(30, 400, (2, M))
        gt = torch.from_numpy(np.stack([gx, gy, gx + gs, gy + gs], 2).astype(np.float32)).cuda()
        valid = torch.ones(2, M, dtype=torch.bool, device="cuda")
        valid[1, 12:] = False
        us = dev_time_us(lambda: C.match_boxes(gt, valid, boxes, 0.7, 0.3, lq), iters)
        out.append(_entry(f"match_boxes (IoU + Matcher, low-quality rule {'on' if lq else 'off'}) {tag}", us, 16 * K + 2 * 8 * K))
        matched = C.match_boxes(gt, valid, boxes, 0.7, 0.3, lq)
        labels = (matched >= 0).float() - (matched == -2).float()
        us = dev_time_us(lambda: C.sample_labels(labels, B, B // 2, with_list=not lq), iters)
        out.append(_entry(f"sample_labels B={B} {tag}", us, 2 * K * (4 + 2)))
    return out


def bench_nms(C, iters):
    out = []
    for n, uniform in [(819, False), (2000, False), (2000, True), (6000, False)]:
        b, s = synth.nms_boxes(n, seed=2, uniform=uniform)
        tb, ts = _t(b), _t(s)
        us = dev_time_us(lambda: C.nms(tb, ts, 0.7), iters)  # includes the count readback
        k = int(C.nms(tb, ts, 0.7).numel())
        out.append(_entry(f"nms n={n} {'uniform' if uniform else 'clustered'} (with .item())", us,
                          20 * n + 8 * k, {"kept": k, "pairs_per_us": round(n * (n - 1) / 2 / us, 1)}))
    segs = synth.rpn_nms_segments()
    boxes = _t(np.concatenate([b for b, _ in segs]))
    scores = _t(np.concatenate([s for _, s in segs]))
    offs = _t(np.cumsum([0] + [len(s) for _, s in segs]).astype(np.int32))
    tot = int(boxes.size(0))
    pairs = sum(len(s) * (len(s) - 1) // 2 for _, s in segs)
    for fused, label in ((0, "one launch"), (3, "one launch, scans last"), (2, "three launches")):
        tune("nms_fused", fused)
        us = dev_time_us(lambda: C.nms_batched(boxes, scores, offs, 2000, 0.7), iters)
        out.append(_entry("nms batched 10 RPN segments, %s (no sync)" % label, us, 20 * tot,
                          {"pairs_per_us": round(pairs / us, 1)}))
    tune("nms_fused", 0)
    # the detector's calls: every segment is top-k output, i.e. already in score order (modeling/rpn/inference.py) — the sort
    # workgroups detect that and skip the network (`nms_no_presorted=1`: always sort)
    so = [np.argsort(-s, kind="stable") for _, s in segs]
    boxes_s = _t(np.concatenate([b[o] for (b, _), o in zip(segs, so)]))
    scores_s = _t(np.concatenate([s[o] for (_, s), o in zip(segs, so)]))
    for off, label in ((0, "in score order"), (1, "in score order, network forced")):
        tune("nms_no_presorted", off)
        us = dev_time_us(lambda: C.nms_batched(boxes_s, scores_s, offs, 2000, 0.7), iters)
        out.append(_entry("nms batched 10 RPN segments %s, one launch (no sync)" % label, us, 20 * tot,
                          {"pairs_per_us": round(pairs / us, 1)}))
    tune("nms_no_presorted", 0)
    return out


def bench_frozen_bn(C, iters):
    """fused FrozenBN+ReLU(+residual) vs the PyTorch elementwise chain, res2-sized activation."""
    out = []
    x = torch.randn(2, 256, 200, 336, device="cuda")
    r = torch.randn_like(x)
    scale = torch.rand(256, device="cuda") + 0.5
    bias = torch.randn(256, device="cuda")
    nb = x.numel() * 4
    us = dev_time_us(lambda: C.frozen_bn_act_forward(x, scale, bias, None, True), iters)
    out.append(_entry("frozen_bn_relu fwd fused [2,256,200,336]", us, 2 * nb))
    us = dev_time_us(lambda: C.frozen_bn_act_forward(x, scale, bias, r, True), iters)
    out.append(_entry("frozen_bn_add_relu fwd fused", us, 3 * nb))
    y = C.frozen_bn_act_forward(x, scale, bias, r, True)
    us = dev_time_us(lambda: C.frozen_bn_act_backward(r, y, scale, True, True), iters)
    out.append(_entry("frozen_bn_add_relu bwd fused (2 outputs)", us, 4 * nb))
    s4, b4 = scale.view(1, -1, 1, 1), bias.view(1, -1, 1, 1)
    us = dev_time_us(lambda: torch.relu_(x * s4 + b4), iters)
    out.append(_entry("frozen_bn_relu fwd torch chain (3 kernels)", us, 2 * nb))
    return out


def bench_focal(C, iters):
    out = []
    for R in (200000, 403200):
        logits, targets = synth.focal_inputs(R, 80)
        tl, tt = _t(logits), _t(targets)
        d = torch.rand_like(tl)
        one = torch.ones((), device="cuda")
        us = dev_time_us(lambda: C.sigmoid_focalloss_forward(tl, tt, 80, 2.0, 0.25), iters)
        out.append(_entry(f"focal_fwd R={R}", us, 2 * 4 * R * 80 + 4 * R))
        us = dev_time_us(lambda: C.sigmoid_focalloss_backward(tl, tt, d, 80, 2.0, 0.25), iters)
        out.append(_entry(f"focal_bwd R={R}", us, 3 * 4 * R * 80 + 4 * R))
        us = dev_time_us(lambda: C.sigmoid_focalloss_forward_sum(tl, tt, 80, 2.0, 0.25), iters)
        out.append(_entry(f"focal_fwd_sum R={R}", us, 4 * R * 80 + 4 * R))
        us = dev_time_us(lambda: C.sigmoid_focalloss_backward_scalar(tl, tt, one, 80, 2.0, 0.25), iters)
        out.append(_entry(f"focal_bwd_scalar R={R}", us, 2 * 4 * R * 80 + 4 * R))
    return out


def bench_dcn_fused(C, iters):
    """deformable conv forward at the cfg-5 layer shapes, fp16 / bf16: fused implicit GEMM on MFMA (columns never
    written) vs im2col kernel + library GEMM; TFLOP/s against the 2.5 PF dense matrix peak and the bytes the
    unfused path moves through HBM for `columns` (write + read)."""
    out = []
    for (Cc, H, W) in [(128, 100, 168), (256, 50, 84), (512, 25, 42)]:
        for dt in (torch.float16, torch.bfloat16):
            x = torch.randn(2, Cc, H, W, device="cuda").to(dt)
            off = (torch.randn(2, 18, H, W, device="cuda") * 2).to(dt)
            msk = torch.rand(2, 9, H, W, device="cuda").to(dt)
            w = (torch.randn(Cc, Cc, 3, 3, device="cuda") / (3 * Cc ** 0.5)).to(dt)
            y = torch.empty(2, Cc, H, W, device="cuda", dtype=dt)
            bufs = [torch.empty(0, device="cuda", dtype=dt), torch.empty(0, device="cuda", dtype=dt)]
            flops = 2.0 * Cc * Cc * 9 * 2 * H * W
            col_bytes = 2 * Cc * 9 * 2 * H * W * 2

            def run():
                C.modulated_deform_conv_forward(x, w, None, bufs[0], off, msk, y, bufs[1], 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, False)

            res = {}
            for tag, env in (("fused MFMA", 1), ("im2col + GEMM", 2)):
                tune("dcn_fused", env)
                us = dev_time_us(run, iters)
                res[tag] = y.float().clone()
                e = _entry(f"dcn_fwd {tag} C={Cc} {H}x{W} {str(dt)[6:]}", us, 2 * (x.numel() + off.numel() + msk.numel() + y.numel()),
                           {"TFLOPs": round(flops / us / 1e6, 1), "frac_of_2.5PF": round(flops / us / 1e6 / 2500.0, 4),
                            "columns_bytes_unfused": col_bytes})
                out.append(e)
            tune("dcn_fused", 0)
            out[-1]["max_abs_diff_fused_vs_unfused"] = float((res["fused MFMA"] - res["im2col + GEMM"]).abs().max())
    return out


def bench_dcn(C, iters, experimental=False):
    out = []
    for (Cc, H, W, smooth) in [(128, 100, 168, False), (256, 50, 84, False), (512, 25, 42, False), (128, 100, 168, True)]:
        for dt, e in ((torch.float32, 4), (torch.float16, 2)):
            x = torch.randn(2, Cc, H, W, device="cuda").to(dt)
            if smooth:  # spatially smooth offsets (a conv-predicted field): low-resolution noise upsampled, ~2 px
                lo = torch.randn(2, 18, H // 8 + 2, W // 8 + 2, device="cuda") * 2
                off = torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False).to(dt)
            else:       # i.i.d. N(0, 2 px) offsets (worst case for locality)
                off = (torch.randn(2, 18, H, W, device="cuda") * 2).to(dt)
            geo = (3, 3, 1, 1, 1, 1, 1, 1, 1)
            ncol = 2 * H * W
            alg = e * (x.numel() + off.numel() + Cc * 9 * ncol)
            us = dev_time_us(lambda: C.deformable_im2col(x, off, None, *geo), iters)
            tagx = " smooth-offsets" if smooth else ""
            out.append(_entry(f"dcn_im2col C={Cc} {H}x{W} {str(dt)[6:]}{tagx}", us, alg))
            col = torch.randn(Cc * 9, ncol, device="cuda").to(dt)
            gim = torch.zeros_like(x)
            us = dev_time_us(lambda: C.deformable_col2im(col, off, None, gim, *geo), iters)
            out.append(_entry(f"dcn_col2im default C={Cc} {H}x{W} {str(dt)[6:]}{tagx}", us, alg))
            tune("dcn_col2im", 1)
            us = dev_time_us(lambda: C.deformable_col2im(col, off, None, gim, *geo), max(3, iters // 5))
            out.append(_entry(f"dcn_col2im gather (index+sort+gather, XCD-contiguous) C={Cc} {H}x{W} {str(dt)[6:]}{tagx}", us, alg))
            tune("dcn_gather_xcd", 1)
            us = dev_time_us(lambda: C.deformable_col2im(col, off, None, gim, *geo), max(3, iters // 5))
            tune("dcn_gather_xcd", 0)
            out.append(_entry(f"dcn_col2im gather (plain block order) C={Cc} {H}x{W} {str(dt)[6:]}{tagx}", us, alg))
            tune("dcn_col2im", 2)
            us = dev_time_us(lambda: C.deformable_col2im(col, off, None, gim, *geo), max(3, iters // 5))
            tune("dcn_col2im", 0)
            out.append(_entry(f"dcn_col2im scatter (LDS atomics) C={Cc} {H}x{W} {str(dt)[6:]}{tagx}", us, alg))
            if True:
                ref_g = torch.zeros_like(x)
                C.deformable_col2im(col, off, None, ref_g, *geo)
                tune("dcn_col2im", 3)
                us = dev_time_us(lambda: C.deformable_col2im(col, off, None, gim, *geo), max(3, iters // 5))
                got = torch.zeros_like(x)
                C.deformable_col2im(col, off, None, got, *geo)
                tune("dcn_col2im", 0)
                err = float((got.float() - ref_g.float()).abs().max())
                out.append(_entry(f"dcn_col2im ELL index C={Cc} {H}x{W} {str(dt)[6:]}{tagx}", us, alg,
                                  {"max_abs_diff_vs_default": err}))
            goff = torch.empty_like(off)
            us = dev_time_us(lambda: C.deformable_col2im_coord(col, x, off, None, goff, None, *geo), iters)
            out.append(_entry(f"dcn_col2im_coord C={Cc} {H}x{W} {str(dt)[6:]}{tagx}", us, alg))
    return out


def bench_dcn_block(C, iters):
    """One modulated deformable 3x3 block of cfg-5 (R-101 + DCN, fp16) at its three layer shapes, forward and backward
    through the `_C` entry points the layers call: channels-last pipeline (default) vs the reference-layout kernels
    (tuning dcn_nhwc = 2), fused MFMA forward where it applies; plus the pipeline's own kernels one by one."""
    out = []
    for (Cc, H, W) in [(128, 100, 168), (256, 50, 84), (512, 25, 42)]:
        for dt in (torch.float16, torch.float32):
            e = 2 if dt == torch.float16 else 4
            x = torch.randn(2, Cc, H, W, device="cuda").to(dt)
            off = (torch.randn(2, 18, H, W, device="cuda") * 2).to(dt)
            msk = torch.rand(2, 9, H, W, device="cuda").to(dt)
            w = (torch.randn(Cc, Cc, 3, 3, device="cuda") / (3 * Cc ** 0.5)).to(dt)
            bias = torch.zeros(Cc, device="cuda", dtype=dt)
            y = torch.empty(2, Cc, H, W, device="cuda", dtype=dt)
            go = torch.randn(2, Cc, H, W, device="cuda").to(dt)
            empty = torch.empty(0, device="cuda", dtype=dt)
            geo = (3, 3, 1, 1, 1, 1, 1, 1, 1, 1)
            flops = 2.0 * Cc * Cc * 9 * 2 * H * W
            io_bytes = e * (x.numel() + off.numel() + msk.numel() + y.numel())
            tag = f"C={Cc} {H}x{W} {str(dt)[6:]}"

            def fwd():
                C.modulated_deform_conv_forward(x, w, bias, empty, off, msk, y, empty, *geo, True)

            gi, gw, gb = torch.zeros_like(x), torch.zeros_like(w), torch.zeros_like(bias)
            goff, gm = torch.zeros_like(off), torch.zeros_like(msk)

            def bwd():
                C.modulated_deform_conv_backward(x, w, bias, empty, off, msk, empty, gi, gw, gb, goff, gm, go, *geo, True)

            res = {}
            for name, key, val in (("channels-last pipeline", "dcn_nhwc", 0), ("reference-layout kernels", "dcn_nhwc", 2)):
                tune("dcn_fused", 2)
                tune(key, val)
                us = dev_time_us(fwd, iters)
                out.append(_entry(f"dcn_block fwd [{name}] {tag}", us, io_bytes, {"TFLOPs": round(flops / us / 1e6, 1)}))
                res["fwd " + name] = y.float().clone()
                us = dev_time_us(bwd, max(3, iters // 2))
                out.append(_entry(f"dcn_block bwd [{name}] {tag}", us, 2 * io_bytes, {"TFLOPs": round(2 * flops / us / 1e6, 1)}))
                gi.zero_(); gw.zero_(); gb.zero_()
                bwd()
                res["bwd " + name] = (gi.float().clone(), gw.float().clone(), goff.float().clone(), gm.float().clone())
                gi.zero_(); gw.zero_(); gb.zero_()
                tune(key, 0)
                tune("dcn_fused", 0)
            if dt == torch.float16:
                tune("dcn_fused", 1)
                us = dev_time_us(fwd, iters)
                tune("dcn_fused", 0)
                out.append(_entry(f"dcn_block fwd [fused MFMA] {tag}", us, io_bytes, {"TFLOPs": round(flops / us / 1e6, 1)}))
            a, b = res["bwd channels-last pipeline"], res["bwd reference-layout kernels"]
            out[-1 if dt != torch.float16 else -2]["max_rel_diff_vs_reference_layout"] = [
                round(float((p - q).abs().max() / q.abs().max().clamp_min(1e-6)), 5) for p, q in zip(a, b)]
            # the pipeline's kernels, one by one
            xT = C._to_nhwc(x)
            us = dev_time_us(lambda: C._to_nhwc(x), iters)
            out.append(_entry(f"dcn nchw->nhwc {tag}", us, 2 * e * x.numel()))
            g9 = (3, 3, 1, 1, 1, 1, 1, 1, 1)
            colT = C._im2col_nhwc(xT, off, msk, 2, Cc, H, W, g9)
            us = dev_time_us(lambda: C._im2col_nhwc(xT, off, msk, 2, Cc, H, W, g9), iters)
            out.append(_entry(f"dcn im2col_nhwc {tag}", us, e * (x.numel() + off.numel() + msk.numel() + colT.numel())))
            us = dev_time_us(lambda: C._coord_nhwc(colT, xT, off, msk, goff, gm, 2, Cc, H, W, g9), iters)
            out.append(_entry(f"dcn coord_nhwc {tag}", us, e * (x.numel() + 2 * off.numel() + 2 * msk.numel() + colT.numel())))
            gT = C._to_nhwc(go)
            us = dev_time_us(lambda: C._transposed_sample(gT, off, msk, 2, Cc, H, W, Cc, g9), iters)
            out.append(_entry(f"dcn transposed_sample (index build + gather) {tag}", us, e * (go.numel() + off.numel() + msk.numel() + colT.numel())))
    return out


def copy_ceiling(iters):
    a = torch.empty(256 * 1024 * 1024 // 4, device="cuda")
    b = torch.empty_like(a)
    us = dev_time_us(lambda: b.copy_(a), iters)
    return _entry("hbm copy 256MiB (torch copy_, ceiling reference)", us, 2 * a.numel() * 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default="")
    ap.add_argument("--sets", default="", help="roi_sets: comma list of ROI-set names (default all)")
    ap.add_argument("--heads", default="", help="roi_sets: box,mask (default both)")
    ap.add_argument("--dir", default="", help="roi_sets: fwd,bwd (default both)")
    ap.add_argument("--model-rois", default=None, help="npz of tools/dump_model_rois.py (default: tests/golden/model_rois.npz)")
    ap.add_argument("--layout", default="nchw", help="roi_sets: nchw | nhwc | both")
    ap.add_argument("--images", type=int, default=2, help="roi_sets: images per batch (4 = the > L3 variant: 365.6 MB of maps)")
    ap.add_argument("--tune", default="", help="library tuning switches for the whole run, key=value,key=value")
    ap.add_argument("--sweep", default="", help="roi_sets: cartesian sweep over tuning switches, 'key=v1|v2,key2=v1|v2'")
    ap.add_argument("--experimental", action="store_true",
                    help="also time the opt-in, not-yet-measured kernel variants (DESIGN.md section 7)")
    args = ap.parse_args()
    from maskrcnn_benchmark import _C as C

    only = set(filter(None, args.only.split(",")))
    res = []
    t0 = time.time()
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        tune(k, int(v))
    if not only or "copy" in only:
        res.append(copy_ceiling(args.iters))
    if not only or "roi_align" in only:
        res += bench_roi_align(C, args.iters, experimental=args.experimental)
    if "roi_align_fpn" in only:
        res += bench_roi_align(C, args.iters, fused_only=True)
    if not only or "roi_sets" in only:
        import itertools
        axes = [(kv.split("=")[0], [int(v) for v in kv.split("=")[1].split("|")]) for kv in filter(None, args.sweep.split(","))]
        for combo in itertools.product(*[vals for _, vals in axes]):
            for (k, _), v in zip(axes, combo):
                tune(k, v)
            for lay in (("nchw", "nhwc") if args.layout == "both" else (args.layout,)):
                res += bench_roi_sets(C, args.iters, args.model_rois, set(filter(None, args.sets.split(","))),
                                      set(filter(None, args.heads.split(","))), tuple(filter(None, args.dir.split(","))) or ("fwd", "bwd"),
                                      images=args.images, tag=" ".join("%s=%d" % (k, v) for (k, _), v in zip(axes, combo)), layout=lay)
        for k, _ in axes:
            tune(k, 0)
    if "roi_align_fwd" in only:
        res += bench_roi_align(C, args.iters, which=("fwd",))
    if not only or "pool" in only:
        res += bench_roi_pool_psroi(C, args.iters)
    if not only or "nms" in only:
        res += bench_nms(C, args.iters)
    if not only or "targets" in only:
        res += bench_targets(C, args.iters)
    if not only or "frozen_bn" in only:
        res += bench_frozen_bn(C, args.iters)
    if not only or "focal" in only:
        res += bench_focal(C, args.iters)
    if not only or "dcn" in only:
        res += bench_dcn(C, args.iters, experimental=args.experimental)
    if not only or "dcn_fused" in only:
        res += bench_dcn_fused(C, args.iters)
    if not only or "dcn_block" in only:
        res += bench_dcn_block(C, args.iters)
    for r in res:
        print("%-70s %10.2f us  %9.1f GB/s  (%.1f%% of 8 TB/s) %s" % (
            r["op"], r["us"], r["gbs"], 100 * r["frac_of_8TBs"],
            {k: v for k, v in r.items() if k not in ("op", "us", "gbs", "frac_of_8TBs", "alg_bytes")} or ""))
    print("total %.1f s" % (time.time() - t0))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
