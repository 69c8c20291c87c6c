#!/usr/bin/env python
"""Capture the ROI sets the detector itself hands to the box-head / mask-head ROIAlign in a training step
(random-init weights, the bench.py workload) so tools/opbench.py can time the kernels on them:

    python tools/dump_model_rois.py [--steps 6] [--out gpurun_out/model_rois.npz]

Writes box_rois [1024,5], box_levels, mask_rois [256,5], mask_levels (the last step's sets).  The committed copy
lives in tests/golden/model_rois.npz (20 KB); VERDICT r02 item 1a."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--config", default="e2e_mask_rcnn_R_50_FPN_1x.yaml")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "model_rois.npz"))
    args = ap.parse_args()
    import numpy as np
    import torch

    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches

    device = torch.device("cuda", 0)
    base = load_cfg(args.config, []).SOLVER.BASE_LR
    cfg = load_cfg(args.config, ["SOLVER.BASE_LR", base * 2 / 16.0, "SOLVER.IMS_PER_BATCH", 2])
    torch.manual_seed(1234)   # bench.py's seed for rank 0
    model, optimizer, scheduler, step = build_training(cfg, device)
    batches = make_device_batches(cfg, device, images_per_gpu=2, num_batches=2, seed=0)
    captured = {}
    orig = _C.roi_align_fpn_backward

    def spy(grad, rois, levels, shapes, scales, ph, pw, sr, **kw):
        captured["box" if ph == 7 else "mask"] = (rois.detach().clone(), levels.detach().clone())
        return orig(grad, rois, levels, shapes, scales, ph, pw, sr, **kw)

    _C.roi_align_fpn_backward = spy
    for i in range(args.steps):
        step(*batches[i % len(batches)])
    torch.cuda.synchronize()
    _C.roi_align_fpn_backward = orig
    out = {}
    for k, (r, l) in captured.items():
        out[k + "_rois"] = r.cpu().numpy()
        out[k + "_levels"] = l.cpu().numpy()
        print(k, tuple(r.shape), "ROIs per level:", np.bincount(out[k + "_levels"], minlength=4).tolist())
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
