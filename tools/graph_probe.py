#!/usr/bin/env python
"""Probe: does one training iteration capture into a HIP graph (torch.cuda.CUDAGraph), and what does replaying buy?
    python tools/graph_probe.py [--dtype float32|bfloat16] [--config ...]
Eager ms/step vs graph-replay ms/step of forward + backward (+ fused SGD) on one fixed device-resident batch."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--config", default="e2e_mask_rcnn_R_50_FPN_1x.yaml")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--stage", default="full", help="backbone | rpn | forward | fwdbwd | full: how much of the step is captured")
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[])
    args = ap.parse_args()
    import bench
    bench.setup_miopen_db(None)
    import torch
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches
    dev = torch.device("cuda", 0)
    base = load_cfg(args.config, []).SOLVER.BASE_LR
    cfg = load_cfg(args.config, list(args.opts) + ["DTYPE", args.dtype, "SOLVER.BASE_LR", base / 8, "SOLVER.IMS_PER_BATCH", 2])
    torch.manual_seed(1234)
    model, optimizer, scheduler, step = build_training(cfg, dev)
    (images, targets), = make_device_batches(cfg, dev, images_per_gpu=2, num_batches=1, seed=0)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, (t1 - t0) / n * 1e3

    for _ in range(8):
        step(images, targets)
    ms, host = timed(lambda: step(images, targets), args.steps)
    print("eager: %.2f ms/step (host enqueue %.2f)" % (ms, host), flush=True)

    # host-to-device copies inside one eager step (each would be replayed from a stale host buffer by a graph)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(images, targets)
        torch.cuda.synchronize()
    copies = [e for e in prof.events() if "memcpy" in e.name.lower() or "Memcpy" in e.name]
    import collections
    print("memcpy events in one eager step:", dict(collections.Counter(e.name for e in copies)), flush=True)

    amp = step.amp_dtype

    def fwd_bwd():
        with torch.autocast(device_type="cuda", dtype=amp, enabled=amp is not None):
            if args.stage == "backbone":
                feats = model.backbone(images.tensors)
                loss_dict = {"sum": sum(f.float().mean() for f in feats)}
            elif args.stage == "rpn":
                feats = model.backbone(images.tensors)
                _, loss_dict = model.rpn(images, feats, targets)
            elif args.stage in ("dummy", "stack", "pad"):
                from maskrcnn_benchmark.modeling.roi_heads.box_head.loss import stack_proposals
                from maskrcnn_benchmark.modeling.rpn.loss import pad_targets
                feats = model.backbone(images.tensors)
                proposals, loss_dict = model.rpn(images, feats, targets)
                with torch.no_grad():
                    if args.stage == "dummy":
                        z = torch.zeros(1000, device="cuda")
                        for _ in range(60):
                            z = z + 1.0
                        loss_dict = dict(loss_dict, chk=z.sum() + sum(p.bbox.sum() for p in proposals))
                    elif args.stage == "stack":
                        boxes, valid = stack_proposals(proposals)
                        loss_dict = dict(loss_dict, chk=boxes.sum() + valid.float().sum())
                    else:
                        gt, row_valid, extra = pad_targets(targets, "cuda", ("labels",))
                        loss_dict = dict(loss_dict, chk=gt.sum() + extra["labels"].float().sum())
            elif args.stage in ("s1", "s2", "s3"):
                from maskrcnn_benchmark.modeling.roi_heads.box_head.loss import stack_proposals
                from maskrcnn_benchmark.modeling.rpn.loss import match_batched, pad_targets
                feats = model.backbone(images.tensors)
                proposals, loss_dict = model.rpn(images, feats, targets)
                ev = model.roi_heads.box.loss_evaluator
                with torch.no_grad():
                    boxes, valid = stack_proposals(proposals)
                    gt, row_valid, extra = pad_targets(targets, boxes.device, ("labels",))
                    loss_dict = dict(loss_dict, chk=boxes.sum() + gt.sum() + valid.float().sum() + extra["labels"].float().sum())
                    if args.stage in ("s2", "s3"):
                        matched = match_batched(ev.proposal_matcher, gt, row_valid, boxes)
                        loss_dict["matched"] = matched.float().sum()
                    if args.stage == "s3":
                        labels, _, _ = ev.prepare_targets(boxes, valid, targets)
                        idx, slot_valid = ev.fg_bg_sampler.sample_fixed(labels)
                        loss_dict["idx"] = idx.float().sum() + slot_valid.float().sum()
            elif args.stage in ("subsample", "pool", "box"):
                feats = model.backbone(images.tensors)
                proposals, loss_dict = model.rpn(images, feats, targets)
                box = model.roi_heads.box
                if args.stage == "box":
                    _, _, lb = box(feats, proposals, targets)
                    loss_dict = dict(loss_dict, **lb)
                else:
                    with torch.no_grad():
                        sampled = box.loss_evaluator.subsample(proposals, targets)
                    loss_dict = dict(loss_dict, lab=sum(p.get_field("labels").float().sum() for p in sampled))
                    if args.stage == "pool":
                        x = box.feature_extractor.pooler(feats, sampled)
                        loss_dict["pooled"] = x.float().mean()
            else:
                loss_dict = model(images, targets)
        if args.stage in ("backbone", "rpn", "forward", "subsample", "pool", "box", "s1", "s2", "s3", "dummy", "stack", "pad"):
            return loss_dict
        losses = sum(loss_dict.values())
        optimizer.zero_grad(set_to_none=True)
        losses.backward()
        if args.stage == "full":
            optimizer.step()
        return loss_dict

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = fwd_bwd()
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        print("CAPTURE FAILED: %r" % (e,))
        return
    for i in range(6):
        g.replay()
        torch.cuda.synchronize()
        print("replay %d losses:" % i, {k: round(float(v.detach()), 4) for k, v in out.items()}, flush=True)
    ms, host = timed(g.replay, args.steps)
    print("graph replay: %.2f ms/step (host enqueue %.2f)" % (ms, host), flush=True)


if __name__ == "__main__":
    main()
