#!/usr/bin/env python
"""Where does the HOST time of a training step go?  (A host-bound configuration — cfg-5, R-101 + DCN under fp16 —
spends its whole step enqueueing.)  Wall-clock split forward / backward / optimizer without device syncs in between,
per-class timers around every custom autograd Function (apply and backward, which runs on the autograd thread) and a
cProfile of the forward pass.

    python tools/host_profile.py [--config FILE] [--dtype float16] [--steps 10] [KEY VALUE ...]"""
import argparse
import cProfile
import collections
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd")]

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="e2e_mask_rcnn_R_50_FPN_1x.yaml")
    ap.add_argument("--dtype", default=None)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("opts", nargs=argparse.REMAINDER)
    args = ap.parse_args()
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches
    opts = list(args.opts) + (["DTYPE", args.dtype] if args.dtype else [])
    cfg = load_cfg(args.config, opts)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model, optimizer, scheduler, step = build_training(cfg, dev)
    batches = make_device_batches(cfg, dev, images_per_gpu=2, num_batches=2)
    for i in range(args.warmup):
        step(*batches[i % 2])
    torch.cuda.synchronize()

    # ---- per-Function timers (apply on the main thread, backward on the autograd thread)
    acc = collections.defaultdict(lambda: [0, 0.0])
    import maskrcnn_benchmark  # noqa: F401
    seen = set()

    def wrap(cls):
        if cls in seen or not cls.__module__.startswith("maskrcnn_benchmark"):
            return
        seen.add(cls)
        for name in ("forward", "backward"):
            fn = cls.__dict__.get(name)
            if fn is None:
                continue
            raw = fn.__func__ if isinstance(fn, staticmethod) else fn

            def timed(*a, _raw=raw, _key="%s.%s" % (cls.__name__, name), **k):
                t0 = time.perf_counter()
                try:
                    return _raw(*a, **k)
                finally:
                    e = acc[_key]
                    e[0] += 1
                    e[1] += time.perf_counter() - t0
            setattr(cls, name, staticmethod(timed))

    def all_subclasses(c):
        for s in c.__subclasses__():
            yield s
            yield from all_subclasses(s)
    for c in list(all_subclasses(torch.autograd.Function)):
        wrap(c)

    # ---- wall-clock split of the step (host-bound: no syncs inside)
    amp = step.amp_dtype
    t_f = t_b = t_o = 0.0
    torch.cuda.synchronize()
    t_all0 = time.perf_counter()
    for i in range(args.steps):
        images, targets = batches[i % 2]
        t0 = time.perf_counter()
        if amp is None:
            loss_dict = model(images, targets)
        else:
            with torch.autocast("cuda", dtype=amp):
                loss_dict = model(images, targets)
        losses = sum(loss_dict.values())
        t1 = time.perf_counter()
        optimizer.zero_grad(set_to_none=True)
        if step.scaler is not None:
            step.scaler.scale(losses).backward()
            t2 = time.perf_counter()
            step.scaler.step(optimizer)
            step.scaler.update()
        else:
            losses.backward()
            t2 = time.perf_counter()
            optimizer.step()
        t3 = time.perf_counter()
        t_f += t1 - t0
        t_b += t2 - t1
        t_o += t3 - t2
    host = time.perf_counter() - t_all0
    torch.cuda.synchronize()
    total = time.perf_counter() - t_all0
    K = args.steps
    print("per step: total %.2f ms | host enqueue %.2f ms = forward %.2f + backward %.2f + optimizer %.2f"
          % (1e3 * total / K, 1e3 * host / K, 1e3 * t_f / K, 1e3 * t_b / K, 1e3 * t_o / K))
    print("custom autograd Functions (host ms per step, calls per step):")
    for key, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("  %-44s %7.3f ms  %5.1f calls  %6.1f us/call" % (key, 1e3 * t / K, n / K, 1e6 * t / max(n, 1)))

    # ---- cProfile of the forward pass
    pr = cProfile.Profile()
    for i in range(args.steps):
        images, targets = batches[i % 2]
        pr.enable()
        if amp is None:
            loss_dict = model(images, targets)
        else:
            with torch.autocast("cuda", dtype=amp):
                loss_dict = model(images, targets)
        pr.disable()
        sum(loss_dict.values()).backward()
        optimizer.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime")
    print("forward pass, cProfile (all %d steps), top by own time:" % K)
    st.print_stats(45)


if __name__ == "__main__":
    main()
