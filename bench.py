#!/usr/bin/env python
"""bench.py — the driver's measurement contract.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config FILE] [--dtype float32|bfloat16|float16]

A "step" is one training iteration (forward + backward + gradient all-reduce + SGD update) of
`e2e_mask_rcnn_R_50_FPN_1x` (BASELINE.json's metric config) on one batch of 2 synthetic
1333x800 COCO-shaped images per GPU (padded to 800x1344), random-init weights, batches resident in
HBM before the timed region.  N > 1: one rank per GPU over RCCL, launched by the driver with torch.distributed.run —
or by this script itself: `python bench.py --gpus N` without WORLD_SIZE in the environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (the reference's
`python -m torch.distributed.launch --nproc_per_node=$NGPUS tools/train_net.py`, README.md:147-163) and refuses to print
a line whose RCCL world size differs from --gpus.  Per-GPU work is fixed (weak scaling), value = total images /
max-over-ranks time.

Besides the contract line, rank 0 reports
  * "roofline": the dominant hand-written kernel family of the step among the SURVEY §8(a) operators (ROIAlign over the
    pyramid, focal loss, deformable conv), timed live with HIP events on its launch stream in a post-pass of
    KERNEL_SAMPLES more training steps that FOLLOWS the timed region (the event records drain the queue, so they stay out
    of `value`); achieved = algorithmic bytes per launch / mean launch time, vs the 8 TB/s HBM peak;
    "roofline_all": the same rule over every hand-written family (the fused FrozenBN streams included);
  * "kernels": the same figures for every hand-written entry point seen in the post-pass;
  * "cpu_baseline": the reference's own CPU ROIAlign kernel (oracle/_ref, built from the reference
    sources) — or the C restatement if that library is absent — timed on this host on a bounded
    sample of the box-head ROIAlign workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
RANK_SECONDS = []      # timed_steps: every rank's own wall time of the timed region (rank order)
KERNEL_SAMPLES = 50    # event pairs per hand-written entry point in the kernel-timing post-pass (>= 50 launches)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="e2e_mask_rcnn_R_50_FPN_1x.yaml")
    ap.add_argument("--dtype", default=None, help="override cfg.DTYPE (float32 | bfloat16 | float16)")
    ap.add_argument("--images-per-gpu", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-timing-steps", type=int, default=KERNEL_SAMPLES,
                    help="steps of the kernel-timing post-pass that follows the (untimed-kernel) timed region")
    ap.add_argument("--layout", default="auto", choices=["auto", "nchw", "backbone", "all"],
                    help="activation layout (engine/bench_step.py::choose_layout): backbone + FPN on channels-last (NHWC) "
                         "activations (\"backbone\"), the RPN / ROI heads as well (\"all\"), or NCHW everywhere")
    ap.add_argument("--channels-last", action="store_true", help="= --layout backbone")
    ap.add_argument("--channels-last-heads", action="store_true", help="= --layout all")
    ap.add_argument("--force-ddp", action="store_true",
                    help="N = 1 only: wrap the model in DDP over a 1-rank RCCL group so the overlapped-SGD hook "
                         "(bucket all-reduce -> update in the completion callback on a side stream) is what runs")
    ap.add_argument("--bucket-mb", type=int, default=None, help="DDP gradient bucket size (default: engine/ddp_step.py)")
    ap.add_argument("--allow-ddp-fallback", action="store_true",
                    help="if the overlapped hook raises in the first distributed step, retry with stock DDP "
                         "(reported in \"ddp\"); default: fail loudly")
    ap.add_argument("--miopen-search", action="store_true",
                    help="torch.backends.cudnn.benchmark = True: MIOpen times every applicable conv solver per "
                         "shape in warm-up (minutes from a cold kernel cache on a fresh box — measured >5 min for "
                         "this model in fp32; default off = MIOpen immediate mode, or the shipped find-db when "
                         "maskrcnn-benchmark_amd/miopen_db/ exists)")
    ap.add_argument("--no-miopen-search", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--export-miopen-db", default=None, metavar="DIR",
                    help="write MIOpen's find-db + kernel cache of this run to DIR (see setup_miopen_db)")
    ap.add_argument("--no-fixed-quota-line", action="store_true",
                    help="skip the 20 extra steps that time the step with the mask head's fixed quota of slots (mask_slots.fixed_quota_ms_per_step)")
    ap.add_argument("--hip-graph", action="store_true",
                    help="N = 1: replay the training iteration from a HIP graph (engine/graph_step.py: one capture per input "
                         "signature, learning rate / sampler seeds / loss scale on the device).  Pays where the step is "
                         "host-bound (R-101 + DCN under fp16); reported as \"hip_graph\": true in the line")
    ap.add_argument("--eval", action="store_true",
                    help="time the INFERENCE forward instead of the training step (eval mode, no_grad, test-size images, the "
                         "PostProcessors included): a context line next to the reference's s/im column (MODEL_ZOO.md:26), "
                         "not the headline metric")
    ap.add_argument("--stub-step", action="store_true",
                    help="test hook: CPU tensors, gloo, a stub step — exercises the launch / world-size / timed-region / "
                         "JSON logic without a GPU; the line is marked as a stub, not a measurement")
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[])
    return ap.parse_args()


def setup_miopen_db(export_dir=None):
    """MIOpen's per-user find-db (measured solver ranking per conv shape) and compiled-kernel cache
    live under $HOME and start EMPTY on a fresh box (the torch wheel ships no gfx950 find-db), so a
    cold process either searches for minutes or falls back to heuristics.  A tuning database
    produced once on an MI355X (`--miopen-search --export-miopen-db DIR`) can be shipped in-tree as
    maskrcnn-benchmark_amd/miopen_db/{db,cache}; it is copied to a scratch dir (MIOpen writes to it)
    and selected through MIOpen's own environment variables.  Must run before `import torch`."""
    import shutil
    import tempfile
    if export_dir:
        os.makedirs(os.path.join(export_dir, "db"), exist_ok=True)
        os.makedirs(os.path.join(export_dir, "cache"), exist_ok=True)
        seed = os.path.join(ROOT, "maskrcnn-benchmark_amd", "miopen_db")
        if os.path.isdir(os.path.join(seed, "db")) and not os.listdir(os.path.join(export_dir, "db")):
            # start from the shipped database: only problem keys it does not hold yet are searched
            shutil.copytree(seed, export_dir, dirs_exist_ok=True)
        os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(export_dir, "db")
        os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(export_dir, "cache")
        return "export:" + export_dir
    shipped = os.path.join(ROOT, "maskrcnn-benchmark_amd", "miopen_db")
    if "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(os.path.join(shipped, "db")):
        return None
    scratch = tempfile.mkdtemp(prefix="miopen_db_rank%s_" % os.environ.get("RANK", "0"))
    shutil.copytree(shipped, scratch, dirs_exist_ok=True)
    os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(scratch, "db")
    if os.path.isdir(os.path.join(scratch, "cache")):
        os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(scratch, "cache")
    return "shipped:" + shipped


# timer-name prefixes of the SURVEY §8(a) operators (a1/a2/a10 ROIAlign over the pyramid, a8 focal loss, a9 deformable conv)
PATH_FAMILIES = ("roi_align_fpn_", "focal_", "dcn_")


def algorithmic_bytes(name, feat_bytes):
    """SURVEY.md §8(d) per-launch ALGORITHMIC bytes of a hand-written entry point, from the shape in its timer name:
    ROIAlign fwd/bwd = pooled tensor + every pooled feature map once + rois; focal fwd(sum) = logits + targets, bwd =
    logits + targets + d_logits; FrozenBN = the activation read once + written once (+ residual / mask reads);
    deformable im2col / col2im / coord-gradient = input (or gradient) map + offsets (+ mask) + the column matrix;
    fused deformable forward = input + offsets (+ mask) + output + weights (no columns term)."""
    import re
    m = re.match(r"roi_align_fpn_(fwd|bwd)\[K=(\d+),C=(\d+),(\d+)x(\d+)\]", name)
    if m:
        K, C, ph, pw = (int(m.group(i)) for i in (2, 3, 4, 5))
        return 4 * K * C * ph * pw + feat_bytes + 20 * K
    m = re.match(r"focal_(fwd_sum|bwd_scalar)\[R=(\d+),C=(\d+)\]", name)
    if m:
        R, C = int(m.group(2)), int(m.group(3))
        return (4 * R * C + 4 * R) if m.group(1) == "fwd_sum" else (8 * R * C + 4 * R)
    m = re.match(r"frozen_bn_fwd\[n=(\d+),nc=\d+,e=(\d+),res=(\d)\]", name)
    if m:
        n, e, res = (int(g) for g in m.groups())
        return e * n * (2 + res)
    m = re.match(r"frozen_bn_bwd\[n=(\d+),nc=\d+,e=(\d+),res=(\d),relu=(\d)\]", name)
    if m:
        n, e, res, relu = (int(g) for g in m.groups())
        return e * n * (2 + res + relu)
    m = re.match(r"bias_act_bwd\[n=(\d+),C=\d+,e=(\d+),relu=(\d)\]", name)
    if m:      # gradient read (+ the saved output for the ReLU mask), masked gradient written when there is a ReLU; C sums out
        n, e, relu = (int(g) for g in m.groups())
        return e * n * (1 + 2 * relu)
    m = re.match(r"dcn_(im2col|col2im|col2im_coord|im2col_nhwc|col2im_nhwc)\[B=(\d+),C=(\d+),(\d+)x(\d+),k=(\d+),e=(\d+),m=(\d)\]", name)
    if m:
        B, C, H, W, k, e, msk = (int(g) for g in m.groups()[1:])
        pix = B * H * W        # 3x3 / stride 1 / pad 1 in every model config: Ho x Wo = H x W
        return e * (B * C * H * W + 2 * k * k * pix + msk * k * k * pix + C * k * k * pix)
    m = re.match(r"dcn_to_nhwc\[n=(\d+),e=(\d+)\]", name)
    if m:      # one read + one write of the tensor
        return 2 * int(m.group(1)) * int(m.group(2))
    m = re.match(r"dcn_coord_nhwc\[B=(\d+),C=(\d+),(\d+)x(\d+),k=(\d+),e=(\d+),m=(\d)\]", name)
    if m:      # column gradient + input + offsets (+ mask) read, offset (+ mask) gradients written
        B, C, H, W, k, e, msk = (int(g) for g in m.groups())
        pix = B * H * W
        return e * (C * k * k * pix + B * C * H * W + 2 * (2 * k * k * pix) + 2 * msk * k * k * pix)
    m = re.match(r"dcn_transposed_sample\[B=(\d+),Cout=(\d+),(\d+)x(\d+),k=(\d+),e=(\d+),m=(\d)\]", name)
    if m:      # output gradient + offsets (+ mask) read, S_T [B HW, k k Cout] written
        B, Co, H, W, k, e, msk = (int(g) for g in m.groups())
        pix = B * H * W
        return e * (B * Co * H * W + 2 * k * k * pix + msk * k * k * pix + Co * k * k * pix)
    m = re.match(r"dcn_fused_fwd\[B=(\d+),C=(\d+),(\d+)x(\d+),Cout=(\d+),k=(\d+),e=(\d+),m=(\d)\]", name)
    if m:
        B, C, H, W, Co, k, e, msk = (int(g) for g in m.groups())
        pix = B * H * W
        return e * (B * C * H * W + 2 * k * k * pix + msk * k * k * pix + B * Co * H * W + Co * C * k * k)
    return None


def rocprof_kernel_us(entry_name):
    """Average duration of the entry point's main kernel in the newest committed rocprofv3 kernel-trace summary
    (profiles/*kernel_times*.txt, written by tools/kernel_times.py from `rocprofv3 --kernel-trace` of the same
    workload) — printed beside the live HIP-event figure so that both can be compared."""
    import glob
    import re
    key = {"roi_align_fpn_bwd": "roi_align_bwd_ring_kernel", "roi_align_fpn_fwd": "roi_align_fwd_dma_kernel",
           "focal_fwd_sum": "focal_kernel", "focal_bwd_scalar": "focal_kernel", "frozen_bn_fwd": "frozen_bn",
           "nms_batched": "nms_fused_kernel",
           "frozen_bn_bwd": "frozen_bn", "bias_act_bwd": "bias_act_bwd_nhwc_kernel", "dcn_col2im": "col2im", "dcn_im2col": "im2col_kernel",
           "dcn_col2im_coord": "col2im_coord", "dcn_fused_fwd": "dcn_fused_fwd", "dcn_to_nhwc": "nchw_to_nhwc",
           "dcn_im2col_nhwc": "im2col_nhwc_kernel", "dcn_coord_nhwc": "coord_nhwc_kernel",
           "dcn_transposed_sample": "sampleT_gather_kernel", "dcn_col2im_nhwc": "col2im_nhwc_gather_kernel"}.get(entry_name.split("[")[0])
    if key is None:
        return None
    bins = re.search(r",(\d+)x(\d+)\]", entry_name)
    bn = re.match(r"frozen_bn_(fwd|bwd)\[n=\d+,nc=(\d+),e=\d+,res=(\d)", entry_name)
    fam = entry_name.split("[")[0]

    def matches(line):
        if "mean=" not in line:
            return False
        if bn:   # frozen_bn_{fwd,bwd}[_nhwc]_kernel<T, V, relu, residual>
            want = "frozen_bn_%s_" % bn.group(1)
            flag = ", %s>" % ("true" if bn.group(3) == "1" else "false")
            if want not in line or flag not in line:
                return False
            if "nhwc_kernel" in line:
                return True      # (the channels-last form's grid does not encode the plane count: the first entry of that flavour)
            return ("grid=%d " % (int(bn.group(2)) * 256)) in line
        if fam == "roi_align_fpn_fwd" and "roi_align_fwd_nhwc_kernel<" in line and bins:
            # channels-last pyramid: <V, kOutNhwc, threads>: the box head's 7 x 7 call returns [K, C, PH, PW] (kOutNhwc = false),
            # the mask head's 14 x 14 call a channels-last tensor
            return (", false," in line) == (bins.group(1) == "7")
        if key in line:
            return (not bins or "roi_align" not in key or ("<%s, %s" % bins.groups()) in line)
        return False

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*kernel_times*.txt")), reverse=True):
        for line in open(path):
            if matches(line):
                m = re.search(r"mean=\s*([0-9.]+) us", line)
                if m:
                    return {"us": float(m.group(1)), "source": os.path.relpath(path, ROOT)}
    return None


def measured_traffic(kernel_name, impl_hint=None):
    """HBM bytes per launch from the committed PMC passes (profiles/*traffic*.json, produced by
    tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
    tools/opbench.py, whose FPN-fused launches have exactly the model's shapes), newest file first,
    or None when no measurement matches this entry point.  `impl_hint` narrows the match to one
    kernel (e.g. "bwd_gather") when several implementations were profiled."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            table = json.load(open(path))
        except Exception:
            continue
        for key, v in table.items():
            if v.get("tag") == kernel_name and (impl_hint is None or impl_hint in key):
                return v["hbm_bytes"], os.path.relpath(path, ROOT) + " :: " + key
    return None, None


def cpu_baseline(sample_rois=1024, min_seconds=8.0):
    """reference CPU ROIAlign forward on a bounded sample of the box-head workload (1 thread)."""
    import numpy as np
    import torch

    import oracle
    import synth

    rng_feats = [np.random.RandomState(10 + i).randn(2, 256, h, w).astype(np.float32)
                 for i, (h, w) in enumerate(synth.fpn_shapes()[:4])]
    rois = synth.fpn_rois(seed=3, per_image=512, n_images=2)
    lv = synth.level_map(rois)
    pick = np.random.RandomState(0).permutation(len(rois))[:sample_rois]
    ref = oracle.ref()
    torch.set_num_threads(1)
    feats_t = [torch.from_numpy(f) for f in rng_feats]
    reps, t0 = 0, time.perf_counter()
    while True:
        for l in range(4):
            sel = pick[lv[pick] == l]
            if sel.size == 0:
                continue
            if ref is not None:
                ref.roi_align_forward(feats_t[l], torch.from_numpy(rois[sel]), 1.0 / (4 << l), 7, 7, 2)
            else:
                oracle.roi_align_forward(rng_feats[l], rois[sel], 1.0 / (4 << l), 7, 7, 2)
        reps += 1
        if time.perf_counter() - t0 >= min_seconds or reps >= 50:
            break
    dt = (time.perf_counter() - t0) / reps
    extra = cpu_baseline_extras(ref)
    try:
        extra["all_cores"] = cpu_baseline_all_cores()
    except Exception as e:  # noqa: BLE001
        extra["all_cores"] = {"error": repr(e)}
    # scale the sample to the full launch: per-ROI cost dominates; bytes per §8d for the full launch
    full_bytes = 4 * 1024 * 256 * 49 + sum(f.nbytes for f in rng_feats) + 20 * 1024
    est_full_s = dt * 1024.0 / sample_rois
    return {"value": round(full_bytes / est_full_s / 1e9, 4), "unit": "GB/s", "cores": 1,
            "kind": "reference" if ref is not None else "port",
            "sample": "ROIAlign fwd box-head (1024 ROIs x 256 ch x 7x7 sr2 over P2-P5 of 2x800x1344): %d of the 1024 "
                      "ROIs per pass, %d passes, %.3f s per pass on 1 thread (scaled x%.1f to the full launch); %s" % (
                          sample_rois, reps, dt, 1024.0 / sample_rois,
                          "reference csrc/cpu/ROIAlign_cpu.cpp compiled in oracle/_ref" if ref is not None
                          else "C restatement oracle/detops_oracle.c"),
            "roi_align_fwd_ms_est": round(est_full_s * 1e3, 2), **extra}


def _roi_worker(args):
    """one host core's share of the box-head ROIAlign forward through the reference's CPU kernel (oracle/_ref)"""
    import numpy as np
    import torch

    import oracle
    import synth

    lo, hi, reps = args
    torch.set_num_threads(1)
    ref = oracle.ref()
    feats = [((torch.arange(2 * 256 * h * w, dtype=torch.float32) % 251.0) * 0.01).view(2, 256, h, w)   # cheap fill
             for (h, w) in synth.fpn_shapes()[:4]]
    rois = synth.fpn_rois(seed=3, per_image=512, n_images=2)
    lv = synth.level_map(rois)
    mine = np.arange(lo, hi)
    t0 = time.perf_counter()
    for _ in range(reps):
        for l in range(4):
            sel = mine[lv[mine] == l]
            if sel.size:
                if ref is not None:
                    ref.roi_align_forward(feats[l], torch.from_numpy(rois[sel]), 1.0 / (4 << l), 7, 7, 2)
                else:
                    oracle.roi_align_forward(feats[l].numpy(), rois[sel], 1.0 / (4 << l), 7, 7, 2)
    return (time.perf_counter() - t0) / reps


def cpu_baseline_all_cores(max_workers=None):
    """the node figure — one single-threaded worker per host core, the 1024 box-head ROIs
    split evenly (what a CPU data-parallel run of the reference does); wall time = the slowest worker."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = max(1, min(cores, max_workers or 32))   # each worker holds its own 183 MB copy of the pyramid: capped at 32
    per = -(-1024 // n)
    jobs = [(i * per, min(1024, (i + 1) * per), 2) for i in range(n) if i * per < 1024]
    ctx = mp.get_context("spawn")   # CUDA is initialised in the parent: no fork
    t0 = time.perf_counter()
    with ctx.Pool(len(jobs)) as pool:
        times = pool.map_async(_roi_worker, jobs).get(timeout=120)
    wall = time.perf_counter() - t0
    full_bytes = 4 * 1024 * 256 * 49 + sum(4 * 2 * 256 * h * w for (h, w) in [(200, 336), (100, 168), (50, 84), (25, 42)]) + 20 * 1024
    return {"workers": len(jobs), "host_cores": cores, "slowest_worker_s": round(max(times), 4),
            "GBs": round(full_bytes / max(times) / 1e9, 3), "pool_wall_s_incl_startup": round(wall, 2)}


def cpu_baseline_extras(ref, budget_s=6.0):
    """The other CPU paths of SURVEY.md §8(d), timed beside the ROIAlign figure on the same host
    (1 thread, bounded): the reference's own `nms` CPU kernel on the step's 10 RPN segments, the C
    restatements of the CUDA-only ROIAlign backward and SigmoidFocalLoss (sampled, scaled)."""
    import numpy as np
    import torch

    import oracle
    import synth

    out = {}
    try:
        segs = synth.rpn_nms_segments()
        t0 = time.perf_counter()
        reps = 0
        while reps < 3 and time.perf_counter() - t0 < budget_s / 3:
            for b, sc in segs:
                if ref is not None:
                    ref.nms(torch.from_numpy(b), torch.from_numpy(sc), 0.7)
                else:
                    oracle.nms(b, sc, 0.7)
            reps += 1
        out["nms_10_rpn_segments_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 2)
        out["nms_kind"] = "reference" if ref is not None else "port"
        # ROIAlign backward (CUDA-only in the reference: C restatement), 64 of the 1024 box-head ROIs on P4
        rois = synth.fpn_rois(seed=3, per_image=512, n_images=2)
        lv = synth.level_map(rois)
        sel = np.nonzero(lv == 2)[0][:64]
        g = np.random.RandomState(1).randn(len(sel), 256, 7, 7).astype(np.float32)
        t0 = time.perf_counter()
        oracle.roi_align_backward(g, rois[sel], 1.0 / 16, 7, 7, 2, 256, 50, 84, 2)
        out["roi_align_bwd_ms_est_port"] = round((time.perf_counter() - t0) * 1024.0 / max(len(sel), 1) * 1e3, 1)
        # SigmoidFocalLoss: the reference's own CPU path is the Python composite `sigmoid_focal_loss_cpu`
        # (layers/sigmoid_focal_loss.py:40-50), restated here line by line on torch CPU tensors and timed on the
        # FULL RetinaNet shape (R = 403,200 rows x 80 classes, all host threads torch gives it); the C restatement
        # of the CUDA formula on a 20,000-row sample is kept beside it
        logits, targets = synth.focal_inputs(403200, 80)
        tl, tt = torch.from_numpy(logits), torch.from_numpy(targets.astype(np.int64))
        torch.set_num_threads(max(1, os.cpu_count() or 1))
        t0 = time.perf_counter()
        num_classes = tl.shape[1]
        class_range = torch.arange(1, num_classes + 1, dtype=tt.dtype).unsqueeze(0)
        t = tt.unsqueeze(1)
        p = torch.sigmoid(tl)
        term1 = (1 - p) ** 2.0 * torch.log(p)
        term2 = p ** 2.0 * torch.log(1 - p)
        loss = -(t == class_range).float() * term1 * 0.25 - ((t != class_range) * (t >= 0)).float() * term2 * (1 - 0.25)
        float(loss.sum())
        out["focal_fwd_ms_reference_python_cpu"] = round((time.perf_counter() - t0) * 1e3, 1)
        out["focal_cpu_threads"] = torch.get_num_threads()
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        oracle.sigmoid_focal_loss_forward(logits[:20000], targets[:20000], 2.0, 0.25)
        out["focal_fwd_ms_est_port"] = round((time.perf_counter() - t0) * 403200.0 / 20000.0 * 1e3, 1)
    except Exception as e:  # extras never cost the main figure
        out["extras_error"] = repr(e)
    return out


HW_QUEUES_DEFAULT = "2"


def mask_slots_report(model):
    try:
        from maskrcnn_benchmark.modeling.roi_heads.mask_head import mask_head as mh
        m = getattr(model, "module", model)
        head = m.roi_heads["mask"] if (m.roi_heads and "mask" in m.roi_heads) else None
        return None if head is None else {"mode": mh.slot_mode(), "slots": head.last_slots, "quota": head.max_positives}
    except Exception:  # noqa: BLE001 — a report field only
        return None


def prewarm_mask_batch_sizes(model, step, batches, images_per_gpu):
    """one training step per possible mask-head batch size (mask_head.SLOT_MODE == "dynamic"): total slots from one granule per
    image to the quota per image, in steps of one granule -> number of steps run"""
    rep = mask_slots_report(model)
    if not rep or rep["mode"] != "dynamic":
        return 0
    from maskrcnn_benchmark.modeling.roi_heads.mask_head import mask_head as mh
    g, quota, n = mh.SLOT_GRANULE, rep["quota"], int(images_per_gpu)
    done = 0
    try:
        for total in range(n * g, n * quota + 1, g):
            per, extra = divmod(total // g, n)           # granules per image, the first `extra` images take one more
            mh.SLOT_MODE = ",".join(str(min(quota, (per + (1 if i < extra else 0)) * g)) for i in range(n))
            step(*batches[done % len(batches)])
            done += 1
    finally:
        mh.SLOT_MODE = "dynamic"
    return done


def graph_env(argv):
    """--hip-graph: ROCm's graph capture of kernel-argument packets must be off BEFORE the HIP runtime starts (the replay of a
    captured training step faults otherwise: profiles/r04a_hip_graph_flags.txt)"""
    if "--hip-graph" in argv:
        os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def pin_hip_queues():
    """GPU_MAX_HW_QUEUES: HIP maps streams onto this many hardware queues.  The data-parallel step has exactly two busy
    streams (compute, RCCL); measured at one rank through the full wrapper (profiles/r03n_data_parallel_overhead.txt):
    8 / runtime default / 2 / 1 queues -> 70.6 / 41.5-42.8 / 41.0 / 40.0 ms per step.  2 keeps the all-reduce on its own
    queue (overlap with backward) without the many-queue penalty; an explicit setting in the environment wins.  Read by
    the HIP runtime at initialisation: must run before the first torch.cuda call."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", HW_QUEUES_DEFAULT)
    # kernel arguments in device memory: this image's runtime default, pinned because the step is ~1500 dependent launches
    # (HIP_FORCE_DEV_KERNARG = 0 measured: fp32 39.0 -> 41.0 ms, bf16 22.1 -> 24.2 ms per step, profiles/r04ak_env_knobs.txt)
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    return os.environ["GPU_MAX_HW_QUEUES"]


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch_command(argv, gpus, port=None):
    """The command `bench.py --gpus N` re-executes itself as when nothing launched it as a rank (no WORLD_SIZE)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def check_world(gpus, world, backend_world=None):
    """--gpus is a promise about the number of RCCL ranks: never print a line measured on a different world."""
    if world != gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, "
                         "or without WORLD_SIZE so that bench.py launches the ranks itself)" % (gpus, world, gpus))
    if backend_world is not None and backend_world != gpus:
        raise SystemExit("bench.py: --gpus %d but the process group has %d ranks" % (gpus, backend_world))


def timed_steps(step, batches, steps, sync, distributed, device, before=None):
    """The contract's timed region: barrier + device sync, EXACTLY `steps` steps, barrier + device sync; the clock is
    the MAX over ranks.  Returns (elapsed seconds, this rank's host-enqueue seconds, the last step's result)."""
    import torch
    import torch.distributed as dist
    sync()
    if before is not None:
        before()
    host_pad = float(os.environ.get("BENCH_HOST_PAD_MS", "0")) / 1000.0   # diagnosis: is the host or the device the limiter?
    out = None
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(*batches[i % len(batches)])
        if host_pad:
            time.sleep(host_pad)
    host_elapsed = time.perf_counter() - t0      # everything enqueued; the device may still be running
    sync()
    elapsed = time.perf_counter() - t0
    RANK_SECONDS[:] = [elapsed]
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(every, t)                       # diagnosis: every rank's own clock (min / max in the line)
        RANK_SECONDS[:] = [float(v.item()) for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, host_elapsed, out


def stub_main(args):
    """Test hook (`--stub-step`, tests/test_bench_launch.py): the launch / rendezvous / world check / timed region /
    max-over-ranks / one-JSON-line logic of this script on CPU tensors over gloo with a stub step — no model, no HIP
    library, NOT a measurement (the line says so).  What it proves: `bench.py --gpus N` really runs N ranks."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    check_world(args.gpus, world)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", init_method="env://")
        check_world(args.gpus, world, dist.get_world_size())
    device = torch.device("cpu")
    w = torch.ones(64, 64)

    def step(x):
        y = (x @ w).sum().reshape(1)
        if distributed:
            dist.all_reduce(y)            # every rank contributes: the sum counts the ranks
        return {"ranks_seen": y / (x @ w).sum()}

    def sync():
        if distributed:
            dist.barrier()

    batches = [(torch.full((8, 64), float(i + 1)),) for i in range(2)]
    for i in range(args.warmup):
        step(*batches[i % 2])
    elapsed, host_elapsed, out = timed_steps(step, batches, args.steps, sync, distributed, device)
    if rank == 0:
        images = args.images_per_gpu * world * args.steps
        print(json.dumps({"metric": "STUB (bench.py --stub-step: launch-logic test, not a measurement)",
                          "value": round(images / elapsed, 3), "unit": "images/sec", "n_gpus": world,
                          "rccl_ranks": dist.get_world_size() if distributed else 1, "backend": "gloo",
                          "ranks_seen_by_allreduce": round(float(out["ranks_seen"]), 3),
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "scaling": "weak", "data": "stub", "stub": True}), flush=True)
    if distributed:
        dist.destroy_process_group()


def eval_main(args, cfg, device, layout, miopen_db, hw_queues, progress):
    """`--eval`: the detector's inference forward (reference engine/inference.py:18-41 compute_on_dataset's `model(images)`,
    timed like its Timer: per-image seconds over the loop), eval mode under no_grad, synthetic test-size images resident on
    the device, the test-time RPN selection and the box / mask (or RetinaNet) PostProcessors inside the timed region.  N = 1.
    With random-init weights every class score is ~1/81 < the yaml's SCORE_THRESH (0.05), so the box PostProcessor's
    per-class NMS sees no candidates; pass `MODEL.ROI_HEADS.SCORE_THRESH 0.0` (`MODEL.RETINANET.INFERENCE_TH 0.0`) for the
    opposite extreme (every candidate of every class enters the segmented NMS).  Both are in DESIGN.md §6."""
    import torch
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.engine.bench_step import choose_layout, make_device_batches
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("bench.py --eval: N = 1 only (inference shards by image; no collective to measure)")
    model = build_detection_model(cfg).to(device).eval()
    layout = choose_layout(cfg, device, layout)
    if layout != "nchw":
        model.set_channels_last(True, heads=layout == "all")
    batches = make_device_batches(cfg, device, images_per_gpu=args.images_per_gpu, num_batches=2, seed=0,
                                  height=cfg.INPUT.MIN_SIZE_TEST, width=cfg.INPUT.MAX_SIZE_TEST)
    H, W = batches[0][0].tensors.shape[-2:]
    amp = {"float32": None, "bfloat16": torch.bfloat16, "float16": torch.float16}[cfg.DTYPE]

    def step(images, _targets=None):
        with torch.no_grad():
            if amp is None:
                return model(images)
            with torch.autocast(device_type="cuda", dtype=amp):
                return model(images)

    det = None
    for i in range(args.warmup):
        det = step(*batches[i % len(batches)])
        torch.cuda.synchronize(device)
        progress("warm-up forward %d done" % (i + 1))
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        det = step(*batches[i % len(batches)])
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    images = args.images_per_gpu * args.steps
    counts = [len(d) for d in det]
    finite = all(bool(torch.isfinite(d.bbox).all()) and bool(torch.isfinite(d.get_field("scores")).all()) for d in det)
    thresh = cfg.MODEL.RETINANET.INFERENCE_TH if cfg.MODEL.RETINANET_ON else cfg.MODEL.ROI_HEADS.SCORE_THRESH
    line = {
        "metric": "inference images/sec %s" % os.path.splitext(os.path.basename(args.config))[0],
        "value": round(images / elapsed, 3), "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "s_per_image": round(elapsed / images, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"float32": "f32", "bfloat16": "bf16", "float16": "f16"}[cfg.DTYPE], "data": "synthetic",
        "config": {"workload": "%s: inference forward + post-processing, %d img/GPU of synthetic %dx%d (padded %dx%d), "
                               "random-init weights" % (os.path.basename(args.config), args.images_per_gpu,
                                                        cfg.INPUT.MAX_SIZE_TEST, cfg.INPUT.MIN_SIZE_TEST, H, W),
                   "global_batch": args.images_per_gpu, "parallelism": "dp1", "score_thresh": thresh},
        "detections_per_image": counts, "detections_finite": finite,
        # context only: the reference's published inference time for e2e_mask_rcnn_R_50_FPN_1x (MODEL_ZOO.md:26, 8 x V100,
        # fp32, trained weights) — other hardware and other score statistics, so vs_baseline stays null
        "reference_published_s_per_image": 0.12966 if "mask_rcnn_R_50_FPN" in args.config else None,
        "max_mem_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
        "miopen": {"search": bool(torch.backends.cudnn.benchmark), "db": miopen_db}, "layout": layout,
        "hip": {"GPU_MAX_HW_QUEUES": hw_queues, "HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG")},
        "nms_repaired_segments": _C.nms_repaired_segments(device),
    }
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    graph_env(sys.argv)
    hw_queues = pin_hip_queues()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        cmd = self_launch_command(sys.argv[1:], args.gpus)
        print("[bench] launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        raise SystemExit(subprocess.call(cmd))
    if args.stub_step:
        return stub_main(args)
    miopen_db = setup_miopen_db(args.export_miopen_db)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    check_world(args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the detection-head operators are HIP-only)"
    if torch.cuda.device_count() < max(1, world if int(os.environ.get("LOCAL_WORLD_SIZE", world)) == world else 1):
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")
        check_world(args.gpus, world, dist.get_world_size())
    elif args.force_ddp:
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % (29500 + os.getpid() % 400),
                                rank=0, world_size=1)

    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches

    opts = list(args.opts)
    if opts and opts[0] == "--":
        opts = opts[1:]
    if args.dtype:
        opts += ["DTYPE", args.dtype]
    # The yaml's BASE_LR (0.02) is for the reference's 16-image global batch (8 GPUs x 2).  For other batch sizes the
    # reference's own recipe is the linear scaling rule (README.md "Single GPU Training": IMS_PER_BATCH 2 -> BASE_LR
    # 0.0025); at 0.02 with 2 images the random-init detector diverges in some runs.  No effect on the work per step.
    global_batch = args.images_per_gpu * world
    if "SOLVER.BASE_LR" not in opts:
        base = load_cfg(args.config, []).SOLVER.BASE_LR
        opts += ["SOLVER.BASE_LR", base * global_batch / 16.0, "SOLVER.IMS_PER_BATCH", global_batch]
    cfg = load_cfg(args.config, opts)
    torch.manual_seed(1234 + rank)
    # MIOpen: immediate mode by default (see --miopen-search)
    torch.backends.cudnn.benchmark = bool(args.miopen_search) and not args.no_miopen_search
    t_start = time.perf_counter()

    def progress(msg):
        if rank == 0:
            print("[bench %6.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    layout = "all" if args.channels_last_heads else ("backbone" if args.channels_last else args.layout)
    if args.eval:
        return eval_main(args, cfg, device, layout, miopen_db, hw_queues, progress)
    model, optimizer, scheduler, step = build_training(cfg, device, distributed, local_rank,
                                                       force_ddp=args.force_ddp and not distributed,
                                                       bucket_cap_mb=args.bucket_mb, layout=layout)
    layout = getattr(getattr(model, "module", model), "layout", layout)
    eager_step = step
    if args.hip_graph:
        if distributed or args.force_ddp:
            raise SystemExit("bench.py --hip-graph: single process only (the data-parallel wrapper is not captured)")
        from maskrcnn_benchmark.engine.graph_step import GraphedTrainStep
        step = GraphedTrainStep(eager_step, warmup=3, max_graphs=4)
    batches = make_device_batches(cfg, device, images_per_gpu=args.images_per_gpu, num_batches=2, seed=rank)
    feat_bytes = 0
    H, W = batches[0][0].tensors.shape[-2:]
    for s in (4, 8, 16, 32):
        feat_bytes += 4 * args.images_per_gpu * 256 * ((H + s - 1) // s) * ((W + s - 1) // s)

    def sync():
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    progress("model + %d device batches ready" % len(batches))
    losses = None
    ddp_mode = "overlapped-sgd-hook" if (distributed or args.force_ddp) else "single"
    call_counter = None
    for i in range(args.warmup):
        if i == args.warmup - 1 and not args.no_kernel_timing:
            call_counter = _C.KernelTimer(count_only=True)   # how often is each entry point called per step?  (no events)
            _C.KERNEL_TIMER = call_counter
        try:
            losses = step(*batches[i % len(batches)])
            torch.cuda.synchronize(device)
        except Exception as e:  # noqa: BLE001
            # Opt-in (--allow-ddp-fallback) safety net for the N > 1 path: a Python-level failure of the update-in-the-
            # all-reduce-callback in the first step falls back to the same wrapper without it (bucketed all-reduce
            # overlapped with backward, then the optimizer).  Off by default: a scaling run must measure the design
            # DESIGN.md describes.
            if not args.allow_ddp_fallback or not distributed or i > 0 or ddp_mode != "overlapped-sgd-hook":
                raise
            progress("overlapped update failed (%r): falling back to all-reduce, then optimizer" % (e,))
            ddp_mode = "allreduce-then-sgd (fallback: %s)" % type(e).__name__
            del model, optimizer, scheduler, step
            torch.manual_seed(1234 + rank)
            model, optimizer, scheduler, step = build_training(cfg, device, distributed, local_rank,
                                                               overlap_optimizer=False, layout=layout)
            losses = step(*batches[i % len(batches)])
            torch.cuda.synchronize(device)
        progress("warm-up step %d done" % (i + 1))
    # The mask head's batch follows the number of positives (mask_slots): up to 15 batch sizes at 2 images per GPU, each a
    # MIOpen problem of its own whose FIRST use loads its kernels (tens of ms).  A long run amortises that; a 20-step timed
    # region would carry whichever sizes the warm-up steps happened not to see (measured: 31.4 instead of 29.1 ms per step).
    # One untimed step per batch size, forced, before the timed region — on every rank (the same collective sequence).
    _C.KERNEL_TIMER = None      # (the call counter of the last warm-up step must not see the steps below)
    prewarmed = prewarm_mask_batch_sizes(model, step, batches, args.images_per_gpu) if (device.type == "cuda" and not args.hip_graph) else 0
    if prewarmed:
        torch.cuda.synchronize(device)
        progress("%d untimed steps over the mask head's batch sizes done" % prewarmed)
    # The timed region carries NO kernel timers (round 6): an event pair around a launch drains the queue
    # (~30 us of device time per timed launch in the fp32 step, ~100 us under fp16), which used to cost the driver's
    # 20-step line ~0.4 ms per step.  `value` comes from this loop; the per-kernel figures from the post-pass below.
    gpu_t0 = time.perf_counter()
    elapsed, host_elapsed, losses = timed_steps(step, batches, args.steps, sync, distributed, device)
    progress("%d timed steps done" % args.steps)
    # Kernel-timing post-pass, OUTSIDE the timed region: more steps of the same training loop with HIP events around the
    # hand-written entry points on their launch stream.  Every entry point gets >= KERNEL_SAMPLES event pairs (SURVEY 8d:
    # >= 50 launches): once-per-step flagships (ROIAlign fwd / bwd, the segmented NMS, the losses) are timed at every call,
    # high-frequency ones (FrozenBN: ~100 calls per step) are sampled with a stride.  Runs on every rank (the same
    # collective sequence everywhere); rank 0's figures are reported.
    timer = None
    post_steps = 0
    exposed_ms = None
    if not args.no_kernel_timing:
        calls = call_counter.calls if call_counter else {}
        post_steps = args.kernel_timing_steps
        strides = {name: max(1, (n * post_steps) // KERNEL_SAMPLES) for name, n in calls.items()}
        timer = _C.KernelTimer(every_cap=max(1, post_steps // 10), strides=strides)
        _C.KERNEL_TIMER = timer
        if hasattr(model, "exposed_wait_events"):
            model.exposed_wait_events = []       # BucketedDataParallel: event pair around the main stream's join with the side stream
        timed_step = step.run_eager if args.hip_graph else step     # (Python-level timers see nothing of a graph replay)
        for i in range(post_steps):
            timed_step(*batches[i % len(batches)])
        sync()
        _C.KERNEL_TIMER = None
        if getattr(model, "exposed_wait_events", None):
            ev = model.exposed_wait_events
            model.exposed_wait_events = None
            exposed_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        progress("%d kernel-timing steps done (outside the timed region)" % post_steps)
    # Transparency line for the mask head's dynamic batch (mask_head.py: every image's positives rounded up to a granule of 16 slots, the
    # reference's workload): the same step with the FIXED quota of 128 slots per image, i.e. what the step costs when every
    # image fills its quota of positives — 20 more steps outside the timed region, on every rank.
    slots_dynamic = mask_slots_report(model)
    fixed_quota_ms = None
    if slots_dynamic and slots_dynamic["mode"] == "dynamic" and not args.no_fixed_quota_line and device.type == "cuda":
        from maskrcnn_benchmark.modeling.roi_heads.mask_head import mask_head as _mh
        _mh.SLOT_MODE = "fixed"
        saved_rank_seconds = list(RANK_SECONDS)
        try:
            for i in range(4):
                step(*batches[i % len(batches)])
            sync()
            q_elapsed, _, _ = timed_steps(step, batches, 20, sync, distributed, device)
            fixed_quota_ms = round(q_elapsed / 20 * 1e3, 3)
        finally:
            _mh.SLOT_MODE = "dynamic"
            RANK_SECONDS[:] = saved_rank_seconds
        progress("20 steps with the fixed mask-head quota done (outside the timed region)")
    gpu_phase_s = time.perf_counter() - gpu_t0
    loss_vals = {k: float(v.detach()) for k, v in losses.items()} if losses else {}
    # N > 1 diagnosis (VERDICT r05 #4b): every rank's communication path and its exposed all-reduce time, gathered on rank 0
    ddp_ranks = None
    if distributed:
        mine = {"rank": rank, "comm_mode": getattr(model, "comm_mode", None), "comm_note": getattr(model, "comm_note", None),
                "exposed_allreduce_ms": None if exposed_ms is None else round(exposed_ms, 4)}
        box = [None] * world
        dist.all_gather_object(box, mine)
        ddp_ranks = box

    if rank == 0:
        images = args.images_per_gpu * world * args.steps
        line = {
            "metric": "training images/sec %s" % os.path.splitext(os.path.basename(args.config))[0],
            "value": round(images / elapsed, 3),
            "unit": "images/sec",
            "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if (distributed or args.force_ddp) else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            # the reference publishes 0.4536 s/iter at 2 im/GPU on 8 x V100 = 35.3 img/s for
            # this config (MODEL_ZOO.md:26, fp32).  That is an 8-GPU figure: compared only at N = 8.
            "vs_baseline": round(images / elapsed / 35.3, 3) if (world == 8 and "mask_rcnn_R_50_FPN" in args.config) else None,
            "dtype": {"float32": "f32", "bfloat16": "bf16", "float16": "f16"}[cfg.DTYPE],
            "data": "synthetic",
            "config": {"workload": "%s: fwd+bwd+allreduce+SGD, %d img/GPU of synthetic 1333x800 (padded %dx%d), "
                                   "random-init weights" % (os.path.basename(args.config), args.images_per_gpu, H, W),
                       "global_batch": args.images_per_gpu * world, "parallelism": "dp%d" % world,
                       "base_lr": cfg.SOLVER.BASE_LR},
            "loss_finite": all(v == v and abs(v) != float("inf") for v in loss_vals.values()),
            "losses": {k: round(v, 4) for k, v in loss_vals.items()},
            "max_mem_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
            "miopen": {"search": bool(torch.backends.cudnn.benchmark), "db": miopen_db},
            # which activations are channels-last (NHWC): "nchw" none | "backbone" ResNet + FPN | "all" the heads as well
            "layout": layout,
            # mask head batch: "dynamic" = every image's positives rounded up to a granule of 16 slots (the reference runs its mask head on the
            # positive boxes only), "fixed" = the quota of 128 slots per image; `slots` = the last step's slot counts per image
            "mask_slots": dict(slots_dynamic, fixed_quota_ms_per_step=fixed_quota_ms) if slots_dynamic else None,
            # True: the timed steps were replays of captured HIP graphs (engine/graph_step.py); the kernel timers' post-pass ran eagerly
            "hip_graph": ({"replays": step.replays, "graphs": len(step._graphs), "eager_steps": step.eager_steps} if args.hip_graph else False),
            "hip": {"GPU_MAX_HW_QUEUES": hw_queues, "HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG")},
            "kernel_timers": ("post-pass of %d steps outside the timed region, %s" % (post_steps, "all ranks" if distributed else "rank 0"))
                             if timer is not None else "off",
            # wall time of the GPU phase of this command (timed region + kernel-timing post-pass): the rest is model build,
            # warm-up and the CPU baseline
            "gpu_phase_s": round(gpu_phase_s, 2),
            "ddp": ddp_mode,
            # which communication path the wrapper runs: "direct" = RCCL on one low-priority side stream (engine/rccl_comm.py),
            # "pg" = ProcessGroupNCCL (with the reason, when the direct path was tried and refused)
            "ddp_comm": ({"mode": getattr(model, "comm_mode", None), "note": getattr(model, "comm_note", None)}
                         if (distributed or args.force_ddp) else None),
            # per-rank view of the timed region (N > 1): each rank's own ms per step (the line's clock is the max), the
            # communication path every rank took, and the main stream's wait for the side stream at the end of backward
            # (= all-reduce + update time the backward pass did not hide; measured in the post-pass)
            "rank_ms_per_step": ({"min": round(min(RANK_SECONDS) / args.steps * 1e3, 3), "max": round(max(RANK_SECONDS) / args.steps * 1e3, 3),
                                  "all": [round(v / args.steps * 1e3, 3) for v in RANK_SECONDS]} if RANK_SECONDS else None),
            "exposed_allreduce_ms": None if exposed_ms is None else round(exposed_ms, 4),
            # NMS segments the single-launch kernel gave up on and the repair launch redid since the start of the process
            # (results are complete either way; non-zero = the scan's waits timed out beside other work on the CUs)
            "nms_repaired_segments": _C.nms_repaired_segments(device),
            "ddp_ranks": ddp_ranks,
            # host time to ENQUEUE the timed steps (rank 0): close to ms_per_step = the host is the limiter
            "host_enqueue_ms_per_step": round(1000.0 * host_elapsed / args.steps, 3),
        }
        if timer is not None:
            kernels, dominant = {}, None
            for name, (calls, timed, total_ms) in sorted(timer.results().items()):
                b = algorithmic_bytes(name, feat_bytes)
                mean_us = total_ms / max(timed, 1) * 1e3
                # sampled entry points (FrozenBN): the per-step figure scales the sampled mean to every call
                entry = {"launches": calls, "timed": timed, "mean_us": round(mean_us, 2),
                         "ms_per_step": round(mean_us * calls / max(post_steps, 1) / 1e3, 4)}
                if b is not None:
                    entry["alg_bytes"] = b
                    entry["achieved_GBs"] = round(b / (mean_us * 1e-6) / 1e9, 1)
                kernels[name] = entry
            # one `roofline` entry point: among the SURVEY §8(a) operators of the path BASELINE.json's metric names
            # (ROIAlign, focal loss, deformable conv — `PATH_FAMILIES`), the kernel FAMILY (name before the shape) with the
            # largest time per step, represented by its member with the largest time per step.  The fused FrozenBN
            # streams are larger per step but are not a §8 row: they stay in `kernels` / `kernel_families_ms_per_step`.
            fam = {}
            for name, e in kernels.items():
                if "alg_bytes" in e:
                    fam.setdefault(name.split("[")[0], []).append((e["ms_per_step"], name))
            path_fam = {k: v for k, v in fam.items() if k.startswith(PATH_FAMILIES)}
            if path_fam:
                best = max(path_fam.values(), key=lambda members: sum(m[0] for m in members))
                name = max(best)[1]
                dominant = (name, kernels[name])
            line["kernels"] = kernels
            line["kernel_families_ms_per_step"] = {k: round(sum(m[0] for m in v), 4) for k, v in sorted(fam.items())}
            def roofline_entry(name, e, selection):
                traffic, source = measured_traffic(name)
                return {"kernel": name, "bound": "hbm", "achieved": e["achieved_GBs"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(e["achieved_GBs"] / HBM_PEAK_GBS, 4),
                        "traffic": traffic,
                        # the PMC passes are separate rocprofv3 runs (committed table), not this run
                        "traffic_source": source,
                        "alg_bytes_per_launch": e["alg_bytes"], "mean_us": e["mean_us"], "timed": e["timed"],
                        "family_ms_per_step": round(sum(m[0] for m in fam[name.split("[")[0]]), 4),
                        "rocprof_us": rocprof_kernel_us(name), "selection": selection}

            if dominant is not None:
                line["roofline"] = roofline_entry(dominant[0], dominant[1],
                                                  "largest family per step among the SURVEY 8(a) operators (%s*)" % "* / ".join(PATH_FAMILIES))
            if fam:
                # the same rule over EVERY hand-written family with an algorithmic-bytes figure (the fused FrozenBN streams
                # included, which are not a SURVEY 8 row): both selections are printed so the line means the same every round
                best_all = max(fam.values(), key=lambda members: sum(m[0] for m in members))
                name_all = max(best_all)[1]
                line["roofline_all"] = roofline_entry(name_all, kernels[name_all], "largest hand-written family per step, any operator")
            # the SURVEY §8(a) flagship entry points of the path BASELINE.json's metric names ("ROIAlign HBM GB/s", NMS,
            # focal loss), whatever kernel family dominates the step: live HIP-event figure, the committed rocprofv3
            # average of the same kernel and the committed PMC traffic beside it
            path = []
            for name, e in kernels.items():
                fam_name = name.split("[")[0]
                if fam_name in ("roi_align_fpn_fwd", "roi_align_fpn_bwd", "focal_fwd_sum", "focal_bwd_scalar"):
                    traffic, source = measured_traffic(name)
                    path.append({"kernel": name, "bound": "hbm", "mean_us": e["mean_us"], "timed": e["timed"],
                                 "alg_bytes_per_launch": e["alg_bytes"], "achieved": e["achieved_GBs"], "unit": "GB/s",
                                 "peak": HBM_PEAK_GBS, "frac": round(e["achieved_GBs"] / HBM_PEAK_GBS, 4),
                                 "rocprof_us": rocprof_kernel_us(name), "traffic": traffic, "traffic_source": source})
                elif fam_name == "nms_batched":
                    # neither HBM nor MFMA bound (O(n^2) VALU + a serial chain): reported as us per launch
                    path.append({"kernel": name, "bound": "valu+latency (no roofline claimed)", "mean_us": e["mean_us"],
                                 "timed": e["timed"], "rocprof_us": rocprof_kernel_us(name)})
            if path:
                line["roofline_path"] = path
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is a reported extra; never lose the bench line
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if distributed or args.force_ddp:
        if hasattr(model, "close"):   # the wrapper's own RCCL communicator + side stream: released before the process group
            model.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
