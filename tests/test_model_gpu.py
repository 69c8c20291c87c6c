"""GPU tests of the model-surface pieces that call the HIP operators: dense-mask segmented NMS vs
the oracle (bit-exact), fused FPN pooler vs per-level oracle ROIAlign, and a finite training step of
each BASELINE architecture on a small image."""
import os

import numpy as np
import pytest
import torch

import oracle
import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


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


def test_nms_batched_mask_12000_candidates_per_segment():
    """PRE_NMS_TOP_N_TRAIN = 12000 (the reference's non-FPN default): segments beyond the in-LDS sort."""
    from maskrcnn_benchmark import _C
    segs = [synth.nms_boxes(n, seed=31 + i) for i, n in enumerate((12000, 9500))]
    boxes = np.concatenate([b for b, _ in segs]).astype(np.float32)
    scores = np.concatenate([s for _, s in segs]).astype(np.float32)
    offs = np.cumsum([0] + [len(b) for b, _ in segs]).astype(np.int32)
    mask, num = _C.nms_batched_mask(torch.from_numpy(boxes).to(_dev()), torch.from_numpy(scores).to(_dev()),
                                    torch.from_numpy(offs).to(_dev()), 12000, 0.7)
    mask, num = mask.cpu().numpy(), num.cpu().numpy()
    for i, (b, s) in enumerate(segs):
        keep = oracle.nms(b, s, 0.7)
        want = np.zeros(len(b), bool)
        want[keep] = True
        assert np.array_equal(mask[offs[i]:offs[i + 1]], want) and num[i] == len(keep)


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


def test_train_step_r101_dcn_fp16_cfg5():
    """BASELINE configs[4]: e2e_mask_rcnn_R_101_FPN_1x + deformable conv in C3-C5 + fp16 mixed precision
    (GradScaler path): training steps are finite and the scaler did not have to skip them all."""
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches
    cfg = load_cfg("e2e_mask_rcnn_R_101_FPN_1x.yaml",
                   ["MODEL.RESNETS.STAGE_WITH_DCN", "(False, True, True, True)", "DTYPE", "float16",
                    "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 300, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64])
    torch.manual_seed(0)
    model, opt, sched, step = build_training(cfg, _dev())
    assert step.scaler is not None
    n_dcn = sum(1 for m in model.modules() if type(m).__name__ in ("DFConv2d", "DeformConv", "ModulatedDeformConv"))
    assert n_dcn >= 30, "R-101 with DCN in C3-C5 has 4 + 23 + 3 deformable 3x3 convs (got %d modules)" % n_dcn
    (images, targets), = make_device_batches(cfg, _dev(), images_per_gpu=1, num_batches=1, height=192, width=256)
    before = [p.detach().clone() for p in list(model.parameters())[:8] if p.requires_grad]
    for _ in range(3):
        losses = step(images, targets)
    assert all(torch.isfinite(v) for v in losses.values())
    assert step.scaler.get_scale() > 0


def test_bucket_pack_and_flat_sgd_kernels_equal_torch_sgd_on_the_device():
    """csrc/optim.hip on the device: pack of ragged gradient tensors into 64-float slots (more than one launch: 48 tensors
    each) and three steps of the flat SGD pass against torch.optim.SGD with the reference's two parameter groups"""
    from maskrcnn_benchmark import _C
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    sizes = [7 * 7 * 256 * 64, 64, 1, 4099, 256, 1 << 20, 2, 300] + [33 + 17 * i for i in range(60)]
    kinds = [i % 3 == 1 for i in range(len(sizes))]                       # True = bias group
    order = [i for i in range(len(sizes)) if not kinds[i]] + [i for i in range(len(sizes)) if kinds[i]]
    offs, at = {}, 0
    for i in order:
        offs[i] = at
        at += -(-sizes[i] // 64) * 64
    split = offs[[i for i in order if kinds[i]][0]]
    params = [torch.randn(s, generator=g).to(dev) for s in sizes]
    P, M = torch.zeros(at, device=dev), torch.zeros(at, device=dev)
    _C.pack_into(P, [params[i] for i in order], [offs[i] for i in order])
    for i in order:
        assert torch.equal(P[offs[i]:offs[i] + sizes[i]], params[i])
    tp = [torch.nn.Parameter(p.clone()) for p in params]
    lr, wd, mom = 0.02, 1e-4, 0.9
    opt = torch.optim.SGD([{"params": [tp[i] for i in range(len(sizes)) if not kinds[i]], "lr": lr, "weight_decay": wd},
                           {"params": [tp[i] for i in range(len(sizes)) if kinds[i]], "lr": 2 * lr, "weight_decay": 0.0}],
                          lr=lr, momentum=mom)
    for step in range(3):
        grads = [torch.randn(s, generator=g).to(dev) for s in sizes]
        G = torch.full((at,), 7.0, device=dev)
        _C.pack_into(G, [grads[i] for i in order], [offs[i] for i in order])
        _C.sgd_momentum_flat_(P, G, M, split, lr, wd, 2 * lr, 0.0, mom)
        for t, gr in zip(tp, grads):
            t.grad = gr.clone()
        opt.step()
        for i in order:
            torch.testing.assert_close(P[offs[i]:offs[i] + sizes[i]], tp[i].detach(), rtol=2e-6, atol=2e-6)
            torch.testing.assert_close(M[offs[i]:offs[i] + sizes[i]], opt.state[tp[i]]["momentum_buffer"], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("comm", ["direct", "pg"])
def test_forced_ddp_hook_world1_nccl_matches_plain_step(comm, monkeypatch):
    """The overlapped-SGD DDP hook on the one MI355X we have: a 1-rank NCCL (= RCCL) process group wraps the
    detector, the bucket update runs on a side stream behind the bucket's all-reduce ("direct": RCCL called on the
    wrapper's own low-priority stream, engine/rccl_comm.py; "pg": in ProcessGroupNCCL's completion callback), and the
    parameters after 3 steps equal the non-DDP run (bit-for-bit when the backward itself is deterministic)."""
    import torch.distributed as dist
    monkeypatch.setenv("DETOPS_DDP_COMM", comm)
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches
    cfg = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml",
                   ["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 300, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 300,
                    "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64])
    (images, targets), = make_device_batches(cfg, _dev(), images_per_gpu=1, num_batches=1, height=192, width=256)

    # "direct" updates with the library's own flat SGD kernel (csrc/optim.hip): the same formula as torch._fused_sgd_ with
    # its roundings fused differently — a few ulps per step.  The detector is not continuous in its weights (top-k, NMS,
    # matcher thresholds), so ulps can become a different proposal set a step later: ONE step is compared there (same
    # forward, same backward, only the update differs); "pg" shares torch's update and is compared over three.
    n_steps = 1 if comm == "direct" else 3

    def run(force):
        torch.manual_seed(0)
        model, opt, sched, step = build_training(cfg, _dev(), force_ddp=force)
        torch.manual_seed(1)
        for _ in range(n_steps):
            step(images, targets)
        torch.cuda.synchronize()
        m = model.module if force else model
        return [p.detach().clone() for p in m.parameters() if p.requires_grad], opt, model

    det0 = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True   # MIOpen: reproducible solvers only, so a stream race cannot hide in noise
    plain_a, _, _ = run(False)
    plain_b, _, _ = run(False)
    deterministic = all(torch.equal(a, b) for a, b in zip(plain_a, plain_b))
    # MIOpen's weight-gradient kernels are not run-to-run reproducible: the plain-vs-plain spread is the yardstick
    noise = max(float((a - b).abs().max()) for a, b in zip(plain_a, plain_b))
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29517 + (comm == "pg")), rank=0, world_size=1)
    try:
        ddp_p, opt, model = run(True)
        from maskrcnn_benchmark.engine.ddp_step import BucketedDataParallel
        assert isinstance(model, BucketedDataParallel) and opt.deferred and len(model.buckets) > 1
        assert model.comm_mode == comm, (model.comm_mode, model.comm_note)
        side = getattr(opt, "last_update_stream", None)
        assert side is not None, "the hook's update callback never ran"
        assert side != torch.cuda.default_stream(_dev()).cuda_stream, "update kernels must run on a side stream"
    finally:
        dist.destroy_process_group()
        torch.backends.cudnn.deterministic = det0
    print("forced-DDP test: backward deterministic = %s, plain-vs-plain spread = %.3g" % (deterministic, noise))
    slack = 0.0 if comm == "pg" else 2e-6
    for a, b in zip(plain_a, ddp_p):
        if deterministic and comm == "pg":
            assert torch.equal(a, b)
        else:
            tol = 4 * noise + 1e-7 + slack * max(1.0, float(a.abs().max()))
            assert float((a - b).abs().max()) <= tol, (float((a - b).abs().max()), noise)
    if comm == "direct":
        assert model._native_update and "native" in model.comm_note


def test_overlapped_sgd_fused_kernel_equals_torch_sgd_on_device():
    """the one-pass multi-tensor update (torch._fused_sgd_) used on CUDA parameters follows torch.optim.SGD
    (momentum 0.9, the reference's weight / bias param-group rule) step for step, first step included"""
    from maskrcnn_benchmark.engine import ddp_step
    assert ddp_step._FUSED_SGD

    class Cfg:
        class SOLVER:
            BASE_LR, MOMENTUM, WEIGHT_DECAY, BIAS_LR_FACTOR, WEIGHT_DECAY_BIAS = 0.05, 0.9, 0.01, 2, 0.0

    def toy():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1)).to(DEV)

    m1, m2 = toy(), toy()
    o1 = ddp_step.make_overlapped_sgd(Cfg, m1)
    w = [p for n, p in m2.named_parameters() if "bias" not in n]
    b = [p for n, p in m2.named_parameters() if "bias" in n]
    o2 = torch.optim.SGD([{"params": w, "lr": 0.05, "weight_decay": 0.01}, {"params": b, "lr": 0.1, "weight_decay": 0.0}],
                         lr=0.05, momentum=0.9)
    g = torch.Generator(device="cpu").manual_seed(0)
    for it in range(5):
        x = torch.randn(2, 3, 9, 11, generator=g).to(DEV)
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            (m(x) ** 2).mean().backward()
            o.step()
    for a, c in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(a, c, rtol=1e-6, atol=1e-7)


def test_overlapped_sgd_under_grad_scaler_skips_and_unscales_on_device():
    """fp16 path: torch.amp.GradScaler hands `grad_scale` / `found_inf` to the optimizer (`_step_supports_amp_scaling`)
    and never reads them back.  A step with non-finite gradients must leave parameters AND momentum state untouched —
    in particular the FIRST step (loss scale 65536: the first iterations of a real run overflow) — and the next finite
    step must equal torch.optim.SGD's step on the unscaled gradients."""
    from maskrcnn_benchmark.engine import ddp_step

    class Cfg:
        class SOLVER:
            BASE_LR, MOMENTUM, WEIGHT_DECAY, BIAS_LR_FACTOR, WEIGHT_DECAY_BIAS = 0.05, 0.9, 0.01, 2, 0.0

    def toy():
        torch.manual_seed(5)
        return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1)).to(DEV)

    m1, m2 = toy(), toy()
    o1 = ddp_step.make_overlapped_sgd(Cfg, m1)
    assert o1._step_supports_amp_scaling
    w = [p for n, p in m2.named_parameters() if "bias" not in n]
    b = [p for n, p in m2.named_parameters() if "bias" in n]
    o2 = torch.optim.SGD([{"params": w, "lr": 0.05, "weight_decay": 0.01}, {"params": b, "lr": 0.1, "weight_decay": 0.0}],
                         lr=0.05, momentum=0.9)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    g = torch.Generator(device="cpu").manual_seed(1)
    before = [p.detach().clone() for p in m1.parameters()]
    for it in range(6):
        x = torch.randn(2, 3, 9, 11, generator=g).to(DEV)
        poison = it in (0, 1, 3)                     # overflowing iterations, the first two included
        o1.zero_grad()
        loss = (m1(x) ** 2).mean()
        if poison:
            loss = loss * float("inf")
        scaler.scale(loss).backward()
        scaler.step(o1)
        scaler.update()
        if poison:
            if it < 2:
                for a, c in zip(m1.parameters(), before):
                    assert torch.equal(a, c), "a skipped step changed a parameter"
            continue
        o2.zero_grad()
        (m2(x) ** 2).mean().backward()
        o2.step()
    assert all(torch.isfinite(p).all() for p in m1.parameters())
    assert scaler.get_scale() < 1024.0
    for a, c in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(a, c, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("launcher", ["plain", "torchrun"])
def test_train_net_entry_script_runs_and_checkpoints(launcher, tmp_path):
    """tools/train_net.py — the reference's entry point (tools/train_net.py:133-197), same command line — executed as a
    subprocess for 3 iterations, alone and under `python -m torch.distributed.run` (one rank: the launcher's
    environment, LOCAL_RANK from torchrun): losses finite in the log, periodic + final checkpoints written, and a
    second invocation resumes from the last checkpoint instead of starting over."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "run")
    opts = ["--config-file", "e2e_mask_rcnn_R_50_FPN_1x.yaml", "SOLVER.MAX_ITER", "3", "SOLVER.IMS_PER_BATCH", "2",
            "SOLVER.BASE_LR", "0.0025", "SOLVER.CHECKPOINT_PERIOD", "2", "OUTPUT_DIR", out,
            "INPUT.MIN_SIZE_TRAIN", "(256,)", "INPUT.MAX_SIZE_TRAIN", "320",
            "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", "500", "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", "500"]
    cmd = [sys.executable]
    if launcher == "torchrun":
        port = 29600 + os.getpid() % 300
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(root, "tools", "train_net.py")] + opts
    env = dict(os.environ, MIOPEN_LOG_LEVEL="1")
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    m = re.search(r"iter: 3 .*?\bloss: ([0-9.eE+-]+|nan|inf)", log)
    assert m, log[-3000:]
    assert np.isfinite(float(m.group(1))), m.group(0)
    assert os.path.exists(os.path.join(out, "model_0000002.pth")) and os.path.exists(os.path.join(out, "model_final.pth"))
    assert open(os.path.join(out, "last_checkpoint")).read().strip().endswith("model_final.pth")
    # resume: the checkpointer picks up iteration 3 of 3 -> nothing left to train
    r2 = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, (r2.stdout + r2.stderr)[-3000:]
    assert "Loading checkpoint from" in (r2.stdout + r2.stderr)


@pytest.mark.gpu
def test_pooler_backward_prepared_at_forward_time_equals_one_call_backward(monkeypatch):
    """The fused multi-level ROIAlign CAN issue its backward's pre-pass at FORWARD time on a side stream (modeling/poolers.py:
    PREPARE_BACKWARD_AT_FORWARD) and launch only the main kernel in the backward pass: same gradients, bit for bit, as the one-call backward, also
    when several poolers are in flight before any backward runs (box head + mask head) and across iterations."""
    import synth
    from maskrcnn_benchmark import _C, _lib
    from maskrcnn_benchmark.modeling import poolers
    from maskrcnn_benchmark.modeling.poolers import roi_align_fpn
    monkeypatch.setattr(poolers, "PREPARE_BACKWARD_AT_FORWARD", True)     # off by default (it costs a second hardware queue)
    dev = torch.device("cuda")
    torch.manual_seed(3)
    feats = [torch.randn(2, 64, h, w, device=dev, requires_grad=True) for (h, w) in synth.fpn_shapes()[:4]]
    scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
    rois_box = torch.from_numpy(synth.fpn_rois(per_image=512)).to(dev)
    rois_mask = torch.from_numpy(synth.fpn_rois(per_image=128, seed=8)).to(dev)
    for it in range(3):
        out_b = roi_align_fpn(feats, rois_box, (7, 7), scales, 2, 2, 5)
        out_m = roi_align_fpn(feats, rois_mask, (14, 14), scales, 2, 2, 5)
        assert out_b.grad_fn.prepared is not None and out_m.grad_fn.prepared is not None, "the prepared path did not engage"
        gb, gm = torch.randn_like(out_b), torch.randn_like(out_m)
        grads = torch.autograd.grad([out_b, out_m], feats, [gb, gm])
        lv_b = _C.roi_align_fpn_forward([f.detach() for f in feats], rois_box, scales, 7, 7, 2, 2, 5)[1]
        lv_m = _C.roi_align_fpn_forward([f.detach() for f in feats], rois_mask, scales, 14, 14, 2, 2, 5)[1]
        shapes = [tuple(f.shape) for f in feats]
        ref_b = _C.roi_align_fpn_backward(gb, rois_box, lv_b, shapes, scales, 7, 7, 2)
        ref_m = _C.roi_align_fpn_backward(gm, rois_mask, lv_m, shapes, scales, 14, 14, 2)
        for g, rb, rm in zip(grads, ref_b, ref_m):
            assert torch.equal(g, rb + rm)
    # a shape the ring plan does not serve (scan kernel forced): prepare answers "unsupported", the one-call path runs
    _lib.tuning_set("roi_bwd_impl", 2)
    out = roi_align_fpn(feats, rois_box, (7, 7), scales, 2, 2, 5)
    assert out.grad_fn.prepared is None
    torch.autograd.grad(out, feats, torch.ones_like(out))


# ------------------------------------------------------------------ eval post-processing on the device
def _canon_det(r):
    """BoxList -> (boxes, scores, labels) in a canonical order (label, then score descending, then box)"""
    b = r.bbox.detach().cpu().numpy()
    s = r.get_field("scores").detach().cpu().numpy()
    l = r.get_field("labels").detach().cpu().numpy()
    order = np.lexsort((b[:, 3], b[:, 2], b[:, 1], b[:, 0], -s, l))
    return b[order], s[order], l[order]


@pytest.mark.parametrize("tag", ["a", "b"])
def test_box_head_postprocessor_on_the_device_equals_the_reference_fixture(tag):
    """PostProcessor.forward / filter_results (reference roi_heads/box_head/inference.py:42-149) ON THE MI355X — per-class NMS
    as one segmented launch of the HIP kernel — against tests/golden/model_postprocess.npz, which the REFERENCE's own
    PostProcessor produced (tests/golden/make_golden_model.py): labels bit-exact, scores <= 1e-6."""
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.modeling.roi_heads.box_head.inference import PostProcessor
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_postprocess.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(_dev())  # noqa: E731
    thr, nms, dets = g["box_%s_cfg" % tag]
    pp = PostProcessor(float(thr), float(nms), int(dets), BoxCoder((10., 10., 5., 5.)))
    sizes = [tuple(int(v) for v in s) for s in g["box_sizes"]]
    props = [BoxList(t(g["box_props_%d" % i]), sizes[i], mode="xyxy") for i in range(2)]
    res = pp((t(g["box_logits"]), t(g["box_reg"])), props)
    assert all(r.bbox.is_cuda for r in res)
    for i, r in enumerate(res):
        b, s, l = _canon_det(r)
        wb, ws, wl = g["box_%s_boxes_%d" % (tag, i)], g["box_%s_scores_%d" % (tag, i)], g["box_%s_labels_%d" % (tag, i)]
        order = np.lexsort((wb[:, 3], wb[:, 2], wb[:, 1], wb[:, 0], -ws, wl))
        np.testing.assert_array_equal(l, wl[order])
        np.testing.assert_allclose(s, ws[order], rtol=0, atol=1e-6)
        np.testing.assert_allclose(b, wb[order], rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_retinanet_postprocessor_on_the_device_equals_the_reference_fixture(tag):
    """RetinaNetPostProcessor.forward (reference rpn/retinanet/inference.py:64-173) on the device against the fixture the
    reference's own module produced: labels bit-exact, scores <= 1e-6."""
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.modeling.rpn.anchor_generator import AnchorGenerator
    from maskrcnn_benchmark.modeling.rpn.retinanet.inference import RetinaNetPostProcessor
    from maskrcnn_benchmark.structures.image_list import ImageList
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_postprocess.npz"))
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    H, W = (int(v) for v in g["ret_canvas"])
    sz = tuple(tuple(s * 2 ** (k / 3.0) for k in range(3)) for s in (32, 64, 128))
    ag = AnchorGenerator(sizes=sz, aspect_ratios=(0.5, 1.0, 2.0), anchor_strides=(8, 16, 32), straddle_thresh=-1).to(dev)
    feats = [torch.zeros(2, 1, H // s, W // s, device=dev) for s in (8, 16, 32)]
    il = ImageList(torch.zeros(2, 3, H, W, device=dev), [tuple(int(v) for v in s) for s in g["ret_image_sizes"]])
    anchors = ag(il, feats)
    thr, topn, nms, post = g["ret_%s_cfg" % tag]
    pp = RetinaNetPostProcessor(float(thr), int(topn), float(nms), int(post), 0, 6, BoxCoder((10., 10., 5., 5.)))
    res = pp(anchors, [t(g["ret_cls_%d" % l]) for l in range(3)], [t(g["ret_reg_%d" % l]) for l in range(3)])
    for i, r in enumerate(res):
        b, s, l = _canon_det(r)
        wb, ws, wl = g["ret_%s_boxes_%d" % (tag, i)], g["ret_%s_scores_%d" % (tag, i)], g["ret_%s_labels_%d" % (tag, i)]
        order = np.lexsort((wb[:, 3], wb[:, 2], wb[:, 1], wb[:, 0], -ws, wl))
        np.testing.assert_array_equal(l, wl[order])
        np.testing.assert_allclose(s, ws[order], rtol=0, atol=1e-6)
        np.testing.assert_allclose(b, wb[order], rtol=1e-5, atol=1e-3)


def test_box_head_postprocessor_at_model_size_equals_per_class_oracle_nms():
    """filter_results at the model's own size (1000 proposals x 81 classes = 80 NMS problems in one segmented launch):
    the kept (proposal, class) INDEX SET equals a per-class loop over the CPU oracle's NMS (reference
    roi_heads/box_head/inference.py:118-137 with csrc/cpu/nms_cpu.cpp semantics) bit-exactly, scores <= 1e-6."""
    from maskrcnn_benchmark.modeling.box_coder import BoxCoder
    from maskrcnn_benchmark.modeling.roi_heads.box_head.inference import PostProcessor
    rng = np.random.RandomState(3)
    n, C, W, H = 1000, 81, 1333, 800
    ctr = rng.uniform([0, 0], [W, H], (n, 1, 2))
    wh = rng.uniform(8, 300, (n, C, 2))
    jit = rng.normal(0, 6, (n, C, 2))
    boxes = np.concatenate([ctr + jit - wh / 2, ctr + jit + wh / 2], axis=2).astype(np.float32)      # [n, C, 4]
    logits = rng.normal(0, 2.5, (n, C)).astype(np.float32)
    scores = np.exp(logits - logits.max(1, keepdims=True))
    scores = (scores / scores.sum(1, keepdims=True)).astype(np.float32)
    pp = PostProcessor(0.05, 0.5, 100, BoxCoder((10., 10., 5., 5.)))
    out = pp.filter_results(torch.from_numpy(boxes.reshape(n, C * 4)).to(_dev()), torch.from_numpy(scores).to(_dev()), (W, H))
    # reference loop on the host
    clipped = boxes.copy()
    clipped[..., 0::2] = np.clip(clipped[..., 0::2], 0, W - 1)
    clipped[..., 1::2] = np.clip(clipped[..., 1::2], 0, H - 1)
    det = []
    for c in range(1, C):
        idx = np.nonzero(scores[:, c] > 0.05)[0]
        if idx.size == 0:
            continue
        keep = oracle.nms(clipped[idx, c], scores[idx, c], 0.5)
        det += [(float(scores[idx[k], c]), c, int(idx[k])) for k in keep]
    det.sort(key=lambda d: -d[0])
    if len(det) > 100:
        thr = det[99][0]
        det = [d for d in det if d[0] >= thr]
    want = sorted((c, i) for _, c, i in det)
    got_l = out.get_field("labels").cpu().numpy()
    got_b = out.bbox.cpu().numpy()
    got_s = out.get_field("scores").cpu().numpy()
    assert len(got_l) == len(want)
    # recover the proposal index of every detection from its (unique) clipped box
    got = []
    for b, l, s in zip(got_b, got_l, got_s):
        i = np.nonzero((clipped[:, l] == b).all(1))[0]
        assert i.size >= 1
        got.append((int(l), int(i[0])))
        assert abs(float(scores[i[0], l]) - float(s)) <= 1e-6
    assert sorted(got) == want


@pytest.mark.parametrize("config", ["e2e_mask_rcnn_R_50_FPN_1x.yaml", "retinanet/retinanet_R-50-FPN_1x.yaml"])
def test_eval_forward_on_the_device_returns_finite_detections(config):
    """eval-mode forward of the whole detector on the device (backbone -> RPN test-time selection -> box head PostProcessor ->
    mask head post-processing / RetinaNetPostProcessor): per image a BoxList with scores, labels (and masks), all finite,
    scores sorted into (0, 1], boxes inside the image"""
    from maskrcnn_benchmark.engine.bench_step import load_cfg, make_device_batches
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = load_cfg(config, ["MODEL.ROI_HEADS.SCORE_THRESH", 0.0, "MODEL.RETINANET.INFERENCE_TH", 0.0])
    torch.manual_seed(0)
    model = build_detection_model(cfg).to(_dev()).eval()
    (images, _), = make_device_batches(cfg, _dev(), images_per_gpu=2, num_batches=1, height=256, width=320)
    with torch.no_grad():
        det = model(images)
    assert len(det) == 2
    for d in det:
        assert d.bbox.is_cuda and len(d) > 0
        s = d.get_field("scores")
        assert torch.isfinite(d.bbox).all() and torch.isfinite(s).all() and (s > 0).all() and (s <= 1).all()
        assert d.get_field("labels").min() >= 1
        W, H = d.size
        assert (d.bbox[:, 0::2] <= W - 1 + 1e-3).all() and (d.bbox[:, 1::2] <= H - 1 + 1e-3).all() and (d.bbox >= 0).all()
        if cfg.MODEL.MASK_ON:
            assert d.get_field("mask").shape[0] == len(d)


# ------------------------------------------------------------------ the training iteration replayed from a HIP graph
def _graph_env_ready():
    import os
    return os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"


@pytest.mark.skipif(not _graph_env_ready(), reason="HIP graph replay of a training step needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 "
                                                   "before the HIP runtime starts (tests/conftest.py sets it)")
def test_graphed_train_step_equals_the_eager_step_where_nothing_is_random():
    """engine/graph_step.py on RetinaNet (no sampler: the iteration is a deterministic function of weights and inputs): the
    losses of five iterations replayed from a HIP graph equal the eager TrainStep's from the same initial weights — through the
    scheduler's warm-up, i.e. with a learning rate that changes every iteration (it lives in a device tensor, not in the
    captured launch arguments) — to the run-to-run noise of MIOpen's atomic weight-gradient kernels."""
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches
    from maskrcnn_benchmark.engine.graph_step import GraphedTrainStep
    cfg = load_cfg("retinanet/retinanet_R-50-FPN_1x.yaml", ["SOLVER.BASE_LR", 0.002, "SOLVER.WARMUP_ITERS", 50])
    (images, targets), = make_device_batches(cfg, _dev(), images_per_gpu=1, num_batches=1, height=192, width=256)

    def run(graphed, n=8):
        torch.manual_seed(0)
        model, opt, sched, step = build_training(cfg, _dev())
        g = GraphedTrainStep(step, warmup=3) if graphed else None
        out = []
        for i in range(n):
            ld = (g if graphed else step)(images, targets)
            out.append({k: float(v.detach()) for k, v in ld.items()})
        torch.cuda.synchronize()
        return out, g, opt, sched

    eager, _, _, _ = run(False)
    graphed, g, opt, sched = run(True)
    assert g.replays == 8 - 0 and g.eager_steps == 3 and len(g._graphs) == 1
    # the graphed run did 3 eager warm-up iterations before its first replay: replay i corresponds to eager iteration i + 3
    for i in range(5):
        for k in eager[i + 3]:
            a, b = eager[i + 3][k], graphed[i][k]
            assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (i, k, a, b)
    # the device-side learning rate follows the scheduler
    want = [float(grp["lr"]) for grp in opt.param_groups]
    assert want[0] != cfg.SOLVER.BASE_LR                        # still warming up: it really changed every iteration
    for gi, v in enumerate(want):
        # the tensor holds the value pushed before the LAST replay; one scheduler step has happened since
        assert abs(float(opt.lr_tensors[gi]) - v) <= abs(v) * 0.2 + 1e-12


@pytest.mark.skipif(not _graph_env_ready(), reason="needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the HIP runtime starts")
def test_graphed_train_step_mask_rcnn_signatures_seeds_and_fallback():
    """Mask R-CNN through GraphedTrainStep: one graph per input signature (two batches with different ground-truth counts), a
    third signature beyond `max_graphs` runs eagerly; every iteration — replayed or not — draws a new sampler stream (the device
    seed word is incremented by the captured step itself); losses stay finite and the weights move."""
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches
    from maskrcnn_benchmark.engine.graph_step import GraphedTrainStep
    cfg = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml",
                   ["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 300, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 300,
                    "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64])
    batches = make_device_batches(cfg, _dev(), images_per_gpu=1, num_batches=3, height=192, width=256)
    counts = [len(t[0]) for _, t in batches]
    assert len(set(counts)) == 3, counts                         # three signatures
    torch.manual_seed(0)
    model, opt, sched, step = build_training(cfg, _dev())
    g = GraphedTrainStep(step, warmup=2, max_graphs=2)
    before = [p.detach().clone() for p in model.parameters() if p.requires_grad][:4]
    seen = []
    for i in range(9):
        ld = g(*batches[i % 3])
        vals = {k: float(v.detach()) for k, v in ld.items()}
        assert all(v == v and abs(v) != float("inf") for v in vals.values()), (i, vals)
        seen.append(int(g.seed))
    torch.cuda.synchronize()
    assert len(g._graphs) == 2 and g.replays == 6 and g.eager_steps == 2 + 3      # warm-up + the third signature's iterations
    assert seen == sorted(set(seen)) and len(seen) == 9                            # a new seed every iteration
    after = [p.detach() for p in model.parameters() if p.requires_grad][:4]
    assert any(not torch.equal(a, b) for a, b in zip(before, after))
    from maskrcnn_benchmark import _C
    assert _C.nms_repaired_segments(_dev()) == 0


# ------------------------------------------------------------------ mixed precision: one multi-tensor weight cast per step
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_half_weights_gradients_equal_the_per_layer_autocast_gradients(dtype):
    """layers/half_weights.py on the device: forward + backward under autocast with every weight's half copy (and every weight
    gradient's fp32 copy) made by one multi-tensor launch gives the losses and the fp32 gradients of autocast's per-layer casts
    — the same cast of the same masters, the same widened gradients.  Compared on RetinaNet, whose training forward has no
    sampler and no NMS (a deterministic function of weights and inputs up to MIOpen's atomic split-K sums; Mask R-CNN's
    proposal set flips with one half-precision ulp of a score and its head losses with it: measured 3 % between two identical
    fp16 runs), against the noise of the per-layer path run twice.  Then Mask R-CNN through TrainStep (GradScaler under fp16):
    finite losses, fp32 parameters, attributes restored.  The exact statement is the CPU test (tests/test_model_cpu.py)."""
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches
    amp = {"bfloat16": torch.bfloat16, "float16": torch.float16}[dtype]
    cfg = load_cfg("retinanet/retinanet_R-50-FPN_1x.yaml", ["DTYPE", dtype, "SOLVER.BASE_LR", 0.002])
    (images, targets), = make_device_batches(cfg, _dev(), images_per_gpu=1, num_batches=1, height=192, width=256)
    torch.manual_seed(0)
    model, opt, sched, step = build_training(cfg, _dev())

    def grads(enabled):
        model.half_weights.enabled = enabled
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=amp):
            ld = model(images, targets)
        (sum(ld.values()) * 64.0).backward()
        torch.cuda.synchronize()
        assert all("weight" not in m.__dict__ for m in model.modules())
        return ({k: float(v.detach()) for k, v in ld.items()},
                {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})

    la, ga = grads(False)
    assert model.half_weights.entries is None
    la2, ga2 = grads(False)
    lb, gb = grads(True)
    assert len(model.half_weights.entries) > 60
    assert ga.keys() == gb.keys()
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-3 * max(1.0, abs(la[k])) + 4 * abs(la[k] - la2[k]), (k, la[k], la2[k], lb[k])
    for n in ga:
        assert gb[n].dtype == torch.float32 and gb[n].shape == ga[n].shape and gb[n].stride() == ga[n].stride(), n
        ref = float(ga[n].double().norm())
        noise = float((ga[n].double() - ga2[n].double()).norm()) / max(ref, 1e-30)
        err = float((ga[n].double() - gb[n].double()).norm()) / max(ref, 1e-30)
        assert err <= 4 * noise + 2e-2, (n, err, noise)
    # Mask R-CNN: three training iterations with the single cast
    cfg = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml",
                   ["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 300, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 300,
                    "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64, "DTYPE", dtype, "SOLVER.BASE_LR", 0.002])
    (images, targets), = make_device_batches(cfg, _dev(), images_per_gpu=1, num_batches=1, height=192, width=256)
    torch.manual_seed(0)
    model, opt, sched, step = build_training(cfg, _dev())
    assert model.half_weights.enabled
    for _ in range(3):
        ld = step(images, targets)
    vals = {k: float(v.detach()) for k, v in ld.items()}
    assert all(v == v and abs(v) != float("inf") for v in vals.values()), vals
    assert len(model.half_weights.entries) > 60 and not model.half_weights.installed
    assert all(p.dtype == torch.float32 for p in model.parameters())
    assert all("weight" not in m.__dict__ for m in model.modules())


# ------------------------------------------------------------------ mask head on the positives only
def test_mask_head_dynamic_slots_equal_the_fixed_quota_on_the_device(monkeypatch):
    """roi_heads/mask_head/mask_head.py on the MI355X (fp32): the mask head on ceil16(positives) slots per image — the counts
    read back through the asynchronous copy the box head starts after its sampler — gives the losses and gradients of the
    fixed quota of 128 slots per image (the extra slots are masked out of the loss either way; the convolutions run other
    MIOpen kernels for the other batch size, hence a tolerance, not bit equality)."""
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches
    from maskrcnn_benchmark.modeling.roi_heads.mask_head import mask_head as MH
    cfg = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml", ["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 500, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 500])
    (images, targets), = make_device_batches(cfg, _dev(), images_per_gpu=2, num_batches=1, height=256, width=320)
    torch.manual_seed(0)
    model, opt, sched, step = build_training(cfg, _dev())

    def grads(mode):
        monkeypatch.setattr(MH, "SLOT_MODE", mode)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(7)
        ld = model(images, targets)
        sum(ld.values()).backward()
        torch.cuda.synchronize()
        return ({k: float(v.detach()) for k, v in ld.items()},
                {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                list(model.roi_heads.mask.last_slots))

    lf, gf, sf = grads("fixed")
    ld, gd, sd = grads("dynamic")
    assert sf == [128, 128] and all(s % MH.SLOT_GRANULE == 0 and MH.SLOT_GRANULE <= s <= 128 for s in sd), (sf, sd)
    for k in lf:
        assert abs(lf[k] - ld[k]) <= 2e-5 * max(1.0, abs(lf[k])), (k, lf[k], ld[k], sd)
    assert gf.keys() == gd.keys()
    for n in gf:
        ref = float(gf[n].double().norm())
        err = float((gf[n].double() - gd[n].double()).norm()) / max(ref, 1e-30)
        assert err <= 2e-3, (n, err)
    # and a training iteration in the default mode
    monkeypatch.setattr(MH, "SLOT_MODE", "dynamic")
    vals = {k: float(v.detach()) for k, v in step(images, targets).items()}
    assert all(v == v and abs(v) != float("inf") for v in vals.values()), vals
