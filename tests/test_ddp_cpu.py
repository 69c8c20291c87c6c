"""World-size-2 `gloo` tests of the data-parallel step (the N > 1 path of bench.py / train_net.py):
DDP gradient averaging with the SGD update chained to each bucket's all-reduce must produce the same
parameters as one process doing torch.optim.SGD on the mean of the two ranks' gradients, and
reduce_dict must average the logged losses on rank 0."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "maskrcnn-benchmark_amd")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(12, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                               torch.nn.Linear(64, 3))


def _cfg():
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    return load_cfg("", ["SOLVER.BASE_LR", 0.05, "SOLVER.WEIGHT_DECAY", 0.01, "SOLVER.MOMENTUM", 0.9])


def _data(rank, it):
    g = torch.Generator().manual_seed(100 * it + rank)
    return torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)


def _worker(rank, world, port, steps, out_dir, bucket_mb):
    sys.path.insert(0, PKG)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maskrcnn_benchmark.engine.ddp_step import BucketedDataParallel, make_overlapped_sgd, wrap_data_parallel
    from maskrcnn_benchmark.utils.comm import reduce_dict
    model = _toy()
    if rank:
        with torch.no_grad():           # the wrapper must hand every rank rank 0's weights
            for p in model.parameters():
                p.add_(1.0)
    opt = make_overlapped_sgd(_cfg(), model)
    ddp = wrap_data_parallel(model, opt, device_ids=None, bucket_cap_mb=bucket_mb)
    assert opt.deferred and isinstance(ddp, BucketedDataParallel)
    n_params = len(list(model.parameters()))
    assert (len(ddp.buckets) == 1) if bucket_mb > 1 else (2 < len(ddp.buckets) <= n_params)
    reduced = None
    for it in range(steps):
        x, y = _data(rank, it)
        # one rank leaves a parameter without gradient once: its bucket is flushed at the end of backward with a
        # zero slice (torch DDP would raise); the reference run gives that rank a zero gradient too
        frozen = it == 2 and rank == 1
        model[0].weight.requires_grad_(not frozen)
        loss = ((ddp(x) - y) ** 2).mean()
        model[0].weight.requires_grad_(True)
        opt.zero_grad(set_to_none=True)     # what TrainStep does: gradients are stolen, then packed per bucket
        loss.backward()   # all-reduce + SGD update happen inside, bucket by bucket
        # after the step p.grad is the AVERAGED gradient (a view into the bucket), as under torch DDP
        assert all(p.grad is not None and p.grad.data_ptr() >= b.flat.data_ptr()
                   for b in ddp.buckets for p in b.params)
        opt.step()        # no-op in deferred mode
        reduced = reduce_dict({"loss": loss.detach(), "twice": 2 * loss.detach()})
    assert ddp._next == 0 and not ddp._armed and not ddp._futures
    torch.save({"params": [p.detach().clone() for p in model.parameters()],
                "reduced": {k: float(v) for k, v in reduced.items()}, "loss": float(loss)},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb", [25, 0.001])  # one bucket / one bucket per parameter
def test_overlapped_sgd_ddp_matches_single_process_sgd(tmp_path, bucket_mb):
    sys.path.insert(0, PKG)
    steps, world = 6, 2
    mp.spawn(_worker, args=(world, _free_port(), steps, str(tmp_path), bucket_mb), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b), "ranks diverged"
    # reference: one process, torch.optim.SGD with the reference's param-group rule, mean of the rank grads
    model = _toy()
    weights = [p for n, p in model.named_parameters() if "bias" not in n]
    biases = [p for n, p in model.named_parameters() if "bias" in n]
    opt = torch.optim.SGD([{"params": weights, "lr": 0.05, "weight_decay": 0.01},
                           {"params": biases, "lr": 0.05 * 2, "weight_decay": 0.0}], lr=0.05, momentum=0.9)
    params = list(model.parameters())
    for it in range(steps):
        mean = [torch.zeros_like(p) for p in params]
        for rank in range(world):
            x, y = _data(rank, it)
            grads = list(torch.autograd.grad(((model(x) - y) ** 2).mean(), params))
            if it == 2 and rank == 1:
                grads[0] = torch.zeros_like(grads[0])     # the gradient that rank did not produce (see _worker)
            for m, g in zip(mean, grads):
                m += g / world
        for p, m in zip(params, mean):
            p.grad = m
        opt.step()
    for a, b in zip(r0["params"], model.parameters()):
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)
    # logged losses: rank 0 holds the mean over ranks
    assert abs(r0["reduced"]["loss"] - 0.5 * (r0["loss"] + r1["loss"])) < 1e-6
    assert abs(r0["reduced"]["twice"] - (r0["loss"] + r1["loss"])) < 1e-6


def test_overlapped_sgd_single_process_equals_torch_sgd():
    sys.path.insert(0, PKG)
    from maskrcnn_benchmark.engine.ddp_step import make_overlapped_sgd
    m1, m2 = _toy(3), _toy(3)
    o1 = make_overlapped_sgd(_cfg(), m1)
    w = [p for n, p in m2.named_parameters() if "bias" not in n]
    b = [p for n, p in m2.named_parameters() if "bias" in n]
    o2 = torch.optim.SGD([{"params": w, "lr": 0.05, "weight_decay": 0.01}, {"params": b, "lr": 0.1, "weight_decay": 0.0}],
                         lr=0.05, momentum=0.9)
    for it in range(5):
        x, y = _data(0, it)
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            ((m(x) - y) ** 2).mean().backward()
            o.step()
    for a, c in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(a, c, rtol=1e-6, atol=1e-7)


def test_flat_bucket_parameters_and_native_kernels_equal_torch_sgd():
    """The native update path of BucketedDataParallel (direct RCCL mode on the GPU) without a GPU: the layout logic is
    device-agnostic, the two kernels of csrc/optim.hip run under the host emulation on the SAME memory (`.numpy()` of a
    CPU tensor).  Parameters and momentum buffers become views of flat arrays (weights first, biases behind `split`,
    64-float slots); three steps of pack + flat SGD per bucket equal torch.optim.SGD with the reference's two parameter
    groups; `state_dict()` is unchanged in keys and values; an `optimizer.load_state_dict()` (which replaces the state
    tensors) is adopted back into the flat momentum array."""
    sys.path.insert(0, PKG)
    sys.path.insert(0, HERE)
    import emu
    from maskrcnn_benchmark.engine.ddp_step import BucketedDataParallel, make_overlapped_sgd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        m1, m2 = _toy(3), _toy(3)
        o1 = make_overlapped_sgd(_cfg(), m1)
        keys_before = list(m1.state_dict().keys())
        ddp = BucketedDataParallel(m1, o1, bucket_cap_mb=0.004)
        assert 2 <= len(ddp.buckets) and ddp.comm_mode == "pg" and not ddp._native_update
        ddp._make_flat_parameters()
        assert ddp._native_update
        for h in ddp._hooks:      # the test drives the buckets by hand
            h.remove()
        for b in ddp.buckets:
            assert b.split % 64 == 0 and all(o % 64 == 0 for o in b.offsets)
            for p_, o in zip(b.params, b.offsets):
                assert p_.data_ptr() == b.flat_p.data_ptr() + 4 * o
                assert o1.state[p_]["momentum_buffer"].data_ptr() == b.flat_m.data_ptr() + 4 * o
                assert (o1._group_of[p_] != 0) == (o >= b.split)
        assert list(m1.state_dict().keys()) == keys_before
        for a, c in zip(m1.state_dict().values(), m2.state_dict().values()):
            assert torch.equal(a, c)
        w = [p_ for n, p_ in m2.named_parameters() if "bias" not in n]
        bs = [p_ for n, p_ in m2.named_parameters() if "bias" in n]
        o2 = torch.optim.SGD([{"params": w, "lr": 0.05, "weight_decay": 0.01}, {"params": bs, "lr": 0.1, "weight_decay": 0.0}],
                             lr=0.05, momentum=0.9)

        def native_step():
            gw, gb = o1.param_groups[0], o1.param_groups[-1]
            for b in ddp.buckets:
                assert emu.pack(b.flat.numpy(), [p_.grad.numpy() for p_ in b.params], b.offsets) == 0
                assert emu.sgd_momentum_flat(b.flat_p.numpy(), b.flat.numpy(), b.flat_m.numpy(), b.split, gw["lr"],
                                             gw["weight_decay"], gb["lr"], gb["weight_decay"], gw["momentum"]) == 0
        for it in range(5):
            if it == 3:           # a checkpoint resume replaces the optimizer's state tensors
                o1.load_state_dict(__import__("copy").deepcopy(o1.state_dict()))
                b0 = ddp.buckets[0]
                assert o1.state[b0.params[0]]["momentum_buffer"].data_ptr() != b0.flat_m.data_ptr() + 4 * b0.offsets[0]
                for b in ddp.buckets:
                    ddp._adopt_momentum(b)
            x, y = _data(0, it)
            m1.zero_grad(set_to_none=True)
            ((m1(x) - y) ** 2).mean().backward()
            native_step()
            o2.zero_grad()
            ((m2(x) - y) ** 2).mean().backward()
            o2.step()
        for a, c in zip(m1.parameters(), m2.parameters()):
            torch.testing.assert_close(a, c, rtol=2e-6, atol=2e-7)
        for a, c in zip(m1.parameters(), m2.parameters()):
            torch.testing.assert_close(o1.state[a]["momentum_buffer"], o2.state[c]["momentum_buffer"], rtol=2e-6, atol=2e-7)
    finally:
        dist.destroy_process_group()


def _recovery_worker(rank, world, port, out_dir):
    sys.path.insert(0, PKG)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maskrcnn_benchmark.engine.ddp_step import make_overlapped_sgd, wrap_data_parallel
    model = _toy()
    opt = make_overlapped_sgd(_cfg(), model)
    ddp = wrap_data_parallel(model, opt, bucket_cap_mb=0.001)

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    for it in range(4):
        x, y = _data(rank, it)
        out = ddp(x)
        opt.zero_grad(set_to_none=True)
        if it == 1:
            # a backward pass that dies half-way: the last layer's buckets have been launched (their collectives match
            # on both ranks), the rest never arrive and the end-of-backward callback does not run
            h = model[2](torch.relu(model[0](x)))
            bad = model[4](torch.relu(Boom.apply(h)))
            with pytest.raises(RuntimeError, match="boom"):
                ((bad - y) ** 2).mean().backward()
            assert ddp._armed
            continue
        ((out - y) ** 2).mean().backward()
        assert not ddp._armed and ddp._next == 0 and not ddp._futures
    # eval-mode forward and the state dict go through the wrapper untouched
    ddp.eval()
    with torch.no_grad():
        assert ddp(_data(rank, 0)[0]).shape == (8, 3)
    assert all(k.startswith("module.") for k in ddp.state_dict())
    torch.save([p.detach().clone() for p in model.parameters()], os.path.join(out_dir, "rec_rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_data_parallel_recovers_from_a_failed_backward(tmp_path):
    """A backward pass that raises leaves buckets half-launched; the next forward starts clean, training continues and
    both ranks still hold identical parameters."""
    mp.spawn(_recovery_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "rec_rank0.pt"), torch.load(tmp_path / "rec_rank1.pt")
    for x, y in zip(a, b):
        assert torch.equal(x, y)
        assert torch.isfinite(x).all()


# ------------------------------------------------------------------ the detector itself under the DDP hook
def _detector_worker(rank, world, port, config, out_dir, dtype="float32", half=None, tag="det"):
    """What bench.py / train_net.py do at N > 1, on the CPU shim: build_training(distributed=True) wraps
    the detector in BucketedDataParallel with the overlapped SGD; three iterations must run and leave both ranks
    with identical weights."""
    sys.path.insert(0, PKG)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_shim
    import maskrcnn_benchmark.layers.sigmoid_focal_loss as sfl
    from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
    from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg
    cfg = load_cfg(config, ["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150,
                            "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32, "MODEL.RESNETS.RES2_OUT_CHANNELS", 16,
                            "MODEL.RESNETS.WIDTH_PER_GROUP", 4, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16,
                            "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32, "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16),
                            "SOLVER.BASE_LR", 0.002, "DTYPE", dtype])
    torch.manual_seed(7 + rank)  # different initial weights per rank: the wrapper must broadcast rank 0's
    model, opt, sched, step = build_training(cfg, torch.device("cpu"), distributed=True, local_rank=rank)
    from maskrcnn_benchmark.engine.ddp_step import BucketedDataParallel
    assert isinstance(model, BucketedDataParallel) and opt.deferred
    if half is not None:
        model.module.half_weights.enabled = half
    ds = SyntheticCOCODataset(length=2, height=96, width=128, with_masks=cfg.MODEL.MASK_ON, min_objects=2,
                              max_objects=4, seed=rank)
    images, targets, _ = BatchCollator(32)([ds[0], ds[1]])
    sfl.SigmoidFocalLoss.forward = lambda self, l, t: sfl.sigmoid_focal_loss_sum(l.float(), t, self.gamma, self.alpha)
    with cpu_shim.install():
        for _ in range(3):
            losses = step(images, list(targets))
    vals = {k: float(v.detach()) for k, v in losses.items()}
    used = model.module.half_weights.entries is not None
    torch.save({"params": [p.detach().clone() for p in model.module.parameters()], "losses": vals, "half_weights_used": used},
               os.path.join(out_dir, "%s_rank%d.pt" % (tag, rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_detector_bf16_per_stage_half_weights_under_ddp_world2(tmp_path):
    """mixed precision under the data-parallel wrapper (CPU autocast, gloo, world 2): the per-stage half-weight casts of
    layers/half_weights.py deliver the fp32 weight gradients to the bucket hooks stage by stage — three iterations leave both ranks
    with identical weights, equal to the run with autocast's per-layer casts"""
    runs = {}
    for half in (True, False):
        tag = "half%d" % int(half)
        mp.spawn(_detector_worker, args=(2, _free_port(), "e2e_mask_rcnn_R_50_FPN_1x.yaml", str(tmp_path), "bfloat16", half, tag),
                 nprocs=2, join=True)
        r0 = torch.load(os.path.join(str(tmp_path), "%s_rank0.pt" % tag))
        r1 = torch.load(os.path.join(str(tmp_path), "%s_rank1.pt" % tag))
        assert r0["half_weights_used"] == half
        assert all(v == v and abs(v) != float("inf") for v in r0["losses"].values())
        for a, b in zip(r0["params"], r1["params"]):
            assert torch.equal(a, b)
        runs[half] = r0
    for a, b in zip(runs[True]["params"], runs[False]["params"]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("config", ["e2e_mask_rcnn_R_50_FPN_1x.yaml", "retinanet/retinanet_R-50-FPN_1x.yaml"])
def test_detector_trains_under_ddp_hook_world2(tmp_path, config):
    port = _free_port()
    mp.spawn(_detector_worker, args=(2, port, config, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "det_rank0.pt"))
    r1 = torch.load(os.path.join(str(tmp_path), "det_rank1.pt"))
    assert all(v == v and abs(v) != float("inf") for v in r0["losses"].values())
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b), "ranks diverged: the averaged update must be identical on both"


def _uid_worker(rank, world, port, out_dir):
    """the unique-id exchange of engine/rccl_comm.py between two real ranks (gloo), with a stand-in for librccl.so that
    records what each rank hands to ncclCommInitRank"""
    sys.path.insert(0, PKG)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes
    from maskrcnn_benchmark.engine import rccl_comm
    seen = {}

    class FakeLib(object):
        def ncclGetUniqueId(self, ref):
            raw = bytes([5, 0, 9, 0, 0, 200] + [(7 * i) % 251 for i in range(122)])      # NUL bytes early on, like a real id
            ctypes.memmove(ref, raw, 128)
            return 0

        def ncclCommInitRank(self, comm_ref, nranks, uid, r):
            seen["id"], seen["nranks"], seen["rank"] = bytes(uid), nranks, r
            return 0

        def ncclGetErrorString(self, rc):
            return b"fake"

    rccl_comm._load = lambda: FakeLib()
    import contextlib
    torch.cuda.device = lambda d: contextlib.nullcontext()         # no GPU in this test: the guard is a no-op
    comm = rccl_comm.RcclComm(torch.device("cpu"))
    assert (seen["nranks"], seen["rank"]) == (world, rank) and comm.world == world
    torch.save(seen["id"], os.path.join(out_dir, "uid_%d.pt" % rank))
    dist.destroy_process_group()


def test_rccl_unique_id_reaches_every_rank_whole(tmp_path):
    """the 128-byte ncclUniqueId of rank 0 must arrive on the other ranks byte for byte — NUL bytes included (a c_char array
    field read as a C string would stop at the first one)"""
    mp.spawn(_uid_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path), "uid_%d.pt" % r)) for r in (0, 1))
    assert len(a) == 128 and a == b and a[:6] == bytes([5, 0, 9, 0, 0, 200])


class _FakeStream(object):
    """stands in for a HIP stream on the CPU: work is executed at enqueue time, so every ordering primitive is a no-op"""
    cuda_stream = None

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass

    def synchronize(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _FakeEvent(object):
    def record(self, stream=None):
        pass


def _direct_worker(rank, world, port, steps, out_dir, bucket_mb):
    """BucketedDataParallel's DIRECT path (RCCL on one side stream, flat buckets, the library's pack + SGD kernels) between
    two real ranks without a GPU: the kernels run under the host emulation (cpu_shim "emu-lib" swaps the library handle
    under `_C`), librccl.so is a stand-in whose ncclAllReduce averages the buffer over gloo, streams / events are no-ops."""
    sys.path.insert(0, PKG)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes
    import numpy as np
    import cpu_shim
    from maskrcnn_benchmark.engine import ddp_step, rccl_comm
    from maskrcnn_benchmark.engine.ddp_step import BucketedDataParallel, make_overlapped_sgd
    calls = []

    class FakeLib(object):
        def ncclGetUniqueId(self, ref):
            ctypes.memmove(ref, bytes([3, 0, 1] + [9] * 125), 128)
            return 0

        def ncclCommInitRank(self, comm_ref, nranks, uid, r):
            assert bytes(uid)[:3] == bytes([3, 0, 1]) and (nranks, r) == (world, rank)
            return 0

        def ncclAllReduce(self, send, recv, count, dtype, op, comm, stream):
            assert send == recv and dtype == 7 and op == 4                      # in place, fp32, ncclAvg
            t = torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * count).from_address(recv)))
            dist.all_reduce(t)
            t /= world
            calls.append(count)
            return 0

        def ncclCommDestroy(self, comm):
            return 0

        def ncclGetErrorString(self, rc):
            return b"fake"

    import contextlib
    rccl_comm._load = lambda: FakeLib()
    torch.cuda.device = lambda d: contextlib.nullcontext()
    torch.cuda.Stream = lambda *a, **k: _FakeStream()
    torch.cuda.Event = lambda *a, **k: _FakeEvent()
    torch.cuda.current_stream = lambda *a, **k: _FakeStream()
    dist_get_backend = dist.get_backend
    dist.get_backend = lambda *a, **k: "nccl"
    model = _toy()
    if rank:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    opt = make_overlapped_sgd(_cfg(), model)
    with cpu_shim.install("emu-lib"):
        ddp = BucketedDataParallel(model, opt, bucket_cap_mb=bucket_mb, comm="direct")
        assert ddp.comm_mode == "direct" and ddp._native_update and "native" in ddp.comm_note, (ddp.comm_mode, ddp.comm_note)
        for it in range(steps):
            x, y = _data(rank, it)
            opt.zero_grad(set_to_none=True)
            ((ddp(x) - y) ** 2).mean().backward()
            opt.step()                      # deferred: the buckets were updated inside backward
    dist.get_backend = dist_get_backend
    assert len(calls) == steps * len(ddp.buckets) + 1          # + the self-test at set-up
    torch.save({"params": [p.detach().clone() for p in model.parameters()], "buckets": len(ddp.buckets)},
               os.path.join(out_dir, "direct_%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb", [25, 0.002])
def test_direct_rccl_path_with_flat_buckets_world2_equals_single_process_sgd(tmp_path, bucket_mb):
    """the default N > 1 path of the GPU build, end to end between two ranks: identical parameters on both ranks, equal to
    single-process torch.optim.SGD on the MEAN gradient of the two ranks' batches (one bucket / many buckets)"""
    world, steps = 2, 4
    mp.spawn(_direct_worker, args=(world, _free_port(), steps, str(tmp_path), bucket_mb), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "direct_%d.pt" % r)) for r in (0, 1))
    assert (r0["buckets"] == 1) if bucket_mb > 1 else (r0["buckets"] > 2)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    sys.path.insert(0, PKG)
    ref = _toy()
    w = [p for n, p in ref.named_parameters() if "bias" not in n]
    bs = [p for n, p in ref.named_parameters() if "bias" in n]
    o = torch.optim.SGD([{"params": w, "lr": 0.05, "weight_decay": 0.01}, {"params": bs, "lr": 0.1, "weight_decay": 0.0}],
                        lr=0.05, momentum=0.9)
    for it in range(steps):
        o.zero_grad()
        loss = sum(((ref(x) - y) ** 2).mean() for x, y in (_data(r, it) for r in range(world))) / world
        loss.backward()
        o.step()
    for a, c in zip(r0["params"], ref.parameters()):
        torch.testing.assert_close(a, c.detach(), rtol=1e-5, atol=1e-6)
