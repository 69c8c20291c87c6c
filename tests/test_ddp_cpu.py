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
    from maskrcnn_benchmark.engine.ddp_step import make_overlapped_sgd, wrap_data_parallel
    from maskrcnn_benchmark.utils.comm import reduce_dict
    model = _toy()
    opt = make_overlapped_sgd(_cfg(), model)
    ddp = wrap_data_parallel(model, opt, device_ids=None, bucket_cap_mb=bucket_mb)
    assert opt.deferred and isinstance(ddp, torch.nn.parallel.DistributedDataParallel)
    reduced = None
    for it in range(steps):
        x, y = _data(rank, it)
        loss = ((ddp(x) - y) ** 2).mean()
        opt.zero_grad(set_to_none=False)
        loss.backward()   # all-reduce + SGD update happen inside, bucket by bucket
        opt.step()        # no-op in deferred mode
        reduced = reduce_dict({"loss": loss.detach(), "twice": 2 * loss.detach()})
    torch.save({"params": [p.detach().clone() for p in model.parameters()],
                "reduced": {k: float(v) for k, v in reduced.items()}, "loss": float(loss)},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb", [25, 0.001])  # one bucket / one bucket per parameter
def test_overlapped_sgd_ddp_matches_single_process_sgd(tmp_path, bucket_mb):
    sys.path.insert(0, PKG)
    steps, world = 4, 2
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
    for it in range(steps):
        opt.zero_grad()
        total = 0
        for rank in range(world):
            x, y = _data(rank, it)
            total = total + ((model(x) - y) ** 2).mean() / world
        total.backward()
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
