"""Distributed helpers (reference utils/comm.py:13-117).  One process per GPU; the process group is
RCCL (backend "nccl" on ROCm) over xGMI on GPU nodes and gloo in the CPU tests."""
import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    """barrier across ranks (no-op for a single process)."""
    if get_world_size() > 1:
        dist.barrier()


def reduce_dict(input_dict, average=True):
    """Reduce a dict of scalar tensors to rank 0 (sum or mean) in ONE collective (reference
    :90-117, used for the logged losses, engine/trainer.py:18-40).  Other ranks get partial data."""
    world_size = get_world_size()
    if world_size < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().float().reshape(()) for k in names], dim=0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values = values / world_size
        return {k: v for k, v in zip(names, values)}


def all_gather_object(data):
    """list with every rank's picklable `data` (reference all_gather :51-87)."""
    world_size = get_world_size()
    if world_size == 1:
        return [data]
    out = [None] * world_size
    dist.all_gather_object(out, data)
    return out
