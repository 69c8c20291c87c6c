"""Small helpers (reference modeling/utils.py:9-16)."""
import torch


def cat(tensors, dim=0):
    """torch.cat that returns the single element untouched (no copy)."""
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


_DEVICE_CONSTANTS = {}


def device_constant(values, dtype, device):
    """A small constant tensor (nested lists / tuples of Python numbers) resident on `device`, uploaded ONCE per distinct
    value: the training iteration then contains no host-to-device copy for clip bounds, sentinels and the like — no
    per-step upload in eager mode, and nothing a captured HIP graph could replay from a freed host buffer."""
    def freeze(v):
        return tuple(freeze(x) for x in v) if isinstance(v, (list, tuple)) else float(v)
    key = (freeze(values), dtype, str(device))
    t = _DEVICE_CONSTANTS.get(key)
    if t is None:
        t = torch.tensor(values, dtype=dtype).to(device)
        _DEVICE_CONSTANTS[key] = t
    return t
