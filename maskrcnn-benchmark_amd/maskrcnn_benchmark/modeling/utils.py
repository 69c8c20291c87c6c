"""Small helpers (reference modeling/utils.py:9-16)."""
import torch


def cat(tensors, dim=0):
    """torch.cat that returns the single element untouched (no copy)."""
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)
