"""fp32 islands under autocast — the torch.amp replacement for apex's `amp.float_function`
(reference layers/nms.py:8, roi_align.py:57, roi_pool.py:56)."""
import functools

import torch


def _to_float(x):
    if isinstance(x, torch.Tensor) and x.is_floating_point() and x.dtype != torch.float32:
        return x.float()
    return x


def float_function(fn):
    """Run `fn` with autocast disabled and every floating tensor argument cast to fp32."""

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        args = [_to_float(a) for a in args]
        kwargs = {k: _to_float(v) for k, v in kwargs.items()}
        with torch.autocast(device_type="cuda", enabled=False):
            return fn(*args, **kwargs)

    return wrapper
