"""`nms(dets[n,4], scores[n], threshold) -> int64 kept indices` (reference layers/nms.py:3-8)."""
from maskrcnn_benchmark import _C

from ._amp import float_function

# Only valid with fp32 inputs - give AMP the hint (reference layers/nms.py:7-8)
nms = float_function(_C.nms)
