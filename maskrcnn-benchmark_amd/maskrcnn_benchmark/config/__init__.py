"""`from maskrcnn_benchmark.config import cfg` (reference config/__init__.py:2)."""
from .defaults import _C as cfg
from .node import CfgNode

__all__ = ["cfg", "CfgNode"]
