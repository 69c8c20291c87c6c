"""Generate tests/golden/whole_model_{mask_rcnn,retinanet}.npz by running the REFERENCE's own detector (build container only).

What the fixture pins (VERDICT r03 "missing" #2): the reference's `GeneralizedRCNN` (modeling/detector/generalized_rcnn.py:46-65)
— backbone + FPN + RPN + box head + mask head, or the RetinaNet flavour — is BUILT from /root/reference, a narrow random-init
R-50-FPN configuration, its `state_dict()` is saved together with one training batch (two images, boxes, labels, binary
masks) and the loss dict its forward pass returns.  tests/test_whole_model_parity.py then builds THIS repository's detector
from the same configuration, loads that state_dict with `strict=True` (identical key set = a reference checkpoint loads,
utils/model_serialization.py:10-71) and must reproduce every loss.

How the reference is made importable here (the reference tree is read-only, nothing is copied): `yacs` (absent) is replaced
by the small CfgNode below, `apex.amp.float_function` by the identity, `cv2` / `pycocotools` by empty modules (binary masks
only), `torch._six.PY3` by True, and `maskrcnn_benchmark._C` by oracle/_ref — the reference's OWN CPU kernels
(csrc/cpu/ROIAlign_cpu.cpp, nms_cpu.cpp) compiled in place by oracle/build_ref.py.  The forward pass of the losses needs
nothing else from `_C` (RetinaNet's SigmoidFocalLoss takes the reference's Python CPU composite,
layers/sigmoid_focal_loss.py:40-66).

Sampling is made deterministic the way the verdict prescribes: the samplers' quotas are at least as large as their candidate
sets (`randperm(numel)[:quota]` then selects every candidate), so no random stream enters the losses.

Run:  python tests/golden/make_golden_whole_model.py      (never runs on the GPU box)
"""
import copy
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
np.float = float  # noqa: removed from numpy >= 1.24, used at modeling/rpn/anchor_generator.py:229-238

sys.path.insert(0, ROOT)
import oracle  # noqa: E402

ref_C = oracle.ref()
assert ref_C is not None, "build oracle/_ref first (python -c 'from oracle import build_ref; build_ref.build()')"
sys.path.remove(ROOT)


class CfgNode(dict):
    """the subset of yacs.config.CfgNode the reference's config/defaults.py and a yaml merge need"""

    def __init__(self, init=None, new_allowed=False):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def defrost(self):
        pass

    def _merge(self, other):
        for k, v in other.items():
            if k not in self:
                raise KeyError("unknown config key %r" % (k,))
            if isinstance(self[k], CfgNode):
                self[k]._merge(v)
            else:
                if isinstance(v, str) and not isinstance(self[k], str):   # yacs: "(4, 8, 16)" in a yaml is a literal
                    import ast
                    v = ast.literal_eval(v)
                if isinstance(self[k], tuple) and isinstance(v, list):
                    v = tuple(v)
                if isinstance(self[k], float) and isinstance(v, int):
                    v = float(v)
                self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def merge_from_list(self, lst):
        for key, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(key)
            node[parts[-1]] = v


def install_reference():
    yacs = types.ModuleType("yacs")
    yacs_config = types.ModuleType("yacs.config")
    yacs_config.CfgNode = CfgNode
    yacs.config = yacs_config
    amp = types.ModuleType("apex.amp")
    amp.float_function = lambda f: f
    apex = types.ModuleType("apex")
    apex.amp = amp
    stubs = {"yacs": yacs, "yacs.config": yacs_config, "apex": apex, "apex.amp": amp}
    for name in ("cv2", "pycocotools", "pycocotools.mask"):
        stubs[name] = types.ModuleType(name)
    stubs["pycocotools"].mask = stubs["pycocotools.mask"]
    sys.modules.update(stubs)
    if not hasattr(torch, "_six"):
        torch._six = types.SimpleNamespace(PY3=True, string_classes=(str,))
        sys.modules["torch._six"] = torch._six
    sys.path.insert(0, REF)
    import maskrcnn_benchmark
    assert maskrcnn_benchmark.__file__.startswith(REF)
    c = types.ModuleType("maskrcnn_benchmark._C")        # the reference's own CPU kernels, compiled from its sources
    c.nms = ref_C.nms
    c.roi_align_forward = ref_C.roi_align_forward
    sys.modules["maskrcnn_benchmark._C"] = c
    maskrcnn_benchmark._C = c


install_reference()
from maskrcnn_benchmark.config import cfg as ref_cfg  # noqa: E402
from maskrcnn_benchmark.modeling.detector import build_detection_model  # noqa: E402
from maskrcnn_benchmark.structures.bounding_box import BoxList  # noqa: E402
from maskrcnn_benchmark.structures.image_list import to_image_list  # noqa: E402
from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask  # noqa: E402

# The narrow R-50-FPN of tests/test_model_cpu.py::test_tiny_model_trains_on_cpu_shim; quotas >= candidates everywhere.
COMMON = ["MODEL.DEVICE", "cpu", "MODEL.RESNETS.RES2_OUT_CHANNELS", 16, "MODEL.RESNETS.WIDTH_PER_GROUP", 4,
          "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16, "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32,
          "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16)]
CASES = {
    "mask_rcnn": ("e2e_mask_rcnn_R_50_FPN_1x.yaml",
                  ["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150,
                   "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 32768, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 512]),
    "retinanet": ("retinanet/retinanet_R-50-FPN_1x.yaml", []),
}
IMG_H, IMG_W = 96, 128


def make_batch(rng, with_masks):
    images = [torch.from_numpy(rng.randn(3, IMG_H, IMG_W).astype(np.float32)),
              torch.from_numpy(rng.randn(3, IMG_H - 10, IMG_W - 22).astype(np.float32))]
    out = {"image_0": images[0].numpy(), "image_1": images[1].numpy()}
    targets = []
    for i, im in enumerate(images):
        H, W = im.shape[-2:]
        n = 3 + i
        w = rng.uniform(14, 60, n)
        h = rng.uniform(14, 50, n)
        x1 = rng.uniform(0, W - 16, n)
        y1 = rng.uniform(0, H - 16, n)
        boxes = np.stack([x1, y1, np.minimum(x1 + w, W - 1), np.minimum(y1 + h, H - 1)], 1).astype(np.float32)
        labels = rng.randint(1, 81, n).astype(np.int64)
        t = BoxList(torch.from_numpy(boxes), (W, H), mode="xyxy")
        t.add_field("labels", torch.from_numpy(labels))
        out["boxes_%d" % i], out["labels_%d" % i] = boxes, labels
        if with_masks:
            yy, xx = np.mgrid[0:H, 0:W]
            m = np.zeros((n, H, W), np.uint8)
            for k, b in enumerate(boxes):     # filled ellipses inside the boxes
                cx, cy, rx, ry = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, (b[2] - b[0]) / 2 + 0.5, (b[3] - b[1]) / 2 + 0.5
                m[k] = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0
            t.add_field("masks", SegmentationMask(torch.from_numpy(m), (W, H), mode="mask"))
            out["masks_%d" % i] = m
        targets.append(t)
    return images, targets, out


def run(name):
    yaml_rel, extra = CASES[name]
    cfg = ref_cfg.clone()
    cfg.merge_from_file(os.path.join(REF, "configs", yaml_rel))
    cfg.merge_from_list(COMMON + extra)
    torch.manual_seed(7)
    model = build_detection_model(cfg).train()
    rng = np.random.RandomState(11)
    images, targets, arrays = make_batch(rng, cfg.MODEL.MASK_ON)
    il = to_image_list(images, cfg.DATALOADER.SIZE_DIVISIBILITY)
    with torch.no_grad():
        losses = model(il, targets)
    arrays.update({"loss__" + k: np.float64(float(v)) for k, v in losses.items()})
    sd = model.state_dict()
    arrays.update({"sd__" + k: v.detach().cpu().numpy() for k, v in sd.items()})
    arrays["opts"] = np.array(repr(COMMON + extra))
    arrays["yaml"] = np.array(yaml_rel)
    arrays["size_divisibility"] = np.int64(cfg.DATALOADER.SIZE_DIVISIBILITY)
    path = os.path.join(HERE, "whole_model_%s.npz" % name)
    np.savez_compressed(path, **arrays)
    print(name, {k: float(v) for k, v in losses.items()}, "state_dict keys:", len(sd),
          "params:", sum(v.numel() for v in sd.values()), "file MB: %.2f" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    for case in (sys.argv[1:] or list(CASES)):
        run(case)
