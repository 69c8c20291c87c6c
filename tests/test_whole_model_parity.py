"""Whole-model parity against the REFERENCE's detector (tests/golden/make_golden_whole_model.py ran the reference's own
`GeneralizedRCNN`, modeling/detector/generalized_rcnn.py:46-65, on the CPU with its own compiled `_C`):

  * checkpoint compatibility — this repository's detector, built from the same configuration, has EXACTLY the reference's
    `state_dict()` key set and shapes (what utils/model_serialization.py:10-71 matches on) and loads it with strict=True;
  * loss parity — with those weights, the same two-image batch and sampler quotas >= candidates (take-all: no random
    stream enters), every entry of the training loss dict agrees with the reference's within 1e-4 relative — on the CPU
    (HIP-only operators served by the oracle through tests/cpu_shim.py) and, in the `-m gpu` suite, on the device through the
    real HIP kernels (ROIAlign, NMS, target kernels, focal loss, fused FrozenBN).
"""
import ast
import contextlib
import os

import numpy as np
import pytest
import torch

import cpu_shim

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def _build(name, device):
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.image_list import to_image_list
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask
    g = np.load(os.path.join(GOLDEN, "whole_model_%s.npz" % name), allow_pickle=False)
    opts = list(ast.literal_eval(str(g["opts"])))
    opts[opts.index("MODEL.DEVICE") + 1] = device
    cfg = load_cfg(str(g["yaml"]), opts)
    model = build_detection_model(cfg)
    ref_sd = {k[len("sd__"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd__")}
    images, targets = [], []
    for i in range(2):
        im = torch.from_numpy(g["image_%d" % i])
        H, W = im.shape[-2:]
        t = BoxList(torch.from_numpy(g["boxes_%d" % i]), (W, H), mode="xyxy")
        t.add_field("labels", torch.from_numpy(g["labels_%d" % i]))
        if "masks_%d" % i in g.files:
            t.add_field("masks", SegmentationMask(torch.from_numpy(g["masks_%d" % i]), (W, H), mode="mask"))
        images.append(im)
        targets.append(t)
    il = to_image_list(images, int(g["size_divisibility"]))
    ref_losses = {k[len("loss__"):]: float(g[k]) for k in g.files if k.startswith("loss__")}
    return cfg, model, ref_sd, il, targets, ref_losses


@pytest.mark.parametrize("name", ["mask_rcnn", "retinanet"])
def test_state_dict_keys_equal_the_reference_and_load_strict(name):
    _, model, ref_sd, _, _, _ = _build(name, "cpu")
    mine = model.state_dict()
    assert set(mine) == set(ref_sd), {"missing here": sorted(set(ref_sd) - set(mine))[:10],
                                      "unknown to the reference": sorted(set(mine) - set(ref_sd))[:10]}
    assert list(mine) == list(ref_sd), "same registration order as the reference (checkpoint files list keys in it)"
    for k, v in ref_sd.items():
        assert tuple(mine[k].shape) == tuple(v.shape), k
        assert mine[k].dtype == v.dtype, k
    model.load_state_dict(ref_sd, strict=True)


@pytest.mark.parametrize("name", ["mask_rcnn", "retinanet"])
@pytest.mark.parametrize("dev", ["cpu", "cpu-device-branches", "cpu-product-wrappers", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_losses_equal_the_reference_with_its_weights(name, dev, monkeypatch):
    # "cpu-device-branches": CPU tensors, but the model takes the branches it takes on the GPU (fused labels / sampler /
    # sampled-slot targets / proposal decode / batched proposal hand-over), served by the HIP sources under the host emulation
    backend = "oracle"
    if dev == "cpu-device-branches":
        dev, backend = "cpu", "emu-device"
    if dev == "cpu-product-wrappers":
        # the product's own `_C` wrappers and autograd functions, every operator of the model included, over the
        # host-emulation build of the HIP sources (cpu_shim backend "emu-lib"): nothing of `_C` is replaced
        dev, backend = "cpu", "emu-lib"
    cfg, model, ref_sd, il, targets, ref_losses = _build(name, dev)
    model.load_state_dict(ref_sd, strict=True)
    model.to(dev).train()
    if dev == "cpu" and backend != "emu-lib":
        import maskrcnn_benchmark.layers.sigmoid_focal_loss as sfl
        monkeypatch.setattr(sfl.SigmoidFocalLoss, "forward",
                            lambda self, l, t: sfl.sigmoid_focal_loss_sum(l.float(), t, self.gamma, self.alpha))
    with (cpu_shim.install(backend) if dev == "cpu" else contextlib.nullcontext()):
        with torch.no_grad():
            losses = model(il.to(dev), [t.to(dev) for t in targets])
    got = {k: float(v) for k, v in losses.items()}
    assert set(got) == set(ref_losses)
    for k, ref in ref_losses.items():
        assert abs(got[k] - ref) <= TOL * max(1.0, abs(ref)), (k, got[k], ref, got, ref_losses)
