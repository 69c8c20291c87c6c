"""INTEGRATION.md's claim, executed: the REFERENCE's own operator wrappers (layers/roi_align.py, roi_pool.py, nms.py,
sigmoid_focal_loss.py, dcn/deform_conv_func.py, dcn/deform_pool_func.py — loaded from /root/reference, `apex` stubbed)
bind to THIS repository's `maskrcnn_benchmark._C` unchanged.

  * CPU branch (`nms`, `ROIAlign` forward on CPU tensors — the two operators the reference's `_C` serves on the CPU,
    csrc/nms.h:19-27, csrc/ROIAlign.h:19-24): run for real through the reference wrappers and compared with the
    vectors produced by the reference's own compiled CPU kernels (oracle/_ref -> tests/golden/ref_cpu_vectors.npz)
    and with the reference's own tests/test_nms.py answers.
  * every other operator is CUDA-only in the reference too: the wrappers are driven with tensors that claim
    `is_cuda`, and a recording proxy checks that each call the reference makes BINDS to this `_C`'s signature
    (positional count, order, keyword names) — the ABI half of "drop-in".  The CUDA-only message is checked as well.

Build container only (needs /root/reference; skipped on the GPU box, where the reference does not exist)."""
import importlib.util
import inspect
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference/maskrcnn_benchmark/layers"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")


class _Recorder(types.ModuleType):
    """Stands in for `maskrcnn_benchmark._C` while the reference wrappers run: CPU-tensor calls of the two CPU-branch
    operators go to the real module; everything else is bound against the real function's signature and answered
    with correctly shaped zeros (the values of the CUDA kernels are the GPU parity suite's business)."""

    def __init__(self, real):
        super().__init__("maskrcnn_benchmark._C")
        self.real, self.calls = real, []

    def __getattr__(self, name):
        fn = getattr(self.real, name)

        def call(*args, **kwargs):
            bound = inspect.signature(fn).bind(*args, **kwargs)   # TypeError = the reference's call does not bind
            self.calls.append((name, len(args), sorted(kwargs)))
            tensors = [a for a in args if isinstance(a, torch.Tensor)]
            if not any(isinstance(t, _FakeCuda) for t in tensors):
                return fn(*args, **kwargs)      # genuine CPU tensors: the real `_C` (CPU branch, or its refusal)
            a = bound.arguments
            plain = lambda t: t.as_subclass(torch.Tensor)  # noqa: E731
            if name == "roi_align_forward":
                return torch.zeros(a["rois"].size(0), a["input"].size(1), a["pooled_height"], a["pooled_width"])
            if name in ("roi_align_backward", "roi_pool_backward"):
                return torch.zeros(a["batch_size"], a["channels"], a["height"], a["width"])
            if name == "roi_pool_forward":
                z = torch.zeros(a["rois"].size(0), a["input"].size(1), a["pooled_height"], a["pooled_width"])
                return z, z.to(torch.int32)
            if name in ("sigmoid_focalloss_forward", "sigmoid_focalloss_backward"):
                return torch.zeros_like(plain(a["logits"]))
            return 0   # the deformable operators write into caller-allocated tensors
        return call


class _FakeCuda(torch.Tensor):
    """a CPU tensor that answers `is_cuda` with True: gets the reference's Python-side `if not input.is_cuda` checks
    out of the way so that the `_C` call itself is reached"""
    is_cuda = property(lambda self: True)


def _fake(*shape, dtype=torch.float32, requires_grad=False):
    t = torch.randn(*shape).to(dtype) if dtype.is_floating_point else torch.zeros(*shape, dtype=dtype)
    t = t.as_subclass(_FakeCuda)
    t.requires_grad_(requires_grad)
    return t


@pytest.fixture(scope="module")
def ref_layers():
    from maskrcnn_benchmark import _C as real
    rec = _Recorder(real)
    amp = types.ModuleType("apex.amp")
    amp.float_function = lambda f: f          # apex casts half inputs to float; irrelevant for fp32 inputs
    apex = types.ModuleType("apex")
    apex.amp = amp
    import maskrcnn_benchmark as pkg
    saved = {k: sys.modules.get(k) for k in ("apex", "apex.amp", "maskrcnn_benchmark._C")}
    saved_attr = pkg._C
    sys.modules.update({"apex": apex, "apex.amp": amp, "maskrcnn_benchmark._C": rec})
    pkg._C = rec
    mods = {}
    try:
        for name, rel in (("roi_align", "roi_align.py"), ("roi_pool", "roi_pool.py"), ("nms", "nms.py"),
                          ("sigmoid_focal_loss", "sigmoid_focal_loss.py"), ("deform_conv_func", "dcn/deform_conv_func.py"),
                          ("deform_pool_func", "dcn/deform_pool_func.py")):
            spec = importlib.util.spec_from_file_location("reference_layers_" + name, os.path.join(REF, rel))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)       # `from maskrcnn_benchmark import _C` resolves to the recorder
            assert mod._C is rec
            mods[name] = mod
        yield types.SimpleNamespace(rec=rec, real=real, **mods)
    finally:
        pkg._C = saved_attr
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_reference_nms_wrapper_runs_on_this_C(ref_layers, golden_dir):
    g = np.load(os.path.join(golden_dir, "nms_reference_tests.npz"))     # the reference's tests/test_nms.py cases
    for i in range(int(g["num_cases"])):
        keep = ref_layers.nms.nms(torch.from_numpy(g[f"boxes_{i}"]), torch.from_numpy(g[f"scores_{i}"]), float(g[f"thresh_{i}"]))
        np.testing.assert_array_equal(keep.numpy(), g[f"expected_{i}"])
    import synth
    v = np.load(os.path.join(golden_dir, "ref_cpu_vectors.npz"))         # outputs of the reference's compiled nms_cpu.cpp
    for key in v["nms_cases"]:
        _, n, uniform, seed, thr = str(key).split("_")
        b, s = synth.nms_boxes(int(n), seed=int(seed), uniform=bool(int(uniform)))
        keep = ref_layers.nms.nms(torch.from_numpy(b), torch.from_numpy(s), int(thr) / 100.0)
        np.testing.assert_array_equal(keep.numpy(), v[str(key)], err_msg=str(key))


def test_reference_roi_align_module_runs_on_this_C_cpu_branch(ref_layers, golden_dir):
    v = np.load(os.path.join(golden_dir, "ref_cpu_vectors.npz"))         # outputs of the reference's compiled ROIAlign_cpu.cpp
    for i in range(4):
        ph, pw, sr = [int(x) for x in v[f"ra_cfg_{i}"]]
        m = ref_layers.roi_align.ROIAlign((ph, pw), float(v["ra_scale"]), sr)
        out = m(torch.from_numpy(v["ra_input"]), torch.from_numpy(v["ra_rois"]))
        assert np.array_equal(out.numpy(), v[f"ra_out_{i}"])
        assert "output_size=(%d, %d)" % (ph, pw) in repr(m)


def test_reference_cuda_only_ops_refuse_cpu_tensors_like_the_reference(ref_layers):
    x = torch.randn(1, 4, 8, 8, requires_grad=True)
    rois = torch.tensor([[0, 0, 0, 4, 4]], dtype=torch.float32)
    out = ref_layers.roi_align.roi_align(x, rois, (2, 2), 1.0, 2)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):   # csrc/ROIAlign.h:44
        out.sum().backward()
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):   # csrc/ROIPool.h:21
        ref_layers.roi_pool.roi_pool(x, rois, (2, 2), 1.0)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):   # csrc/SigmoidFocalLoss.h:22
        ref_layers.real.sigmoid_focalloss_forward(torch.randn(4, 3), torch.zeros(4, dtype=torch.int32), 3, 2.0, 0.25)


def test_reference_wrappers_bind_every_cuda_only_call(ref_layers):
    rec = ref_layers.rec
    del rec.calls[:]

    def bw(out):   # the upstream gradient claims `is_cuda` too (deform_conv_func.py:78 checks it)
        out.backward(_fake(*out.shape))

    # ROIAlign / ROIPool autograd functions: forward + backward argument lists (layers/roi_align.py:20-44, roi_pool.py:18-43)
    x = _fake(2, 8, 16, 16, requires_grad=True)
    rois = _fake(5, 5)
    bw(ref_layers.roi_align.roi_align(x, rois, (7, 7), 0.25, 2))
    bw(ref_layers.roi_pool.roi_pool(x, rois, (7, 7), 0.25))
    # SigmoidFocalLoss (layers/sigmoid_focal_loss.py:18-35)
    logits = _fake(10, 80, requires_grad=True)
    targets = _fake(10, dtype=torch.int32)
    bw(ref_layers.sigmoid_focal_loss.sigmoid_focal_loss_cuda(logits, targets, 2.0, 0.25))
    # deformable conv v1 / v2 (layers/dcn/deform_conv_func.py:49-70, 87-127, 182-242)
    inp = _fake(2, 8, 12, 12, requires_grad=True)
    w = _fake(8, 8, 3, 3, requires_grad=True)
    off = _fake(2, 18, 12, 12, requires_grad=True)
    # (the reference's v1 backward returns 8 gradients for 9 inputs, deform_conv_func.py:128-129: torch >= 1.x refuses
    # that AFTER both `_C` calls have been made — which is all this test is about)
    with pytest.raises(RuntimeError, match="incorrect number of gradients"):
        bw(ref_layers.deform_conv_func.deform_conv(inp, off, w, 1, 1, 1, 1, 1, 64))
    msk = _fake(2, 9, 12, 12, requires_grad=True)
    bias = _fake(8, requires_grad=True)
    bw(ref_layers.deform_conv_func.modulated_deform_conv(inp, off, msk, w, bias, 1, 1, 1, 1, 1))
    # deformable PS-ROI pooling (layers/dcn/deform_pool_func.py:41-95)
    data = _fake(2, 8 * 9, 12, 12, requires_grad=True)
    trans = _fake(5, 2, 3, 3, requires_grad=True)
    bw(ref_layers.deform_pool_func.deform_roi_pooling(data, rois, trans, 0.25, 3, 8, False, 3, 3, 4, 0.1))
    seen = {c[0] for c in rec.calls}
    expected = {"roi_align_forward", "roi_align_backward", "roi_pool_forward", "roi_pool_backward",
                "sigmoid_focalloss_forward", "sigmoid_focalloss_backward", "deform_conv_forward",
                "deform_conv_backward_input", "deform_conv_backward_parameters", "modulated_deform_conv_forward",
                "modulated_deform_conv_backward", "deform_psroi_pooling_forward", "deform_psroi_pooling_backward"}
    assert expected <= seen, "not reached: %s" % sorted(expected - seen)


def test_C_exports_exactly_the_reference_names():
    """csrc/vision.cpp:9-25 — the 14 names of the reference's pybind module"""
    import re
    src = open("/root/reference/maskrcnn_benchmark/csrc/vision.cpp").read()
    names = set(re.findall(r'm\.def\("([a-z_]+)"', src))
    from maskrcnn_benchmark import _C
    assert len(names) == 14 and all(callable(getattr(_C, n)) for n in names)
