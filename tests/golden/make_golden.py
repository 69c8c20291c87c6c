"""Generate tests/golden/*.npz from the REFERENCE itself (run in the build container only).

What is captured
  nms_reference_tests.npz   the reference's own known-answer NMS tests, /root/reference/tests/
                            test_nms.py:11-58 (5 boxes x 5 thresholds) and :60-217 (53 boxes ->
                            26 kept @0.5): the test module is EXECUTED with a recording stub for
                            `maskrcnn_benchmark.layers.nms`, so inputs and expected index lists are
                            the reference's, not re-typed.
  box_coder_reference_tests.npz  /root/reference/tests/test_box_coder.py:11-105 vectors, captured
                            the same way (inputs of BoxCoder.decode + the expected output).
  ref_cpu_vectors.npz       outputs of the reference's own CPU kernels compiled in place
                            (oracle/_ref, see oracle/build_ref.py): `roi_align_forward` and `nms`
                            on small seeded inputs.
  focal_python_composite.npz  outputs of the reference's Python CPU focal loss
                            (layers/sigmoid_focal_loss.py:40-50, extracted with `ast` and exec'd —
                            the package itself is not importable here: apex is absent).

/root/reference does not exist on the GPU box; the fixtures do.  Re-run with
    python tests/golden/make_golden.py
"""
import ast
import os
import sys
import types
import unittest

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import oracle  # noqa: E402
import synth  # noqa: E402
from oracle import build_ref  # noqa: E402


def capture_reference_nms_tests(ref_mod):
    calls = []  # (boxes, scores, thresh)
    expected = []

    def rec_nms(boxes, scores, thresh):
        calls.append((boxes.numpy().copy(), scores.numpy().copy(), float(thresh)))
        return ref_mod.nms(boxes, scores, thresh)

    stub_pkg = types.ModuleType("maskrcnn_benchmark")
    stub_layers = types.ModuleType("maskrcnn_benchmark.layers")
    stub_layers.nms = rec_nms
    stub_pkg.layers = stub_layers
    sys.modules["maskrcnn_benchmark"] = stub_pkg
    sys.modules["maskrcnn_benchmark.layers"] = stub_layers

    orig_assert = np.testing.assert_array_equal

    def rec_assert(actual, desired, *a, **k):
        expected.append(np.asarray(desired).astype(np.int64).copy())
        return orig_assert(actual, desired, *a, **k)  # the compiled reference must pass its own test

    np.testing.assert_array_equal = rec_assert
    try:
        src = open(os.path.join(REF, "tests", "test_nms.py")).read()
        mod = types.ModuleType("ref_test_nms")
        exec(compile(src, "ref_test_nms.py", "exec"), mod.__dict__)
        suite = unittest.defaultTestLoader.loadTestsFromTestCase(mod.TestNMS)
        res = unittest.TextTestRunner(verbosity=0).run(suite)
        assert res.wasSuccessful(), "compiled reference failed the reference's own NMS tests"
    finally:
        np.testing.assert_array_equal = orig_assert
        del sys.modules["maskrcnn_benchmark"], sys.modules["maskrcnn_benchmark.layers"]
    assert len(calls) == len(expected) == 6
    out = {"num_cases": np.int64(len(calls))}
    for i, ((b, s, t), e) in enumerate(zip(calls, expected)):
        out[f"boxes_{i}"] = b
        out[f"scores_{i}"] = s
        out[f"thresh_{i}"] = np.float32(t)
        out[f"expected_{i}"] = e
    np.savez(os.path.join(HERE, "nms_reference_tests.npz"), **out)
    print("nms_reference_tests.npz:", [len(e) for e in expected])


def capture_reference_box_coder_test():
    """Run the reference's BoxCoder (pure torch, modeling/box_coder.py) inside its own test."""
    src = open(os.path.join(REF, "maskrcnn_benchmark", "modeling", "box_coder.py")).read()
    bc = types.ModuleType("maskrcnn_benchmark.modeling.box_coder")
    exec(compile(src, "box_coder.py", "exec"), bc.__dict__)
    rec = {}
    orig_decode = bc.BoxCoder.decode

    def rec_decode(self, rel_codes, boxes):
        out = orig_decode(self, rel_codes, boxes)
        rec["weights"] = np.asarray(self.weights, np.float32)
        rec["rel_codes"] = rel_codes.numpy().copy()
        rec["boxes"] = boxes.numpy().copy()
        rec["decoded"] = out.numpy().copy()
        return out

    bc.BoxCoder.decode = rec_decode
    pkg = types.ModuleType("maskrcnn_benchmark")
    modeling = types.ModuleType("maskrcnn_benchmark.modeling")
    sys.modules.update({"maskrcnn_benchmark": pkg, "maskrcnn_benchmark.modeling": modeling,
                        "maskrcnn_benchmark.modeling.box_coder": bc})
    orig_close = np.testing.assert_allclose

    def rec_close(actual, desired, *a, **k):
        rec["expected"] = np.asarray(desired).copy()
        rec["atol"] = np.float64(k.get("atol", 0))
        return orig_close(actual, desired, *a, **k)

    np.testing.assert_allclose = rec_close
    try:
        src = open(os.path.join(REF, "tests", "test_box_coder.py")).read()
        mod = types.ModuleType("ref_test_box_coder")
        exec(compile(src, "ref_test_box_coder.py", "exec"), mod.__dict__)
        suite = unittest.defaultTestLoader.loadTestsFromTestCase(mod.TestBoxCoder)
        res = unittest.TextTestRunner(verbosity=0).run(suite)
        assert res.wasSuccessful()
    finally:
        np.testing.assert_allclose = orig_close
        for k in ("maskrcnn_benchmark", "maskrcnn_benchmark.modeling",
                  "maskrcnn_benchmark.modeling.box_coder"):
            del sys.modules[k]
    # also capture encode on a seeded pair for round-trip coverage
    torch.manual_seed(0)
    prop = torch.rand(64, 4) * 200
    prop[:, 2:] += prop[:, :2] + 1
    gt = prop + torch.randn(64, 4) * 5
    gt[:, 2:] = torch.max(gt[:, 2:], gt[:, :2] + 1)
    coder = bc.BoxCoder(weights=(10.0, 10.0, 5.0, 5.0))
    rec["enc_weights"] = np.asarray(coder.weights, np.float32)
    rec["enc_proposals"] = prop.numpy()
    rec["enc_reference_boxes"] = gt.numpy()
    rec["enc_codes"] = bc.BoxCoder.encode(coder, gt, prop).numpy()
    np.savez(os.path.join(HERE, "box_coder_reference_tests.npz"), **rec)
    print("box_coder_reference_tests.npz:", rec["decoded"].shape)


def capture_ref_cpu_vectors(ref_mod):
    out = {}
    # ROIAlign forward: cfg-1 geometry (14x14 map, 512 ROIs spilling over the border), 8 channels
    inp, rois, scale = synth.cfg1_roi_align(seed=0, K=64, C=8)
    out["ra_input"], out["ra_rois"], out["ra_scale"] = inp, rois, np.float32(scale)
    for i, (ph, pw, sr) in enumerate([(7, 7, 2), (14, 14, 2), (7, 7, 0), (3, 5, 3)]):
        y = ref_mod.roi_align_forward(torch.from_numpy(inp), torch.from_numpy(rois), scale, ph,
                                      pw, sr).numpy()
        out[f"ra_cfg_{i}"] = np.asarray([ph, pw, sr], np.int64)
        out[f"ra_out_{i}"] = y
    # two-image FPN-like map with batch index 1 and big ROIs (adaptive grid > 2)
    rng = np.random.RandomState(11)
    inp2 = rng.randn(2, 4, 25, 42).astype(np.float32)
    rois2 = synth.fpn_rois(seed=12, per_image=16, n_images=2)
    out["ra2_input"], out["ra2_rois"] = inp2, rois2
    for i, (ph, pw, sr) in enumerate([(7, 7, 2), (7, 7, 0)]):
        out[f"ra2_cfg_{i}"] = np.asarray([ph, pw, sr], np.int64)
        out[f"ra2_out_{i}"] = ref_mod.roi_align_forward(
            torch.from_numpy(inp2), torch.from_numpy(rois2), 1.0 / 32, ph, pw, sr).numpy()
    # NMS: clustered + uniform, several thresholds (inputs are regenerated from tools/synth.py in
    # the tests; only the kept indices are stored)
    cases = []
    for n, uniform, seed in [(300, False, 2), (819, False, 3), (2000, False, 4), (2000, True, 5),
                             (4500, False, 6), (65, False, 7), (64, True, 8)]:
        b, s = synth.nms_boxes(n, seed=seed, uniform=uniform)
        for thr in (0.3, 0.5, 0.7):
            k = ref_mod.nms(torch.from_numpy(b), torch.from_numpy(s), thr).numpy()
            key = f"nms_{n}_{int(uniform)}_{seed}_{int(thr * 100)}"
            out[key] = k.astype(np.int64)
            cases.append(key)
    out["nms_cases"] = np.asarray(cases)
    np.savez_compressed(os.path.join(HERE, "ref_cpu_vectors.npz"), **out)
    print("ref_cpu_vectors.npz:", len(out), "arrays")


def capture_focal_python_composite():
    path = os.path.join(REF, "maskrcnn_benchmark", "layers", "sigmoid_focal_loss.py")
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "sigmoid_focal_loss_cpu"]
    assert len(fn) == 1
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    logits, targets = synth.focal_inputs(512, 80, seed=5)
    targets[:8] = np.arange(1, 9)  # make sure positives exist
    lt = torch.from_numpy(logits).requires_grad_(True)
    loss = ns["sigmoid_focal_loss_cpu"](lt, torch.from_numpy(targets), 2.0, 0.25)
    d_loss = torch.from_numpy(np.random.RandomState(9).rand(512, 80).astype(np.float32))
    (loss * d_loss).sum().backward()
    np.savez_compressed(os.path.join(HERE, "focal_python_composite.npz"), logits=logits,
                        targets=targets, gamma=np.float32(2.0), alpha=np.float32(0.25),
                        losses=loss.detach().numpy(), d_losses=d_loss.numpy(),
                        d_logits=lt.grad.numpy())
    print("focal_python_composite.npz:", loss.shape)


if __name__ == "__main__":
    assert os.path.isdir(REF), "run in the build container (needs /root/reference)"
    build_ref.build()
    ref_mod = oracle.ref()
    capture_reference_nms_tests(ref_mod)
    capture_reference_box_coder_test()
    capture_ref_cpu_vectors(ref_mod)
    capture_focal_python_composite()
