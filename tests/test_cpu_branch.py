"""The CPU branch of the drop-in `_C`: `nms` and `roi_align_forward` on CPU tensors, as the reference dispatches
them (csrc/nms.h:19-27, csrc/ROIAlign.h:19-24).  Host code of libdetops_gfx950.so (csrc/cpu_branch.hip) — checked
against the reference's own test vectors, the vectors produced by the reference's compiled CPU kernels, and the
oracle restatement.  Runs without a GPU."""
import os

import numpy as np
import pytest
import torch

import oracle
import synth


@pytest.fixture(scope="module")
def C():
    from maskrcnn_benchmark import _C
    return _C


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _nms(C, b, s, thr):
    return C.nms(torch.from_numpy(np.ascontiguousarray(b, np.float32)), torch.from_numpy(np.ascontiguousarray(s, np.float32)),
                 thr).numpy()


def test_cpu_nms_reference_known_answers(C, golden_dir):
    g = _load(golden_dir, "nms_reference_tests.npz")          # the reference's tests/test_nms.py cases
    for i in range(int(g["num_cases"])):
        np.testing.assert_array_equal(_nms(C, g[f"boxes_{i}"], g[f"scores_{i}"], float(g[f"thresh_{i}"])), g[f"expected_{i}"])


def test_cpu_nms_matches_compiled_reference_vectors(C, golden_dir):
    g = _load(golden_dir, "ref_cpu_vectors.npz")               # outputs of the reference's nms_cpu.cpp, built in place
    for key in g["nms_cases"]:
        _, n, uniform, seed, thr = str(key).split("_")
        b, s = synth.nms_boxes(int(n), seed=int(seed), uniform=bool(int(uniform)))
        np.testing.assert_array_equal(_nms(C, b, s, int(thr) / 100.0), g[str(key)], err_msg=str(key))


@pytest.mark.parametrize("n", [1, 2, 64, 65, 1000])
def test_cpu_nms_vs_oracle_ties_and_edges(C, n):
    b, s = synth.nms_boxes(n, seed=n)
    s[::3] = s[0]                                              # score ties: ascending-index order
    for thr in (0.0, 0.3, 0.7, 1.0):
        np.testing.assert_array_equal(_nms(C, b, s, thr), oracle.nms(b, s, thr))
    out = C.nms(torch.zeros(0, 4), torch.zeros(0), 0.5)
    assert out.dtype == torch.long and out.numel() == 0


def test_cpu_roi_align_forward_golden_reference_vectors(C, golden_dir):
    g = _load(golden_dir, "ref_cpu_vectors.npz")               # outputs of the reference's ROIAlign_cpu.cpp
    t = torch.from_numpy
    for i in range(4):
        ph, pw, sr = [int(v) for v in g[f"ra_cfg_{i}"]]
        out = C.roi_align_forward(t(g["ra_input"]), t(g["ra_rois"]), float(g["ra_scale"]), ph, pw, sr)
        assert np.array_equal(out.numpy(), g[f"ra_out_{i}"])
    for i in range(2):
        ph, pw, sr = [int(v) for v in g[f"ra2_cfg_{i}"]]
        out = C.roi_align_forward(t(g["ra2_input"]), t(g["ra2_rois"]), 1.0 / 32, ph, pw, sr)
        assert np.array_equal(out.numpy(), g[f"ra2_out_{i}"])


@pytest.mark.parametrize("ph,pw,sr", [(7, 7, 2), (14, 14, 2), (7, 7, 0), (3, 5, 3), (1, 1, 1)])
def test_cpu_roi_align_forward_vs_oracle(C, ph, pw, sr):
    inp, rois, scale = synth.cfg1_roi_align(K=96, C=24)        # threaded over ROI ranges
    out = C.roi_align_forward(torch.from_numpy(inp), torch.from_numpy(rois), scale, ph, pw, sr).numpy()
    assert np.array_equal(out, oracle.roi_align_forward(inp, rois, scale, ph, pw, sr))


def test_cpu_roi_align_forward_edges_and_layer(C):
    x = torch.randn(2, 3, 10, 12)
    assert C.roi_align_forward(x, torch.zeros(0, 5), 0.5, 7, 7, 2).shape == (0, 3, 7, 7)
    rois = np.array([[0, -500, -500, -400, -400], [1, 0, 0, 5000, 4000], [1, 3, 3, 3, 3], [0, 5.5, 2.25, 20.75, 17.5]], np.float32)
    for sr in (0, 2):
        out = C.roi_align_forward(x, torch.from_numpy(rois), 1.0, 4, 6, sr).numpy()
        assert np.array_equal(out, oracle.roi_align_forward(x.numpy(), rois, 1.0, 4, 6, sr))
        assert not out[0].any()
    with pytest.raises(RuntimeError):                          # batch index outside the input: an error, not a wild read
        C.roi_align_forward(x, torch.tensor([[2.0, 0, 0, 4, 4]]), 1.0, 2, 2, 2)
    from maskrcnn_benchmark.layers import ROIAlign, nms
    y = ROIAlign((4, 6), 1.0, 2)(x, torch.from_numpy(rois))   # forward-only use of the layer on CPU tensors
    assert np.array_equal(y.numpy(), oracle.roi_align_forward(x.numpy(), rois, 1.0, 4, 6, 2))
    b, s = synth.nms_boxes(50, seed=1)
    np.testing.assert_array_equal(nms(torch.from_numpy(b), torch.from_numpy(s), 0.5).numpy(), oracle.nms(b, s, 0.5))


def test_cpu_branch_float64_is_bit_equal_to_the_compiled_reference(C):
    """double tensors (the reference dispatches float and double: csrc/cpu/ROIAlign_cpu.cpp:242, csrc/cpu/nms_cpu.cpp:71): the
    CPU branch's f64 instantiation against the reference's own kernels compiled in place (oracle/_ref), bit for bit; needs
    /root/reference (build container)."""
    ref = oracle.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    inp, rois, scale = synth.cfg1_roi_align(seed=4, K=60, C=7)
    x, r = torch.from_numpy(inp).double(), torch.from_numpy(rois).double()
    x = x + torch.randn_like(x) * 1e-9          # values that are NOT float32-representable
    for ph, pw, sr in [(7, 7, 2), (3, 5, 0), (14, 14, 2)]:
        a = C.roi_align_forward(x, r, scale, ph, pw, sr)
        b = ref.roi_align_forward(x, r, scale, ph, pw, sr)
        assert a.dtype == torch.float64 and torch.equal(a, b), (ph, pw, sr)
    bx, sc = synth.nms_boxes(800, seed=2)
    tb, ts = torch.from_numpy(bx).double() + 1e-7, torch.from_numpy(sc).double()
    for thr in (0.3, 0.7):
        assert torch.equal(C.nms(tb, ts, thr), ref.nms(tb, ts, thr))
