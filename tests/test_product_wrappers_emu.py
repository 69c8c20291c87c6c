"""The product's own Python side of the operators — `maskrcnn_benchmark._C` wrappers (argument checks, workspaces, ctypes
marshalling, layout pipelines) and the autograd Functions / nn.Modules of `maskrcnn_benchmark.layers` — run WITHOUT a GPU over
the host-emulation build of the HIP sources (tests/cpu_shim.py backend "emu-lib": only the library handle under `_C` is
swapped) and checked against the oracle.  The `-m gpu` suite checks the same surface on the device; this file is what
catches a wrapper bug before GPU minutes are spent."""
import numpy as np
import pytest
import torch

import cpu_shim
import oracle
import synth


@pytest.fixture(autouse=True)
def _product_wrappers():
    with cpu_shim.install("emu-lib"):
        yield


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("modulated", [False, True])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float16, 3e-2)])
@pytest.mark.parametrize("geo", [dict(C=16, Cout=24, H=13, W=17, k=3, pad=1, stride=1, dil=1, g=1, dg=1),
                                 dict(C=32, Cout=32, H=12, W=10, k=3, pad=2, stride=2, dil=2, g=2, dg=2)])
def test_deform_conv_modules_forward_and_backward_equal_the_oracle(modulated, dtype, tol, geo):
    """DeformConv / ModulatedDeformConv (layers/dcn; reference layers/dcn/deform_conv_func.py:14-262): forward, and the
    gradients w.r.t. input, offset, mask, weight and bias — through `_C.deform_conv_forward` / `deform_conv_backward_all`
    (channels-last pipeline and the kept forward copies where the plan allows them, the im2col path elsewhere)"""
    from maskrcnn_benchmark.layers import DeformConv, ModulatedDeformConv
    rng = np.random.RandomState(7)
    C, Cout, H, W, k = geo["C"], geo["Cout"], geo["H"], geo["W"], geo["k"]
    pad, stride, dil, g, dg = geo["pad"], geo["stride"], geo["dil"], geo["g"], geo["dg"]
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    B = 2
    x = rng.randn(B, C, H, W).astype(np.float32)
    off = (rng.randn(B, 2 * dg * k * k, Ho, Wo) * 1.5).astype(np.float32)
    mask = rng.rand(B, dg * k * k, Ho, Wo).astype(np.float32) if modulated else None
    gout = rng.randn(B, Cout, Ho, Wo).astype(np.float32)
    torch.manual_seed(3)
    if modulated:
        m = ModulatedDeformConv(C, Cout, k, stride=stride, padding=pad, dilation=dil, groups=g, deformable_groups=dg, bias=True)
        m.bias.data.normal_()
    else:
        m = DeformConv(C, Cout, k, stride=stride, padding=pad, dilation=dil, groups=g, deformable_groups=dg)
    m = m.to(dtype)
    # the oracle sees the values the layer sees (rounded to the layer's dtype)
    rd = lambda a: None if a is None else _t(a).to(dtype).float().numpy()   # noqa: E731
    x, off, mask, gout = rd(x), rd(off), rd(mask), rd(gout)
    w = m.weight.detach().float().numpy()
    b = m.bias.detach().float().numpy() if modulated else None
    tx, toff = _t(x).to(dtype).requires_grad_(), _t(off).to(dtype).requires_grad_()
    tmask = _t(mask).to(dtype).requires_grad_() if modulated else None
    out = m(tx, toff, tmask) if modulated else m(tx, toff)
    assert out.dtype == dtype and tuple(out.shape) == (B, Cout, Ho, Wo)
    ref = oracle.deform_conv_forward(x, off, mask, w, b, (pad, pad), (stride, stride), (dil, dil), g, dg)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(out.detach().float().numpy() - ref).max() <= tol * scale
    out.backward(_t(gout).to(dtype))
    gin, goff, gmask, gw, gb = oracle.deform_conv_backward(x, off, mask, w, gout, modulated, (pad, pad), (stride, stride),
                                                           (dil, dil), g, dg)
    for name, got, want in (("input", tx.grad, gin), ("offset", toff.grad, goff), ("mask", None if tmask is None else tmask.grad, gmask),
                            ("weight", m.weight.grad, gw), ("bias", m.bias.grad if modulated else None, gb)):
        if want is None:
            continue
        assert got is not None and got.dtype == dtype, name
        s = max(1.0, float(np.abs(want).max()))
        assert np.abs(got.float().numpy() - want).max() <= tol * s, (name, np.abs(got.float().numpy() - want).max(), s)


def test_roi_align_modules_over_the_pyramid_equal_the_oracle():
    """Pooler (modeling/poolers.py; reference :45-121) through `_C.roi_align_fpn_forward` / `_backward`: the in-kernel level
    assignment, the workspace / host-array plumbing of the multi-level launch, and the autograd glue"""
    from maskrcnn_benchmark.modeling.poolers import Pooler
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    rng = np.random.RandomState(11)
    N, C = 2, 8
    scales = (0.25, 0.125, 0.0625, 0.03125)
    feats = [rng.randn(N, C, int(160 * s), int(224 * s)).astype(np.float32) for s in scales]
    boxes = []
    for n in range(N):
        wh = np.exp(rng.uniform(np.log(6), np.log(900), (40, 2)))   # every pyramid level gets ROIs
        xy = rng.uniform(0, [224, 160], (40, 2)) - wh / 4
        boxes.append(np.concatenate([xy, xy + wh], 1).astype(np.float32))
    pooler = Pooler((7, 7), scales, 2)
    tf = [_t(f).requires_grad_() for f in feats]
    out = pooler(tf, [BoxList(_t(b), (224, 160), mode="xyxy") for b in boxes])
    rois = np.concatenate([np.concatenate([np.full((len(b), 1), i, np.float32), b], 1) for i, b in enumerate(boxes)])
    lv = oracle.fpn_level(rois, 2, 5)
    ref = np.zeros((len(rois), C, 7, 7), np.float32)
    for l, (f, s) in enumerate(zip(feats, scales)):
        sel = np.nonzero(lv == l)[0]
        assert sel.size > 0
        ref[sel] = oracle.roi_align_forward(f, rois[sel], s, 7, 7, 2)
    assert np.abs(out.detach().numpy() - ref).max() <= 1e-5
    g = rng.randn(*ref.shape).astype(np.float32)
    out.backward(_t(g))
    for l, (f, s) in enumerate(zip(feats, scales)):
        sel = np.nonzero(lv == l)[0]
        want = oracle.roi_align_backward(g[sel], rois[sel], s, 7, 7, *f.shape, 2, acc64=True)
        assert np.abs(tf[l].grad.numpy() - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("modulated", [False, True])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 2e-2)])
def test_deform_conv_modules_on_channels_last_tensors_equal_the_nchw_run(modulated, dtype, tol):
    """DeformConv / ModulatedDeformConv on a channels-last input (and channels-last parameters, as `model.to(memory_format=
    torch.channels_last)` leaves them): the channels-last pipeline reads the input in place, writes a channels-last output
    with one GEMM and returns a channels-last input gradient — same values as the NCHW call (the GEMMs' summation order
    differs: 1e-5 fp32, 2e-2 half)."""
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.layers import DeformConv, ModulatedDeformConv
    torch.manual_seed(3)
    B, C, H, W, Cout, k = 2, 32, 11, 13, 32, 3
    x = torch.randn(B, C, H, W).to(dtype)
    off = (torch.randn(B, 2 * k * k, H, W) * 1.5).to(dtype)
    msk = torch.rand(B, k * k, H, W).to(dtype)
    layer = (ModulatedDeformConv(C, Cout, k, 1, 1, bias=True) if modulated else DeformConv(C, Cout, k, 1, 1)).to(dtype)
    g = torch.randn(B, Cout, H, W).to(dtype)
    cl = torch.channels_last

    def run(fmt):
        layer.to(memory_format=fmt)
        for p in layer.parameters():
            p.grad = None
        xi = x.clone(memory_format=fmt).requires_grad_()
        oi = off.clone(memory_format=fmt).requires_grad_()
        args = (xi, oi) + ((msk.clone(memory_format=fmt).requires_grad_(),) if modulated else ())
        y = layer(*args)
        y.backward(g.clone(memory_format=fmt))
        return y, xi.grad, oi.grad, [p.grad.clone() for p in layer.parameters()]

    y0, gx0, go0, gp0 = run(torch.contiguous_format)
    y1, gx1, go1, gp1 = run(cl)
    assert _C.is_channels_last(y1) and _C.is_channels_last(gx1)
    scale = float(y0.float().abs().max())
    assert torch.allclose(y1.float(), y0.float(), rtol=tol, atol=tol * scale)
    assert torch.allclose(gx1.float(), gx0.float(), rtol=tol, atol=tol * float(gx0.float().abs().max()))
    assert torch.allclose(go1.float(), go0.float(), rtol=tol, atol=tol * float(go0.float().abs().max()))
    for a, b in zip(gp1, gp0):
        assert torch.allclose(a.float(), b.float(), rtol=tol, atol=tol * float(b.float().abs().max()) + 1e-6)


def test_nms_wrappers_equal_the_oracle_bit_exactly():
    from maskrcnn_benchmark import _C
    b, s = synth.nms_boxes(700, seed=5)
    keep = _C.nms(_t(b), _t(s), 0.6).numpy()
    assert np.array_equal(keep, oracle.nms(b, s, 0.6))
    # segmented launch with a dense mask (what the RPN proposal selector calls)
    segs = np.asarray([0, 250, 250, 700], np.int32)
    order = np.concatenate([np.argsort(-s[lo:hi], kind="stable") + lo for lo, hi in zip(segs[:-1], segs[1:])])
    bb, ss = b[order], s[order]
    mask, num = _C.nms_batched_mask(_t(bb), _t(ss), _t(segs), 450, 0.6)
    for i, (lo, hi) in enumerate(zip(segs[:-1], segs[1:])):
        want = np.zeros(hi - lo, bool)
        if hi > lo:
            want[oracle.nms(bb[lo:hi], ss[lo:hi], 0.6)] = True
        assert np.array_equal(mask.numpy()[lo:hi], want) and int(num[i]) == int(want.sum())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.float16, 2e-3), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("shape", [((2, 6, 10, 16), (5, 8)), ((1, 3, 7, 9), (4, 5)), ((2, 4, 13, 21), (7, 11)), ((1, 2, 6, 6), (6, 6))])
def test_fpn_topdown_equals_interpolate_plus_add(dtype, tol, shape):
    """_C.fpn_topdown (csrc/fpn_topdown.hip) = lateral + F.interpolate(top, size, mode="nearest") (reference
    modeling/backbone/fpn.py:59-64), forward and both gradients: exact 2x, odd sizes (25 -> 13-like ratios), equal sizes"""
    from maskrcnn_benchmark import _C
    (N, C, H, W), (h, w) = shape
    g = torch.Generator().manual_seed(H * W + h)
    lat = torch.randn(N, C, H, W, generator=g).to(dtype).requires_grad_()
    top = torch.randn(N, C, h, w, generator=g).to(dtype).requires_grad_()
    up = torch.randn(N, C, H, W, generator=g).to(dtype)
    out = _C.fpn_topdown(lat, top)
    out.backward(up)
    lat2, top2 = lat.detach().float().requires_grad_(), top.detach().float().requires_grad_()
    ref = lat2 + torch.nn.functional.interpolate(top2, size=(H, W), mode="nearest")
    ref.backward(up.float())
    assert torch.allclose(out.float(), ref, rtol=tol, atol=tol)
    assert torch.equal(lat.grad, up)
    assert torch.allclose(top.grad.float(), top2.grad, rtol=tol, atol=tol * 4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [((2, 8, 10, 16), (5, 8)), ((1, 12, 7, 9), (4, 5)), ((2, 4, 13, 21), (7, 11)), ((1, 6, 6, 6), (6, 6))])
def test_fpn_topdown_channels_last_is_bit_equal_to_the_nchw_kernels(dtype, shape):
    """a channels-last lateral takes the NHWC kernels (detops_fpn_topdown_*_nhwc) and returns channels-last tensors; values and
    both gradients are bit-equal to the NCHW kernels' (same arithmetic, same summation order)"""
    from maskrcnn_benchmark import _C
    (N, C, H, W), (h, w) = shape
    g = torch.Generator().manual_seed(H * W + h + C)
    lat = torch.randn(N, C, H, W, generator=g).to(dtype)
    top = torch.randn(N, C, h, w, generator=g).to(dtype)
    up = torch.randn(N, C, H, W, generator=g).to(dtype)
    a, b = lat.clone().requires_grad_(), top.clone().requires_grad_()
    ref = _C.fpn_topdown(a, b)
    ref.backward(up)
    cl = torch.channels_last
    a2, b2 = lat.contiguous(memory_format=cl).requires_grad_(), top.contiguous(memory_format=cl).requires_grad_()
    out = _C.fpn_topdown(a2, b2)
    assert _C.is_channels_last(out)
    out.backward(up.contiguous(memory_format=cl))
    assert torch.equal(out, ref)
    assert torch.equal(a2.grad, a.grad) and torch.equal(b2.grad, b.grad)
    assert _C.is_channels_last(b2.grad) or b2.grad.is_contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C,H,W", [(8, 9, 11), (12, 5, 7), (64, 6, 10), (3, 4, 5), (2048, 2, 3)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True), (False, True)])
def test_frozen_bn_channels_last_is_bit_equal_to_the_nchw_kernels(dtype, C, H, W, relu, res):
    """FrozenBatchNorm2d.fused on a channels-last activation: NHWC kernels, channels-last output and gradients, bit-equal to the
    NCHW kernels (vector widths 1-8, channel counts that are / are not powers of two, C larger than a block's span)"""
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.layers import FrozenBatchNorm2d
    rng = np.random.RandomState(C + H)
    bn = FrozenBatchNorm2d(C)
    bn.weight.copy_(_t(rng.rand(C).astype(np.float32) + 0.5))
    bn.bias.copy_(_t(rng.randn(C).astype(np.float32)))
    bn.running_mean.copy_(_t(rng.randn(C).astype(np.float32)))
    bn.running_var.copy_(_t(rng.rand(C).astype(np.float32) + 0.2))
    x = _t(rng.randn(2, C, H, W).astype(np.float32)).to(dtype)
    r = _t(rng.randn(2, C, H, W).astype(np.float32)).to(dtype) if res else None
    gy = _t(rng.randn(2, C, H, W).astype(np.float32)).to(dtype)
    cl = torch.channels_last

    def run(fmt):
        xi = x.clone(memory_format=fmt).requires_grad_()
        ri = r.clone(memory_format=fmt).requires_grad_() if res else None
        y = bn.fused(xi, relu=relu, residual=ri)
        y.backward(gy.contiguous(memory_format=fmt))
        return y, xi.grad, (ri.grad if res else None)

    y0, gx0, gr0 = run(torch.contiguous_format)
    y1, gx1, gr1 = run(cl)
    assert _C.is_channels_last(y1) and _C.is_channels_last(gx1)
    assert torch.equal(y1, y0) and torch.equal(gx1, gx0)
    if res:
        assert torch.equal(gr1, gr0)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 2e-3), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C,H,W", [(256, 5, 7), (64, 9, 3), (8, 6, 6), (12, 5, 7), (81, 4, 4), (3, 7, 5)])
@pytest.mark.parametrize("relu", [False, True])
def test_conv_bias_act_on_channels_last_equals_conv_with_bias(dtype, tol, C, H, W, relu):
    """layers.misc.conv_bias_act (csrc/bias_act.hip): the convolution without its bias + the fused bias(+ReLU) pass whose
    backward also produces the bias gradient (fp32, deterministic column sums); widths that take the fused kernel (C | 1024,
    C % 4 == 0) and those that take column_sum (12, 81, 3); fp32 and the two autocast storage types"""
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.layers.misc import conv_bias_act
    rng = np.random.RandomState(C * 3 + H)
    conv = torch.nn.Conv2d(4, C, 1)
    with torch.no_grad():
        conv.bias.copy_(_t(rng.randn(C).astype(np.float32)))
    x = _t(rng.randn(2, 4, H, W).astype(np.float32))
    gy = _t(rng.randn(2, C, H, W).astype(np.float32)).to(dtype)
    # the convolution itself runs in fp32 on the host and is cast: only the bias pass is under test
    pre = torch.nn.functional.conv2d(x, conv.weight).detach().to(dtype).contiguous(memory_format=torch.channels_last)
    xi = pre.clone(memory_format=torch.channels_last).requires_grad_()
    assert _C.bias_act_supported(xi, conv.bias)
    y = _C.bias_act(xi, conv.bias, relu)
    assert _C.is_channels_last(y) and y.dtype == dtype
    y.backward(gy.contiguous(memory_format=torch.channels_last))
    ref = pre.float() + conv.bias.detach().view(1, -1, 1, 1)
    ref = ref.relu() if relu else ref
    assert torch.allclose(y.float(), ref, rtol=tol, atol=tol)
    mask = (y.float() > 0).float() if relu else torch.ones_like(ref)
    g = gy.float() * mask
    assert torch.equal(xi.grad.float(), g)                                   # a select: exact in every storage type
    assert conv.bias.grad.dtype == torch.float32
    want = g.double().sum((0, 2, 3))
    assert torch.allclose(conv.bias.grad.double(), want, rtol=1e-5, atol=1e-4)   # fp32 sums of the stored values
    # deterministic: the same call again gives the same bits
    first = conv.bias.grad.clone()
    conv.bias.grad = None
    xi2 = pre.clone(memory_format=torch.channels_last).requires_grad_()
    _C.bias_act(xi2, conv.bias, relu).backward(gy.contiguous(memory_format=torch.channels_last))
    assert torch.equal(conv.bias.grad, first)
    # the module-level entry takes the same route (fp32 only on the host: F.conv2d has no half kernels here)
    if dtype == torch.float32:
        conv.bias.grad = None
        xc = x.clone(memory_format=torch.channels_last)
        out = conv_bias_act(conv, xc, relu=relu)
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)


def test_fpn_module_uses_the_fused_topdown_step_and_equals_the_composition(monkeypatch):
    from maskrcnn_benchmark import _C
    from maskrcnn_benchmark.modeling.backbone import fpn as fpn_mod
    from maskrcnn_benchmark.modeling.make_layers import conv_with_kaiming_uniform
    torch.manual_seed(0)
    net = fpn_mod.FPN([4, 8, 16, 32], 8, conv_with_kaiming_uniform(), fpn_mod.LastLevelMaxPool())
    feats = [torch.randn(2, c, s, s + 2, requires_grad=True) for c, s in ((4, 32), (8, 16), (16, 8), (32, 4))]
    calls = []
    monkeypatch.setattr(_C, "fpn_topdown", (lambda a, b, _f=_C.fpn_topdown: (calls.append(1), _f(a, b))[1]))
    outs = net(feats)
    assert len(calls) == 3 and len(outs) == 5
    sum(o.sum() for o in outs).backward()
    grads = [f.grad.clone() for f in feats]
    for f in feats:
        f.grad = None
    monkeypatch.setattr(_C, "on_device", lambda t: False)      # the ATen composition
    ref = net(feats)
    sum(o.sum() for o in ref).backward()
    for a, b in zip(outs, ref):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)
    for a, f in zip(grads, feats):
        assert torch.allclose(a, f.grad, rtol=1e-5, atol=1e-5)


def test_frozen_batch_norm_module_and_focal_loss_equal_the_torch_formulas():
    from maskrcnn_benchmark.layers import FrozenBatchNorm2d, SigmoidFocalLoss
    rng = np.random.RandomState(2)
    bn = FrozenBatchNorm2d(12)
    bn.weight.copy_(_t(rng.rand(12).astype(np.float32) + 0.5))
    bn.bias.copy_(_t(rng.randn(12).astype(np.float32)))
    bn.running_mean.copy_(_t(rng.randn(12).astype(np.float32)))
    bn.running_var.copy_(_t(rng.rand(12).astype(np.float32) + 0.2))
    x = _t(rng.randn(2, 12, 9, 11).astype(np.float32)).requires_grad_()
    y = bn(x)
    scale = bn.weight * bn.running_var.rsqrt()
    want = x.detach() * scale.reshape(1, -1, 1, 1) + (bn.bias - bn.running_mean * scale).reshape(1, -1, 1, 1)
    assert torch.allclose(y, want, rtol=1e-5, atol=1e-5)
    y.backward(torch.ones_like(y))
    assert torch.allclose(x.grad, scale.reshape(1, -1, 1, 1).expand_as(x), rtol=1e-6, atol=0)
    logits, targets = synth.focal_inputs(500, 20)
    tl = _t(logits).requires_grad_()
    loss = SigmoidFocalLoss(2.0, 0.25)(tl, _t(targets))
    ref = oracle.sigmoid_focal_loss_forward(logits, targets, 2.0, 0.25)
    assert abs(loss.item() - float(ref.sum())) <= 1e-4 * float(ref.sum())
    loss.backward()
    want = oracle.sigmoid_focal_loss_backward(logits, targets, np.ones_like(logits), 2.0, 0.25)
    assert np.allclose(tl.grad.numpy(), want, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("config", ["e2e_mask_rcnn_R_50_FPN_1x.yaml", "retinanet/retinanet_R-50-FPN_1x.yaml"])
def test_detector_trains_and_detects_through_the_product_wrappers(config):
    """TrainStep (forward, backward, fused-SGD update) for a few iterations and an eval-mode forward of the tiny detector
    with every operator served by `_C` as shipped: falling loss, finite gradients, and the SAME detections as with the
    oracle stand-ins (tests/cpu_shim.py default backend) from the same weights"""
    from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    from maskrcnn_benchmark.engine.ddp_step import TrainStep, make_overlapped_sgd
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = load_cfg(config, ["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150,
                            "MODEL.RPN.PRE_NMS_TOP_N_TEST", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TEST", 60,
                            "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32, "MODEL.RESNETS.RES2_OUT_CHANNELS", 16,
                            "MODEL.RESNETS.WIDTH_PER_GROUP", 4, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16,
                            "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32, "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16),
                            "MODEL.ROI_HEADS.SCORE_THRESH", 0.0, "MODEL.RETINANET.INFERENCE_TH", 0.0, "SOLVER.BASE_LR", 0.002])
    torch.manual_seed(0)
    model = build_detection_model(cfg).train()
    ds = SyntheticCOCODataset(length=2, height=96, width=128, with_masks=cfg.MODEL.MASK_ON, min_objects=2, max_objects=4)
    images, targets, _ = BatchCollator(32)([ds[0], ds[1]])
    step = TrainStep(model, make_overlapped_sgd(cfg, model), None, "float32", "cpu")
    first = {k: v.item() for k, v in step(images, list(targets)).items()}
    for _ in range(3):
        last = {k: v.item() for k, v in step(images, list(targets)).items()}
    assert all(np.isfinite(v) for v in last.values()) and sum(last.values()) < sum(first.values())
    model.eval()
    with torch.no_grad():
        det = model(images)
    # the same weights with the oracle stand-ins and the CPU compositions: inside this file's emu-lib install, hand `_C` its
    # own library handle and device test back for the duration of the stand-in run
    from maskrcnn_benchmark import _C, _lib
    on, lib = _C.on_device, _C.lib
    _C.on_device, _C.lib = (lambda t: t.is_cuda), _lib.lib
    try:
        with cpu_shim.install("oracle"), torch.no_grad():
            ref = model(images)
    finally:
        _C.on_device, _C.lib = on, lib
    assert len(det) == len(ref) == 2
    for d, r in zip(det, ref):
        assert d.has_field("scores") and d.has_field("labels") and len(d) > 0
        assert abs(len(d) - len(r)) <= max(2, len(r) // 10)
        k = min(len(d), len(r), 5)
        ds_, rs_ = d.get_field("scores").sort(descending=True), r.get_field("scores").sort(descending=True)
        assert torch.allclose(ds_[0][:k], rs_[0][:k], rtol=1e-3, atol=1e-4)
        assert torch.allclose(d.bbox[ds_[1][:k]], r.bbox[rs_[1][:k]], rtol=1e-3, atol=5e-2)
        if cfg.MODEL.MASK_ON:
            assert d.has_field("mask") and d.get_field("mask").shape[0] == len(d)


@pytest.mark.parametrize("config", ["e2e_mask_rcnn_R_50_FPN_1x.yaml", "retinanet/retinanet_R-50-FPN_1x.yaml"])
@pytest.mark.parametrize("heads", [False, True])
def test_detector_on_a_channels_last_pyramid_equals_the_nchw_run(heads, config):
    """GeneralizedRCNN.set_channels_last: backbone + FPN on NHWC activations (the fused FrozenBN / top-down kernels follow
    the layout), the pyramid handed over as NCHW (heads=False) or as it is (heads=True).  A memory format changes no value:
    losses and parameter gradients of one training forward / backward equal the NCHW run's (CPU convolutions may pick another
    summation order per layout: 1e-4)."""
    from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = load_cfg(config,
                   ["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150,
                    "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 512, "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 100000,
                    "MODEL.RESNETS.RES2_OUT_CHANNELS", 16,
                    "MODEL.RESNETS.WIDTH_PER_GROUP", 4, "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16,
                    "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32, "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16)])
    torch.manual_seed(0)
    model = build_detection_model(cfg).train()
    ds = SyntheticCOCODataset(length=2, height=96, width=128, with_masks=cfg.MODEL.MASK_ON, min_objects=2, max_objects=4)
    images, targets, _ = BatchCollator(32)([ds[0], ds[1]])

    def run():
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(5)       # the samplers' keys (quotas >= candidates: take-all, but keep the streams equal anyway)
        losses = model(images, list(targets))
        sum(losses.values()).backward()
        return ({k: float(v.detach()) for k, v in losses.items()},
                {n: p.grad.detach().clone().contiguous() for n, p in model.named_parameters() if p.grad is not None})

    l0, g0 = run()
    keys = list(model.state_dict().keys())
    model.set_channels_last(True, heads=heads)
    assert list(model.state_dict().keys()) == keys
    l1, g1 = run()
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-4 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    assert g0.keys() == g1.keys()
    for n in g0:
        assert torch.allclose(g0[n], g1[n], rtol=1e-3, atol=1e-5), n


def test_roi_pool_and_deformable_psroi_pooling_layers_equal_the_oracle():
    """ROIPool (layers/roi_pool.py; reference :11-63) and DeformRoIPooling (layers/dcn/deform_pool_module.py; reference
    layers/dcn/deform_pool_func.py:8-95) with their autograd functions, through `_C` as shipped"""
    from maskrcnn_benchmark.layers import ROIPool
    from maskrcnn_benchmark.layers.dcn.deform_pool_module import DeformRoIPooling
    rng = np.random.RandomState(13)
    N, C, H, W, K = 2, 6, 20, 28, 30
    x = rng.randn(N, C, H, W).astype(np.float32)
    wh = rng.uniform(8, 200, (K, 2))
    xy = rng.uniform(0, [W * 16 - 8, H * 16 - 8], (K, 2))
    rois = np.concatenate([rng.randint(0, N, (K, 1)), xy, xy + wh], 1).astype(np.float32)
    tx = _t(x).requires_grad_()
    out = ROIPool((5, 4), 1.0 / 16)(tx, _t(rois))
    ref, argmax = oracle.roi_pool_forward(x, rois, 1.0 / 16, 5, 4)
    assert np.array_equal(out.detach().numpy(), ref)
    g = rng.randn(*ref.shape).astype(np.float32)
    out.backward(_t(g))
    want = oracle.roi_pool_backward(g, rois, argmax, N, C, H, W)
    assert np.abs(tx.grad.numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    # deformable position-sensitive pooling: output_dim 3, group 2, pooled 4, part 4, 2 samples per part, learned shifts
    D, G, P, S, std = 3, 2, 4, 2, 0.1
    data = rng.randn(N, D * G * G, H, W).astype(np.float32)
    trans = (rng.randn(K, 2, P, P) * 0.5).astype(np.float32)
    for no_trans in (True, False):
        td, tt = _t(data).requires_grad_(), _t(trans).requires_grad_()
        layer = DeformRoIPooling(1.0 / 16, P, D, no_trans, group_size=G, part_size=P, sample_per_part=S, trans_std=std)
        out = layer(td, _t(rois), tt)
        ro, rc = oracle.deform_psroi_pool_forward(data, rois, trans, no_trans, 1.0 / 16, D, G, P, P, S, std)
        assert np.abs(out.detach().numpy() - ro).max() <= 1e-5
        g = rng.randn(*ro.shape).astype(np.float32)
        out.backward(_t(g))
        dg, tg = oracle.deform_psroi_pool_backward(g, data, rois, trans, rc, no_trans, 1.0 / 16, D, G, P, P, S, std, acc64=True)
        assert np.abs(td.grad.numpy() - dg).max() <= 1e-4 * max(1.0, np.abs(dg).max())
        if not no_trans:
            assert np.abs(tt.grad.numpy() - tg).max() <= 1e-4 * max(1.0, np.abs(tg).max())


def test_deformable_detector_trains_through_the_product_wrappers():
    """R-50-FPN with deformable convolutions in C3-C5 (the cfg-5 family, narrow): forward + backward of the detector through
    `ModulatedDeformConvPack` / `_C.deform_conv_*` as shipped: finite losses, a finite gradient on every trainable parameter"""
    from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
    from maskrcnn_benchmark.engine.bench_step import load_cfg
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    cfg = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml", ["MODEL.DEVICE", "cpu", "MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 100,
                                                     "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 150, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 32,
                                                     "MODEL.RESNETS.RES2_OUT_CHANNELS", 16, "MODEL.RESNETS.WIDTH_PER_GROUP", 4,
                                                     "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 16, "MODEL.ROI_BOX_HEAD.MLP_HEAD_DIM", 32,
                                                     "MODEL.ROI_MASK_HEAD.CONV_LAYERS", (16, 16),
                                                     "MODEL.RESNETS.STAGE_WITH_DCN", "(False, True, True, True)"])
    torch.manual_seed(0)
    model = build_detection_model(cfg).train()
    assert sum(type(m).__name__ == "ModulatedDeformConvPack" or type(m).__name__ == "DeformConv" or "DCN" in type(m).__name__
               for m in model.modules()) > 0
    ds = SyntheticCOCODataset(length=2, height=96, width=128, with_masks=True, min_objects=2, max_objects=4)
    images, targets, _ = BatchCollator(32)([ds[0], ds[1]])
    losses = model(images, list(targets))
    assert all(torch.isfinite(v) for v in losses.values())
    sum(losses.values()).backward()
    trainable = [p for p in model.parameters() if p.requires_grad]
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in trainable) and len(trainable) > 90


def test_float64_operators_equal_the_float32_oracle_and_the_reference_semantics():
    """The reference dispatches ROIAlign / ROIPool / SigmoidFocalLoss / nms over float AND double (AT_DISPATCH_FLOATING_TYPES:
    ROIAlign_cuda.cu:283,329, ROIPool_cuda.cu:137,185, SigmoidFocalLoss_cuda.cu:129,173, nms_cpu.cpp:71).  float64 tensors take
    csrc/f64_ops.hip: same results as the fp32 operators on the same (fp32-representable) inputs to fp32 rounding, NMS keeps
    exactly the same boxes, and ROIAlign forward / backward agree with a float64 torch formulation to 1e-12."""
    from maskrcnn_benchmark import _C
    from torch_refs import roi_align_torch
    inp, rois, scale = synth.cfg1_roi_align(seed=12, K=40, C=6)
    x64, r64 = _t(inp).double(), _t(rois).double()
    out = _C.roi_align_forward(x64, r64, scale, 7, 7, 2)
    assert out.dtype == torch.float64
    xg = x64.clone().requires_grad_()
    ref = roi_align_torch(xg, r64, scale, 7, 7, 2)
    assert torch.allclose(out, ref.detach(), rtol=1e-12, atol=1e-12)
    assert np.abs(out.numpy() - oracle.roi_align_forward(inp, rois, scale, 7, 7, 2)).max() <= 1e-5
    g = torch.randn(out.shape, dtype=torch.float64)
    ref.backward(g)
    gin = _C.roi_align_backward(g, r64, scale, 7, 7, *inp.shape, 2)
    assert gin.dtype == torch.float64 and torch.allclose(gin, xg.grad, rtol=1e-10, atol=1e-12)
    # ROIPool
    o32, a32 = oracle.roi_pool_forward(inp, rois, scale, 5, 4)
    o64, a64 = _C.roi_pool_forward(x64, r64, scale, 5, 4)
    assert np.array_equal(o64.numpy().astype(np.float32), o32) and np.array_equal(a64.numpy(), a32)
    gp = torch.randn(o64.shape, dtype=torch.float64)
    gi = _C.roi_pool_backward(gp, x64, r64, a64, scale, 5, 4, *inp.shape)
    want = oracle.roi_pool_backward(gp.numpy().astype(np.float32), rois, a32, *inp.shape)
    assert np.abs(gi.numpy() - want).max() <= 1e-4
    # focal loss
    logits, targets = synth.focal_inputs(300, 20)
    f = _C.sigmoid_focalloss_forward(_t(logits).double(), _t(targets), 20, 2.0, 0.25)
    assert f.dtype == torch.float64
    np.testing.assert_allclose(f.numpy(), oracle.sigmoid_focal_loss_forward(logits, targets, 2.0, 0.25), rtol=1e-4, atol=1e-6)
    d = np.random.RandomState(3).rand(*logits.shape).astype(np.float32)
    b = _C.sigmoid_focalloss_backward(_t(logits).double(), _t(targets), _t(d).double(), 20, 2.0, 0.25)
    np.testing.assert_allclose(b.numpy(), oracle.sigmoid_focal_loss_backward(logits, targets, d, 2.0, 0.25), rtol=1e-4, atol=1e-6)
    # NMS: the same kept set as the fp32 oracle on fp32-representable boxes
    bx, sc = synth.nms_boxes(500, seed=9)
    keep = _C.nms(_t(bx).double(), _t(sc).double(), 0.6)
    assert keep.dtype == torch.int64 and np.array_equal(keep.numpy(), oracle.nms(bx, sc, 0.6))
