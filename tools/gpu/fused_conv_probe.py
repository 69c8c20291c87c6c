"""Probe (round 6, late): does MIOpen's fused convolution + bias + ReLU (torch.miopen_convolution_relu / _add_relu) run at the
speed of the plain convolution on the ResNet shapes of the fp32 step?  If it did, FrozenBN (an affine per channel, foldable
into weight and bias) + ReLU would cost no pass of its own in the forward."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "maskrcnn-benchmark_amd"))
import torch
from maskrcnn_benchmark import _C

dev = torch.device("cuda")
torch.backends.cudnn.benchmark = True


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [  # (Cin, Cout, k, stride, H, W)   ResNet-50 at 800 x 1344, batch 2
    (256, 64, 1, 1, 200, 336), (64, 64, 3, 1, 200, 336), (64, 256, 1, 1, 200, 336),
    (512, 128, 1, 1, 100, 168), (128, 128, 3, 1, 100, 168), (128, 512, 1, 1, 100, 168),
    (1024, 256, 1, 1, 50, 84), (256, 256, 3, 1, 50, 84), (256, 1024, 1, 1, 50, 84),
    (2048, 512, 1, 1, 25, 42), (512, 512, 3, 1, 25, 42), (512, 2048, 1, 1, 25, 42),
]
for dtype in (torch.float32, torch.bfloat16):
    for (ci, co, k, s, H, W) in shapes:
        x = torch.randn(2, ci, H, W, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, k, k, device=dev, dtype=dtype) * 0.05).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device=dev, dtype=dtype)
        sc = torch.rand(co, device=dev) + 0.5
        sh = torch.randn(co, device=dev)
        pad = k // 2
        plain = lambda: torch.nn.functional.conv2d(x, w, None, s, pad)
        y = plain()
        try:
            fused = lambda: torch.miopen_convolution_relu(x, w, b, (s, s), (pad, pad), (1, 1), 1)
            yf = fused()
            ref = torch.relu(y.float() + b.float().view(1, -1, 1, 1))
            err = float((yf.float() - ref).abs().max())
            tf = t(fused)
        except Exception as e:  # noqa
            tf, err = float("nan"), str(e)[:80]
        tp = t(plain)
        bn = lambda: _C.frozen_bn_act_forward(y, sc, sh, None, True)
        try:
            tb = t(bn)
        except Exception as e:  # noqa
            tb = float("nan")
        print("%-8s Cin %4d Cout %4d k%d %3dx%3d  conv %7.1f us  conv+bias+relu fused %7.1f us  (err %s)  separate bn/relu pass %6.1f us" % (
            str(dtype).replace("torch.", ""), ci, co, k, H, W, tp, tf, err if isinstance(err, str) else "%.2e" % err, tb), flush=True)
