"""Probe (round 6, late): the 3 x 3 convolutions of the fp32 step, NCHW (Winograd available) against channels-last (NHWC
implicit GEMM), per direction, MIOpen Find on (cudnn.benchmark) for both layouts.  Question: would a layout exception for the
large 3 x 3 convolutions (FPN output / RPN head at P2) pay for the layout changes around it?"""
import torch
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [  # (N, C, H, W, calls fwd, calls bwd-data, calls wrw, what)
    (2, 256, 200, 336, 2, 2, 2, "FPN out / RPN conv, P2"), (2, 256, 100, 168, 2, 2, 2, "P3"), (2, 256, 50, 84, 2, 2, 2, "P4"),
    (2, 256, 25, 42, 2, 2, 2, "P5"), (2, 64, 200, 336, 3, 0, 0, "layer1 conv2 (frozen)"), (2, 128, 100, 168, 4, 4, 4, "layer2 conv2"),
    (2, 256, 50, 84, 6, 6, 6, "layer3 conv2"), (2, 512, 25, 42, 3, 3, 3, "layer4 conv2"), (256, 256, 14, 14, 4, 4, 4, "mask head"),
]
tot = {"nchw": 0.0, "nhwc": 0.0, "best": 0.0}
for (N, C, H, W, nf, nb, nw, what) in shapes:
    row = {}
    for name, fmt in (("nchw", torch.contiguous_format), ("nhwc", torch.channels_last)):
        x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=fmt)
        w = (torch.randn(C, C, 3, 3, device=dev) * 0.02).contiguous(memory_format=fmt)
        dy = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=fmt)
        f = t(lambda: torch.nn.functional.conv2d(x, w, None, 1, 1))
        b = t(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False)))
        g = t(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False)))
        tr = t(lambda: x.contiguous(memory_format=torch.channels_last if fmt == torch.contiguous_format else torch.contiguous_format))
        row[name] = (f, b, g, tr)
        tot[name] += nf * f + nb * b + nw * g
    best = sum(n * min(row["nchw"][i], row["nhwc"][i]) for i, n in enumerate((nf, nb, nw)))
    tot["best"] += best
    print("%-26s C %3d %3dx%3d N %3d | NCHW fwd %7.1f bwd %7.1f wrw %7.1f | NHWC fwd %7.1f bwd %7.1f wrw %7.1f | layout change %6.1f us" % (
        what, C, H, W, N, *row["nchw"][:3], *row["nhwc"][:3], row["nchw"][3]), flush=True)
print("per step (us): all NCHW %.0f | all NHWC %.0f | best of both per direction, layout changes not counted %.0f" % (tot["nchw"], tot["nhwc"], tot["best"]))
