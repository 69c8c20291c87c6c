"""Probe (round 6, late): weight gradient of the 1 x 1 stride-1 convolutions as ONE library GEMM (dW = dY^T X over the
channels-last [N H W, C] views, no zero-fill launch, no split-K atomics) against MIOpen's wrw path (igemm_wrw + its
SubTensorOpWithScalar1d zero-fill), on the shapes of the R-50-FPN step at 2 x 800 x 1344."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "maskrcnn-benchmark_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
print("miopen db:", bench.setup_miopen_db(None))        # the shipped find-db (before torch is imported)
import torch
dev = torch.device("cuda")


def t(fn, n=50):
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


shapes = [  # (Cin, Cout, H, W, count per step)
    (256, 128, 200, 336, 0), (512, 128, 100, 168, 3), (128, 512, 100, 168, 4),
    (1024, 256, 50, 84, 5), (256, 1024, 50, 84, 6), (2048, 512, 25, 42, 2), (512, 2048, 25, 42, 3),
    (256, 256, 200, 336, 1), (512, 256, 100, 168, 1), (1024, 256, 50, 84, 1), (2048, 256, 25, 42, 1),     # FPN laterals
    (256, 3, 200, 336, 1), (256, 12, 200, 336, 1), (256, 3, 100, 168, 1), (256, 12, 100, 168, 1), (256, 81, 28, 28, 0),
]
tot_m = tot_g = 0.0
for dtype in (torch.float32, torch.bfloat16):
    for (ci, co, H, W, cnt) in shapes:
        x = torch.randn(2, ci, H, W, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, 1, 1, device=dev, dtype=dtype) * 0.05).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(2, co, H, W, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
        mi = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False))[1]
        x2 = x.permute(0, 2, 3, 1).reshape(-1, ci)
        d2 = dy.permute(0, 2, 3, 1).reshape(-1, co)
        assert x2.data_ptr() == x.data_ptr() and d2.data_ptr() == dy.data_ptr()
        gm = lambda: torch.mm(d2.t(), x2)
        a, b = mi().float().reshape(co, ci), gm().float()
        err = float((a - b).abs().max() / a.abs().max())
        tm, tg = t(mi), t(gm)
        if dtype == torch.float32:
            tot_m += cnt * tm; tot_g += cnt * tg
        print("%-8s Cin %4d Cout %4d %3dx%3d x%d  MIOpen wrw %7.1f us   mm(dY^T, X) %7.1f us   rel diff %.1e" % (
            str(dtype).replace("torch.", ""), ci, co, H, W, cnt, tm, tg, err), flush=True)
print("fp32, weighted by the calls per step: MIOpen %.2f ms, GEMM %.2f ms" % (tot_m / 1e3, tot_g / 1e3))
