"""cProfile of the HOST side of one deformable-conv layer (forward + deform_conv_backward_all) at the cfg-5 layer3 shape, fp16:
where the ~270 us of Python per layer and step go."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "maskrcnn-benchmark_amd")]
import torch
from maskrcnn_benchmark import _C as C
dt = torch.float16
Cc, H, W = 256, 50, 84
x = torch.randn(2, Cc, H, W, device="cuda").to(dt)
off = (torch.randn(2, 18, H, W, device="cuda") * 0.5).to(dt)
w = (torch.randn(Cc, Cc, 3, 3, device="cuda") / 48).to(dt)
go = torch.randn(2, Cc, H, W, device="cuda").to(dt)
out = torch.empty(2, Cc, H, W, device="cuda", dtype=dt)
e = torch.empty(0, device="cuda", dtype=dt)
def fwd():
    keep = []
    C.deform_conv_forward(x, w, off, out, e, e, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 2, keep=keep)
    return keep[0] if keep else None
def bwd(saved):
    return C.deform_conv_backward_all(x, off, None, w, go, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, saved=saved)
for _ in range(20):
    bwd(fwd())
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
saved = [fwd() for _ in range(N)]
t1 = time.perf_counter()
for s in saved: bwd(s)
t2 = time.perf_counter()
torch.cuda.synchronize()
print("host per call: forward %.1f us, backward_all %.1f us" % ((t1 - t0) / N * 1e6, (t2 - t1) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    bwd(fwd())
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
