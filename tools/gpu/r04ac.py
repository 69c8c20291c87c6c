import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "maskrcnn-benchmark_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from maskrcnn_benchmark import _C as C, _lib
from opbench import dev_time_us
Cc, H, W = 128, 100, 168
dt = torch.float16
off = (torch.randn(2, 18, H, W, device="cuda") * 2).to(dt)
msk = torch.rand(2, 9, H, W, device="cuda").to(dt)
go = torch.randn(2, Cc, H, W, device="cuda").to(dt)
gT = C._to_nhwc(go)
g9 = (3, 3, 1, 1, 1, 1, 1, 1, 1)
for v in (0, 1):
    _lib.tuning_set("dcn_ell_build", v)
    us = dev_time_us(lambda: C._transposed_sample(gT, off, msk, 2, Cc, H, W, Cc, g9), 20)
    print("dcn_ell_build=%d transposed_sample %.1f us" % (v, us))
_lib.tuning_set("dcn_ell_build", 0)
