"""R-50 conv stack (no norm layers: every 3x3 / 1x1 / strided conv of the backbone at the bench's 2 x 3 x 800 x 1344 input) forward +
backward under autocast(bf16 / fp16) and in fp32: NCHW vs channels_last — device time per pass and kernel launches (torch profiler)."""
import sys, time, torch, torch.nn as nn
from torch.profiler import profile, ProfilerActivity

class Bottleneck(nn.Module):
    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.c1 = nn.Conv2d(cin, mid, 1, stride, bias=False)   # stride in 1x1 (caffe2 style, STRIDE_IN_1X1)
        self.c2 = nn.Conv2d(mid, mid, 3, 1, 1, bias=False)
        self.c3 = nn.Conv2d(mid, cout, 1, bias=False)
        self.down = nn.Conv2d(cin, cout, 1, stride, bias=False) if (cin != cout or stride != 1) else None
    mixed = False   # round 5: 1x1 convolutions and block boundaries channels-last, the 3x3 convolution NCHW (Winograd); the two
                    # layout changes around it are explicit copies here (the fused FrozenBN kernels would do them on the way)
    def forward(self, x):
        if Bottleneck.mixed:
            t = torch.relu(self.c1(x)).contiguous(memory_format=torch.contiguous_format)
            t = torch.relu(self.c2(t)).contiguous(memory_format=torch.channels_last)
            y = self.c3(t)
            return torch.relu(y + (self.down(x) if self.down is not None else x))
        y = self.c3(torch.relu(self.c2(torch.relu(self.c1(x)))))
        return torch.relu(y + (self.down(x) if self.down is not None else x))

def body():
    layers = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.ReLU(), nn.MaxPool2d(3, 2, 1)]
    cin = 64
    for mid, cout, n, stride in ((64, 256, 3, 1), (128, 512, 4, 2), (256, 1024, 6, 2), (512, 2048, 3, 2)):
        for i in range(n):
            layers.append(Bottleneck(cin, mid, cout, stride if i == 0 else 1)); cin = cout
    return nn.Sequential(*layers)

torch.manual_seed(0)
net = body().cuda()
x0 = torch.randn(2, 3, 800, 1344, device="cuda")
for dt in (torch.float32, torch.bfloat16):
    for fmt in (torch.contiguous_format, torch.channels_last, "mixed"):
        Bottleneck.mixed = fmt == "mixed"
        if fmt == "mixed":
            m = net.to(memory_format=torch.channels_last)
            for mod in m.modules():
                if isinstance(mod, Bottleneck):
                    mod.c2.to(memory_format=torch.contiguous_format)
            x = x0.to(memory_format=torch.channels_last).requires_grad_(True)
        else:
            m = net.to(memory_format=fmt)
            x = x0.to(memory_format=fmt).requires_grad_(True)
        def step():
            with torch.autocast("cuda", dtype=dt, enabled=dt != torch.float32):
                y = m(x)
            y.float().mean().backward()
        for _ in range(4): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step(); torch.cuda.synchronize()
        ev = [e for e in prof.key_averages() if e.device_time_total > 0]
        n = sum(e.count for e in ev)
        tr = sum(e.count for e in ev if "transpose" in e.key.lower())
        trt = sum(e.device_time_total for e in ev if "transpose" in e.key.lower()) / 1e3
        dev = sum(e.device_time_total for e in ev) / 1e3
        cp = sum(e.device_time_total for e in ev if "copy" in e.key.lower()) / 1e3
        name = "mixed" if fmt == "mixed" else ("channels_last" if fmt == torch.channels_last else "nchw")
        print("%-8s %-14s wall %.2f ms/step, device-busy %.2f ms, launches %d (transposes %d = %.2f ms; copy kernels %.2f ms)" %
              (str(dt)[6:], name, ms, dev, n, tr, trt, cp), flush=True)
        if fmt == "mixed":
            for e in sorted(ev, key=lambda e: -e.device_time_total)[:14]:
                print("      %8.3f ms  %4d x  %s" % (e.device_time_total / 1e3, e.count, e.key[:110]))
