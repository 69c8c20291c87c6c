"""cProfile of the host side of the training step (python tools/host_profile.py [steps]) — where the Python time goes."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd")):
    sys.path.insert(0, p)
import torch
from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
cfg = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml", [])
torch.manual_seed(1234)
model, optimizer, scheduler, step = build_training(cfg, dev, False, 0)
batches = make_device_batches(cfg, dev, images_per_gpu=2, num_batches=2, seed=0)
for i in range(8):
    step(*batches[i % 2]); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(*batches[i % 2])
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:70]))
