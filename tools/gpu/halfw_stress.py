"""stress: three Mask R-CNN training iterations under autocast in a fresh process; prints every step's losses (diagnosis of a
one-off NaN seen in tests/test_model_gpu.py::test_half_weights_*)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "maskrcnn-benchmark_amd"))
from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg, make_device_batches
dtype = sys.argv[1] if len(sys.argv) > 1 else "bfloat16"
cfg = load_cfg("e2e_mask_rcnn_R_50_FPN_1x.yaml", ["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 300, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 300,
                                                  "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 64, "DTYPE", dtype, "SOLVER.BASE_LR", 0.002])
(images, targets), = make_device_batches(cfg, "cuda", images_per_gpu=1, num_batches=1, height=192, width=256)
torch.manual_seed(0)
model, opt, sched, step = build_training(cfg, "cuda")
out = []
for _ in range(3):
    ld = step(images, targets)
    out.append([round(float(ld[k].detach()), 4) for k in sorted(ld)])
bad = any(v != v for row in out for v in row)
print("NAN" if bad else "ok", dtype, os.environ.get("DETOPS_HALF_WEIGHTS", "1"), out, model.roi_heads.mask.last_slots)
