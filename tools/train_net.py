#!/usr/bin/env python
"""tools/train_net.py — same command line as the reference's (tools/train_net.py:133-196):

    python tools/train_net.py --config-file configs/e2e_mask_rcnn_R_50_FPN_1x.yaml [KEY VALUE ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_net.py ...

One process per GPU; `--local_rank` is accepted for the reference's launcher but LOCAL_RANK from the
environment (torchrun) wins.  Data is the synthetic COCO-shaped generator (no network / datasets)."""
import argparse
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maskrcnn-benchmark_amd"))
# two busy streams per rank (compute, RCCL): two HIP hardware queues — see bench.py::pin_hip_queues for the measurement;
# read by the HIP runtime at initialisation, an explicit setting in the environment wins
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # kernel arguments in device memory (bench.py::pin_hip_queues)

import torch  # noqa: E402

from maskrcnn_benchmark.data import make_data_loader  # noqa: E402
from maskrcnn_benchmark.engine.bench_step import build_training, load_cfg  # noqa: E402
from maskrcnn_benchmark.engine.trainer import do_train  # noqa: E402
from maskrcnn_benchmark.utils.checkpoint import DetectronCheckpointer  # noqa: E402
from maskrcnn_benchmark.utils.comm import get_rank, synchronize  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="MI355X-native Mask R-CNN training")
    ap.add_argument("--config-file", default="e2e_mask_rcnn_R_50_FPN_1x.yaml", metavar="FILE")
    ap.add_argument("--local_rank", type=int, default=0)
    ap.add_argument("--skip-test", dest="skip_test", action="store_true")
    ap.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE config overrides")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", args.local_rank))
    distributed = world > 1
    cfg = load_cfg(args.config_file, args.opts or [])
    device = torch.device(cfg.MODEL.DEVICE, local_rank) if cfg.MODEL.DEVICE == "cuda" else torch.device(cfg.MODEL.DEVICE)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    if distributed:
        torch.distributed.init_process_group(backend="nccl" if device.type == "cuda" else "gloo", init_method="env://")
        synchronize()
    logging.basicConfig(level=logging.INFO if get_rank() == 0 else logging.WARNING,
                        format="%(asctime)s %(name)s %(levelname)s: %(message)s")
    logger = logging.getLogger("maskrcnn_benchmark")
    logger.info("Using %d GPUs", world)
    logger.info("Running with config:\n%s", cfg.dump())
    model, optimizer, scheduler, _ = build_training(cfg, device, distributed, local_rank)
    # which communication path the data-parallel wrapper took ("direct" RCCL on its side stream, or the ProcessGroupNCCL
    # fallback with the reason) and the activation layout — a silent fall-back costs ~1.3 ms per step (profiles/r05b_ddp_paths.txt)
    logger.info("data parallel: comm_mode=%s (%s); layout=%s", getattr(model, "comm_mode", "single process"),
                getattr(model, "comm_note", "-"), getattr(getattr(model, "module", model), "layout", "nchw"))
    out_dir = cfg.OUTPUT_DIR if cfg.OUTPUT_DIR != "." else ""
    if out_dir and get_rank() == 0:
        os.makedirs(out_dir, exist_ok=True)      # reference tools/train_net.py:166-168
    checkpointer = DetectronCheckpointer(cfg, model, optimizer, scheduler, out_dir)
    arguments = {"iteration": 0}
    arguments.update(checkpointer.load(cfg.MODEL.WEIGHT or None))
    # the reference's IterationBasedBatchSampler yields MAX_ITER - start_iter batches (data/samplers/iteration_based_batch_sampler.py)
    remaining = cfg.SOLVER.MAX_ITER - arguments["iteration"]
    if remaining <= 0:
        logger.info("iteration %d of %d: nothing left to train", arguments["iteration"], cfg.SOLVER.MAX_ITER)
    else:
        loader = make_data_loader(cfg, is_train=True, is_distributed=distributed, start_iter=arguments["iteration"],
                                  length=remaining * max(cfg.SOLVER.IMS_PER_BATCH // world, 1))
        do_train(cfg, model, loader, optimizer, scheduler, checkpointer, device, cfg.SOLVER.CHECKPOINT_PERIOD, arguments)
    if hasattr(model, "close"):       # BucketedDataParallel: its own RCCL communicator and side stream go before the process group
        model.close()
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
