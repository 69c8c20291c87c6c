"""do_train (reference engine/trainer.py:43-150): the training loop around TrainStep with the
reference's logging cadence (every 20 iterations) and loss reduction across ranks for the log."""
import datetime
import logging
import os
import time

import torch

from maskrcnn_benchmark import _C
from maskrcnn_benchmark.utils.comm import get_world_size, reduce_dict
from maskrcnn_benchmark.utils.metric_logger import MetricLogger

from .ddp_step import TrainStep


def reduce_loss_dict(loss_dict):
    """losses averaged over ranks on rank 0 (reference :18-40); a no-op for one process."""
    return reduce_dict(loss_dict, average=True) if get_world_size() > 1 else loss_dict


def do_train(cfg, model, data_loader, optimizer, scheduler, checkpointer, device, checkpoint_period, arguments,
             log_period=20):
    logger = logging.getLogger("maskrcnn_benchmark.trainer")
    logger.info("Start training")
    if hasattr(model, "comm_mode"):    # BucketedDataParallel: say which communication path runs (a silent fall-back is slower)
        logger.info("data parallel: comm_mode=%s (%s)", model.comm_mode, model.comm_note)
    meters = MetricLogger(delimiter="  ")
    start_iter = arguments["iteration"]
    max_iter = start_iter + len(data_loader)      # the loader holds the REMAINING iterations (reference: IterationBasedBatchSampler)
    model.train()
    step = TrainStep(model, optimizer, scheduler, dtype=cfg.DTYPE, device_type=torch.device(device).type)
    if os.environ.get("DETOPS_HIP_GRAPH", "0") == "1":
        # opt-in: replay the iteration from HIP graphs (engine/graph_step.py: one graph per input signature, at most
        # DETOPS_HIP_GRAPH_MAX of them, anything else runs eagerly) — for fixed-shape input pipelines on a host-bound step;
        # needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before the process starts
        if hasattr(model, "comm_mode") or torch.device(device).type != "cuda":
            logger.warning("DETOPS_HIP_GRAPH=1 ignored: single-process GPU training only")
        else:
            from .graph_step import GraphedTrainStep
            step = GraphedTrainStep(step, max_graphs=int(os.environ.get("DETOPS_HIP_GRAPH_MAX", "4")))
            logger.info("training iteration replayed from HIP graphs (up to %d input signatures)", step.max_graphs)
    start_training_time = time.time()
    end = time.time()
    for iteration, (images, targets, _) in enumerate(data_loader, start_iter):
        if any(len(t) < 1 for t in targets):
            logger.error("Iteration=%d || an image has no ground-truth box; skipped", iteration + 1)
            continue
        data_time = time.time() - end
        iteration = iteration + 1
        arguments["iteration"] = iteration
        images = images.to(device)
        targets = [t.to(device) for t in targets]
        loss_dict = step(images, targets)
        if iteration % log_period == 0 or iteration == max_iter:
            reduced = reduce_loss_dict(loss_dict)
            meters.update(loss=sum(v for v in reduced.values()), **reduced)
            # the losses are read back here anyway: also look at the segmented NMS's sticky status word (segments its
            # single launch gave up on and the repair launch redid — results are complete either way, this is the signal)
            repaired = _C.nms_repaired_segments(device, reset=True) if torch.device(device).type == "cuda" else 0
            if repaired:
                logger.warning("iter %d: %d NMS segment(s) since the last log line were redone by the repair launch "
                               "(the single-launch kernel's waits timed out: compute units held by other work)",
                               iteration, repaired)
        batch_time = time.time() - end
        end = time.time()
        meters.update(time=batch_time, data=data_time)
        if iteration % log_period == 0 or iteration == max_iter:
            eta = str(datetime.timedelta(seconds=int(meters.time.global_avg * (max_iter - iteration))))
            mem = torch.cuda.max_memory_allocated() / 1024.0 / 1024.0 if torch.cuda.is_available() else 0.0
            logger.info(meters.delimiter.join(["eta: {}".format(eta), "iter: {}".format(iteration), str(meters),
                                               "lr: {:.6f}".format(optimizer.param_groups[0]["lr"]),
                                               "max mem: {:.0f}".format(mem)]))
        if checkpointer is not None and checkpoint_period > 0 and iteration % checkpoint_period == 0:
            checkpointer.save("model_{:07d}".format(iteration), **arguments)
        if iteration == max_iter:
            if checkpointer is not None:
                checkpointer.save("model_final", **arguments)
            break
    total = time.time() - start_training_time
    logger.info("Total training time: {} ({:.4f} s / it)".format(str(datetime.timedelta(seconds=total)),
                                                                 total / max(max_iter, 1)))
