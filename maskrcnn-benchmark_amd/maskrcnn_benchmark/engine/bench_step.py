"""Builders shared by bench.py, tools/train_net.py and __graft_entry__.smoke(): config -> model +
overlapped SGD + schedule (+ DDP) and device-resident synthetic batches."""
import os

import torch

from maskrcnn_benchmark.config import cfg as _default_cfg
from maskrcnn_benchmark.data.synthetic import BatchCollator, SyntheticCOCODataset
from maskrcnn_benchmark.modeling.detector import build_detection_model
from maskrcnn_benchmark.solver import make_lr_scheduler

from .ddp_step import TrainStep, make_overlapped_sgd, wrap_data_parallel

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "configs")


def load_cfg(config_file, opts=()):
    cfg = _default_cfg.clone()
    if config_file:
        if not os.path.isabs(config_file) and not os.path.exists(config_file):
            config_file = os.path.join(CONFIG_DIR, config_file)
        cfg.merge_from_file(config_file)
    cfg.merge_from_list(list(opts))
    cfg.freeze()
    return cfg


def choose_layout(cfg, device, layout="auto"):
    """-> "nchw" | "backbone" | "all": which part of the detector runs on channels-last (NHWC) activations
    (GeneralizedRCNN.set_channels_last).  "auto" (overridable with DETOPS_LAYOUT): channels-last on the GPU for the FPN
    detectors in every precision — the configurations it was measured on (fp32, bf16 autocast: 97.6 -> 128.6 img/s, R-101 +
    DCN under fp16: 54.2 -> 64.3 img/s, profiles/r06p_half_precision_layouts.txt), with the tuned MIOpen find-db
    for the NHWC problem keys that ships in-tree (e2e_mask_rcnn_R_50_FPN_1x on one box: NCHW 38.0 ms per step, backbone + FPN
    channels-last 36.2, the heads as well 33.4: profiles/r06g_*); NCHW on the CPU and for non-FPN bodies (not measured)."""
    layout = os.environ.get("DETOPS_LAYOUT", layout) if layout == "auto" else layout
    if layout in ("nchw", "backbone", "all"):
        return layout
    if layout != "auto":
        raise ValueError("layout must be auto | nchw | backbone | all, got %r" % (layout,))
    ok = (torch.device(device).type == "cuda" and "FPN" in cfg.MODEL.BACKBONE.CONV_BODY
          and cfg.MODEL.META_ARCHITECTURE == "GeneralizedRCNN")
    if not ok:
        return "nchw"
    # deformable stages: the DCN layers read / write channels-last tensors in place (layers/dcn); measured for R-101 + DCN
    # under fp16 (profiles/r06p_*, r06q_*)
    return AUTO_LAYOUT_DCN if any(cfg.MODEL.RESNETS.STAGE_WITH_DCN) else AUTO_LAYOUT


AUTO_LAYOUT_DCN = "all"
AUTO_LAYOUT = "all"


def build_training(cfg, device, distributed=False, local_rank=0, overlap_optimizer=True, force_ddp=False, bucket_cap_mb=None,
                   layout="auto"):
    """-> (model (DDP-wrapped when distributed), optimizer, scheduler, TrainStep).  `force_ddp` wraps
    even at world size 1 (the process group must exist).  `layout`: see choose_layout."""
    model = build_detection_model(cfg).to(device)
    model.train()
    layout = choose_layout(cfg, device, layout)
    if layout != "nchw":
        model.set_channels_last(True, heads=layout == "all")
    model.layout = layout
    optimizer = make_overlapped_sgd(cfg, model)
    scheduler = make_lr_scheduler(cfg, optimizer)
    fp16 = cfg.DTYPE == "float16"
    if distributed or force_ddp:
        ids = [local_rank] if torch.device(device).type == "cuda" else None
        kw = {} if bucket_cap_mb is None else {"bucket_cap_mb": bucket_cap_mb}
        model = wrap_data_parallel(model, optimizer, device_ids=ids, overlap_optimizer=overlap_optimizer,
                                   force=force_ddp, **kw)
    step = TrainStep(model, optimizer, scheduler, dtype=cfg.DTYPE, device_type=torch.device(device).type)
    return model, optimizer, scheduler, step


def make_device_batches(cfg, device, images_per_gpu=2, num_batches=2, seed=0, height=None, width=None):
    """`num_batches` synthetic (ImageList, targets) pairs resident on `device`."""
    H = height or cfg.INPUT.MIN_SIZE_TRAIN[0]
    W = width or cfg.INPUT.MAX_SIZE_TRAIN
    ds = SyntheticCOCODataset(length=num_batches * images_per_gpu, height=H, width=W,
                              num_classes=cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES, with_masks=cfg.MODEL.MASK_ON, seed=seed)
    collate = BatchCollator(cfg.DATALOADER.SIZE_DIVISIBILITY)
    out = []
    for b in range(num_batches):
        images, targets, _ = collate([ds[b * images_per_gpu + i] for i in range(images_per_gpu)])
        out.append((images.to(device), [t.to(device) for t in targets]))
    return out


def smoke_train_step(device, config_file="e2e_mask_rcnn_R_50_FPN_1x.yaml", steps=2):
    """tiny forward + backward + update of the flagship detector (small image, full architecture)."""
    cfg = load_cfg(config_file, ["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 500, "MODEL.RPN.FPN_POST_NMS_TOP_N_TRAIN", 500,
                                 "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 128])
    torch.manual_seed(0)
    model, optimizer, scheduler, step = build_training(cfg, device)
    (images, targets), = make_device_batches(cfg, device, images_per_gpu=1, num_batches=1, height=256, width=320)
    losses = None
    for _ in range(steps):
        losses = step(images, targets)
    vals = {k: float(v.detach()) for k, v in losses.items()}
    assert all(v == v and abs(v) != float("inf") for v in vals.values()), "non-finite loss: %r" % vals
    print("smoke: one %s training step on %s ->" % (os.path.basename(config_file), device),
          {k: round(v, 4) for k, v in vals.items()})
    return vals
