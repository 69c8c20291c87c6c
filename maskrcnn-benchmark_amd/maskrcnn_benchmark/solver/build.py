"""make_optimizer / make_lr_scheduler (reference solver/build.py:7-31).

Same hyper-parameter rule as the reference — SGD with momentum; biases get
`BASE_LR * BIAS_LR_FACTOR` and `WEIGHT_DECAY_BIAS` — but the parameters are collected into TWO
groups (weights, biases) instead of one single-tensor group per parameter (84 groups -> an 84-step
Python loop in `step()`), and the update runs as multi-tensor (`foreach`) kernels.
"""
import torch

from .lr_scheduler import WarmupMultiStepLR


def make_optimizer(cfg, model):
    S = cfg.SOLVER
    weights, biases = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (biases if "bias" in name else weights).append(p)
    groups = []
    if weights:
        groups.append({"params": weights, "lr": S.BASE_LR, "weight_decay": S.WEIGHT_DECAY})
    if biases:
        groups.append({"params": biases, "lr": S.BASE_LR * S.BIAS_LR_FACTOR, "weight_decay": S.WEIGHT_DECAY_BIAS})
    return torch.optim.SGD(groups, S.BASE_LR, momentum=S.MOMENTUM, foreach=True)


def make_lr_scheduler(cfg, optimizer):
    S = cfg.SOLVER
    return WarmupMultiStepLR(optimizer, S.STEPS, S.GAMMA, warmup_factor=S.WARMUP_FACTOR,
                             warmup_iters=S.WARMUP_ITERS, warmup_method=S.WARMUP_METHOD)
