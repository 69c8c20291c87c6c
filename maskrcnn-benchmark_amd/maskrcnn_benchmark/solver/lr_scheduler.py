"""WarmupMultiStepLR (reference solver/lr_scheduler.py:10-52): step decay by `gamma` at the
milestones, multiplied during the first `warmup_iters` iterations by a constant or a linear ramp
from `warmup_factor` to 1."""
from bisect import bisect_right

import torch


class WarmupMultiStepLR(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=1.0 / 3, warmup_iters=500,
                 warmup_method="linear", last_epoch=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        if warmup_method not in ("constant", "linear"):
            raise ValueError("Only 'constant' or 'linear' warmup_method accepted got {}".format(warmup_method))
        self.milestones = milestones
        self.gamma = gamma
        self.warmup_factor = warmup_factor
        self.warmup_iters = warmup_iters
        self.warmup_method = warmup_method
        super(WarmupMultiStepLR, self).__init__(optimizer, last_epoch)

    def _warmup(self):
        if self.last_epoch >= self.warmup_iters:
            return 1.0
        if self.warmup_method == "constant":
            return self.warmup_factor
        alpha = float(self.last_epoch) / self.warmup_iters
        return self.warmup_factor * (1 - alpha) + alpha

    def get_lr(self):
        f = self._warmup() * self.gamma ** bisect_right(self.milestones, self.last_epoch)
        return [base_lr * f for base_lr in self.base_lrs]
