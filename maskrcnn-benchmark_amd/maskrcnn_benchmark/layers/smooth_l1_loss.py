"""smooth_l1_loss with the beta parameter (reference layers/smooth_l1_loss.py:6-16)."""
import torch


def smooth_l1_loss(input, target, beta=1. / 9, size_average=True):
    n = torch.abs(input - target)
    loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if size_average:
        return loss.mean()
    return loss.sum()
