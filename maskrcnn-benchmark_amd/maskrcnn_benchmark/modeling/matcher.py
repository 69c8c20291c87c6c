"""Matcher: assign each prediction to a ground-truth element by IoU (reference
modeling/matcher.py:5-112).

`matches[n]` = index of the best ground truth, or BELOW_LOW_THRESHOLD (-1) / BETWEEN_THRESHOLDS
(-2).  With `allow_low_quality_matches`, every prediction that attains the maximum IoU of some
ground truth (ties included) gets its arg-max ground truth back even if that IoU is below the
thresholds.

Unlike the reference this implementation never calls `nonzero` (a device->host sync on the
`[M, 268569]` RPN matrix every iteration): the low-quality rule is evaluated
with a column-wise `any`.  It also accepts a batch `[B, M, N]` with a row-validity mask so that
images with different numbers of ground-truth boxes are matched in one pass.
"""
import torch


class Matcher(object):
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False):
        assert low_threshold <= high_threshold
        self.high_threshold = high_threshold
        self.low_threshold = low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix, row_valid=None):
        """match_quality_matrix [M,N] (or [B,M,N]); row_valid optional bool [M] (or [B,M]) marking
        real ground-truth rows (padding rows must hold a quality < 0, e.g. -1)."""
        q = match_quality_matrix
        if q.numel() == 0:
            if q.shape[-2] == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
            raise ValueError("No proposal boxes available for one of the images during training")
        matched_vals, matches = q.max(dim=-2)
        all_matches = matches
        below = matched_vals < self.low_threshold
        between = (matched_vals >= self.low_threshold) & (matched_vals < self.high_threshold)
        matches = torch.where(below, torch.full_like(matches, Matcher.BELOW_LOW_THRESHOLD), matches)
        matches = torch.where(between, torch.full_like(matches, Matcher.BETWEEN_THRESHOLDS), matches)
        if self.allow_low_quality_matches:
            best_per_gt = q.max(dim=-1, keepdim=True).values
            is_best = q == best_per_gt
            if row_valid is not None:
                is_best = is_best & row_valid.unsqueeze(-1)
            matches = torch.where(is_best.any(dim=-2), all_matches, matches)
        return matches
