"""BalancedPositiveNegativeSampler (reference modeling/balanced_positive_negative_sampler.py:5-68).

Picks, per image, up to `batch_size_per_image * positive_fraction` random positives (label >= 1)
and fills the rest of the batch with random negatives (label == 0); label -1 is ignored.

The reference draws the subsets with `nonzero` + `randperm(numel)` — four host syncs per image.
Here a subset is drawn by ranking i.i.d. uniform keys, entirely on device: the `k` smallest keys
among the candidates are a uniformly random `k`-subset, which is the same distribution.
`__call__` returns the reference's two lists of boolean masks; `sample_fixed` returns a
fixed-length index set (positives first) for the padded training path.
"""
import os

import torch

_THRESHOLD_SELECT = os.environ.get("DETOPS_SAMPLER", "") == "topk"


class BalancedPositiveNegativeSampler(object):
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction

    def _masks(self, labels):
        """labels [..., n] -> (pos_mask, neg_mask) bool, same shape, sampled per trailing row."""
        B = self.batch_size_per_image
        max_pos = int(B * self.positive_fraction)
        n = labels.shape[-1]
        from maskrcnn_benchmark import _C
        if _C.on_device(labels) and labels.dim() == 2 and B <= 512:
            # hashed-key threshold filter + a small per-image sort (csrc/targets.hip) instead of four argsorts
            return _C.sample_labels(labels, B, max_pos)
        pos = labels >= 1
        neg = labels == 0
        n_pos = pos.sum(dim=-1, keepdim=True).clamp(max=max_pos)
        n_neg = neg.sum(dim=-1, keepdim=True).clamp(max=B).minimum(B - n_pos)
        if _THRESHOLD_SELECT and n >= B:
            # EXPERIMENTAL (DETOPS_SAMPLER=topk, off by default, not yet measured): the k smallest keys
            # are found with one top-k per class and a threshold compare instead of two full argsorts
            # per class (4 sorts of [N, 268,569] for the RPN = 1.6 ms/step of merge-sort launches).
            # float64 keys: a tie AT the threshold has probability ~1e-10 instead of ~1e-2.
            u = torch.rand(labels.shape, device=labels.device, dtype=torch.float64)

            def pick(cand, k_dev, k_max):
                keys = torch.where(cand, u, u + 2)
                small = keys.topk(k_max, dim=-1, largest=False, sorted=True).values     # ascending
                thr = small.gather(-1, (k_dev - 1).clamp(min=0))
                return cand & (keys <= thr) & (k_dev > 0)

            return pick(pos, n_pos, max_pos), pick(neg, n_neg, B)
        u = torch.rand(labels.shape, device=labels.device)
        # rank of each candidate's key inside its class (0 = smallest); non-candidates pushed last
        rank_pos = torch.where(pos, u, u + 2).argsort(dim=-1).argsort(dim=-1)
        rank_neg = torch.where(neg, u, u + 2).argsort(dim=-1).argsort(dim=-1)
        return pos & (rank_pos < n_pos), neg & (rank_neg < n_neg)

    def __call__(self, matched_idxs):
        pos_idx, neg_idx = [], []
        for labels in matched_idxs:
            p, q = self._masks(labels)
            pos_idx.append(p)
            neg_idx.append(q)
        return pos_idx, neg_idx

    def sample_fixed(self, labels):
        """labels [N, n] -> (index [N, B] int64, valid [N, B] bool).  Slots are ordered positives
        first, then negatives; `valid` is False for slots that could not be filled (fewer than B
        candidates).  No host synchronisation."""
        B = self.batch_size_per_image
        from maskrcnn_benchmark import _C
        if _C.on_device(labels) and labels.dim() == 2 and B <= 512:
            _, _, idx, valid = _C.sample_labels(labels, B, int(B * self.positive_fraction), with_list=True)
            return idx, valid
        pos_mask, neg_mask = self._masks(labels)
        key = pos_mask.to(torch.float32) * 2 + neg_mask.to(torch.float32)
        key = key + torch.rand(labels.shape, device=labels.device) * 0.5
        k = min(B, labels.shape[-1])
        top, idx = key.topk(k, dim=-1, sorted=True)
        valid = top >= 1
        if k < B:
            pad = B - k
            idx = torch.cat([idx, idx.new_zeros(idx.shape[:-1] + (pad,))], dim=-1)
            valid = torch.cat([valid, valid.new_zeros(valid.shape[:-1] + (pad,))], dim=-1)
        return idx, valid
