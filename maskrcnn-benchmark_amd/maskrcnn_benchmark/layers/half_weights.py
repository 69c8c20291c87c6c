"""Half-precision copies of the weights for a mixed-precision (autocast) forward, made by ONE multi-tensor launch (round 6).

Under torch.autocast every convolution / linear layer casts its fp32 weight to fp16 / bf16 on its own — one `to_copy` launch
per layer and forward — and autograd casts every half weight gradient back to fp32 — one more launch per layer and backward:
R-50-FPN Mask R-CNN 113 + 100 launches (1.0 ms of a 15.4 ms bf16 step), R-101 + DCN 224 + 211 (2.0 ms of 28 ms,
profiles/r06_mixed_precision_traces.txt), each ~4.5 us of device time for a few KB of payload, and as many host-side
dispatches in a step that is host-bound.  The reference has the same structure (apex O1 patches the functional ops and caches
the casts per iteration, engine/trainer.py:66-79 + apex.amp): this is not a parity item, it is launch count.

Here the half copies of all weights of one STAGE of the model (ResNet stem / layer1-4, FPN, RPN head, box head, mask head: nine
groups) come from one `torch._foreach_copy_` through one autograd node per stage, and the node's backward turns the stage's half
gradients into fp32 gradients of the masters in one `_foreach_copy_` again.  The values are exactly autocast's (the same round-to-nearest cast of the same fp32
master, the same half gradient widened to fp32), the optimizer / GradScaler / checkpoint code sees nothing: parameters stay the
fp32 masters in `module._parameters`; for the duration of one forward the modules' `weight` ATTRIBUTE resolves to the half
copy (an instance attribute shadows `nn.Module.__getattr__`'s parameter look-up), and `remove()` drops the shadows.

WHEN a stage's node is created matters.  The autograd engine runs ready nodes in order of creation, latest first: one node for
the whole model, created at the start of the forward, would run LAST — every weight gradient would appear in one burst at the
end of the backward pass (measured: two ~0.6 ms host gaps around that burst in the device-bound bf16 step; and
`BucketedDataParallel`, which all-reduces a bucket as soon as its gradients exist, would lose its overlap).  A stage's node is
therefore created by a forward pre-hook of the stage's module, right before the stage runs: it is younger than every node of
the earlier stages, so in the backward pass it runs as soon as the stage's last weight gradient exists — before the engine moves
on to the earlier stages — and the stage's fp32 gradients reach the optimizer / the bucket hooks at that point."""
import os

import torch
from torch import nn

ENABLED = os.environ.get("DETOPS_HALF_WEIGHTS", "1") != "0"


class _CastAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dtype, *masters):
        ctx.set_materialize_grads(False)
        outs = [torch.empty_like(m, dtype=dtype) for m in masters]     # preserve_format: a channels-last weight stays channels-last
        torch._foreach_copy_(outs, [m.detach() for m in masters])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        idx = [i for i, g in enumerate(grads) if g is not None and ctx.needs_input_grad[i + 1]]
        out = [None] * (len(grads) + 1)
        if idx:
            src = [grads[i] for i in idx]
            dst = [torch.empty_like(g, dtype=torch.float32) for g in src]
            torch._foreach_copy_(dst, src)
            for i, d in zip(idx, dst):
                out[i + 1] = d
        return tuple(out)


def _weight_modules():
    from .dcn.deform_conv_module import DeformConv, ModulatedDeformConv
    return (nn.Conv2d, nn.ConvTranspose2d, nn.Linear, DeformConv, ModulatedDeformConv)


class _StageHook(object):
    """forward pre-hook of a stage's module (an object, not a closure: deep copies and pickles of the model keep a hook that
    points at THEIR HalfWeights)"""

    def __init__(self, owner, stage):
        self.owner, self.stage = owner, stage

    def __call__(self, module, inputs):
        self.owner._install_stage(self.stage)


class HalfWeights(object):
    """The fp32 `weight` parameters of a model's convolution / linear / deformable-convolution modules, grouped by stage, and
    their per-forward half copies.  `install(dtype)` arms the stages' pre-hooks ... forward ... `remove()`; nothing else of the
    model changes."""

    # module-path prefixes that are split one level further (their children are the stages)
    SPLIT = ("backbone.body",)

    def __init__(self, model):
        self.model = model
        self.enabled = ENABLED
        self.entries = None          # [(module, fp32 parameter)]
        self.groups = None           # stage name -> [(module, fp32 parameter)]
        self.installed = False
        self._dtype = None           # not None: inside a forward that uses the copies
        self._done = set()
        self._hooks = []

    @classmethod
    def _stage_of(cls, name):
        parts = name.split(".")
        depth = 3 if ".".join(parts[:2]) in cls.SPLIT else 2
        return ".".join(parts[:depth])

    def _collect(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        kinds = _weight_modules()
        self.entries, self.groups = [], {}
        for name, m in self.model.named_modules():
            if isinstance(m, kinds):
                p = m._parameters.get("weight")
                if p is not None and p.dtype == torch.float32 and p.dim() >= 2:
                    self.entries.append((m, p))
                    self.groups.setdefault(self._stage_of(name), []).append((m, p))
        lookup = dict(self.model.named_modules())
        for stage in self.groups:
            anchor = lookup[stage]
            self._hooks.append(anchor.register_forward_pre_hook(_StageHook(self, stage)))

    def _stale(self):
        return self.entries is None or any(m._parameters.get("weight") is not p for m, p in self.entries)

    def usable(self, x):
        """an autocast forward in fp16 / bf16 (the GPU in production; the CPU suite drives the same code under CPU autocast)"""
        kind = x.device.type
        return (self.enabled and kind in ("cuda", "cpu") and torch.is_autocast_enabled(kind)
                and torch.get_autocast_dtype(kind) in (torch.float16, torch.bfloat16))

    def install(self, dtype):
        """arm the stages' pre-hooks for this forward: each stage casts its weights when it is about to run"""
        if self._stale():
            self._collect()
        if not self.entries:
            return False
        self._dtype = dtype
        self._done = set()
        self.installed = True
        return True

    def _install_stage(self, stage):
        dtype = self._dtype
        if dtype is None or stage in self._done:
            return
        self._done.add(stage)
        entries = self.groups[stage]
        live = [p for _, p in entries if p.requires_grad]
        frozen = [p for _, p in entries if not p.requires_grad]
        halves = {}
        if live:
            for p, h in zip(live, _CastAll.apply(dtype, *live)):
                halves[id(p)] = h
        if frozen:
            with torch.no_grad():
                outs = [torch.empty_like(p, dtype=dtype) for p in frozen]
                torch._foreach_copy_(outs, frozen)
            for p, h in zip(frozen, outs):
                halves[id(p)] = h
        for m, p in entries:
            m.__dict__["weight"] = halves[id(p)]

    def remove(self):
        if self.installed:
            for m, _ in self.entries:
                m.__dict__.pop("weight", None)
            self.installed = False
        self._dtype = None
        self._done = set()
