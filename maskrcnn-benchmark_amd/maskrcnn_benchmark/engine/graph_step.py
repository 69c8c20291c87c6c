"""A training iteration replayed from a HIP graph (round 6).

The iteration of this package never reads anything back from the device (fixed-length proposal / target
tensors, device-side counts), so forward + backward + optimizer can be captured ONCE and replayed: the host then enqueues one
graph launch instead of ~1,000-1,800 kernel launches.  That pays exactly where the step is HOST-bound — R-101 + deformable
convolutions under fp16 (BASELINE configs[4]): 36.5 ms eager vs 25.2 ms replayed on a slow-host box, tools/graph_probe.py —
and nothing where the device is the limit (the fp32 headline step, bf16).  The reference has no counterpart (its trainer loop,
engine/trainer.py:54-89, launches every kernel from Python and synchronises ~66 times per iteration).

What a capture freezes are the launch ARGUMENTS; everything that must change from one iteration to the next therefore lives in
device memory:
  * inputs: images and targets are copied into the static tensors the graph was captured on (one graph per input SIGNATURE:
    padded image shape, per-image sizes, number of ground-truth boxes per image, mask sizes — a fixed-shape pipeline such as
    the synthetic benchmark has two; a dataset with free image sizes would need one per shape, so the cache is bounded and
    anything beyond it runs eagerly);
  * learning rate: OverlappedSGD.lr_tensors (torch._fused_sgd_'s tensor-lr form), written before every replay from the
    scheduler's value;
  * the samplers' randomness: `_C.GRAPH_SEED`, an int64 device word the captured step increments first thing; the sampler
    kernels mix it into their seed when they run (detops_sample_labels_dseed);
  * loss scaling (fp16): torch.amp.GradScaler's scale / growth tracker are device tensors already, and the fused SGD takes
    grad_scale / found_inf from the device.
Single-process only: the data-parallel wrapper enqueues RCCL calls and stream waits from autograd hooks, which this class does
not capture (wrap_data_parallel + GraphedTrainStep raises)."""
import collections
import logging
import os

import torch

from maskrcnn_benchmark import _C

log = logging.getLogger("maskrcnn_benchmark.graph_step")


def _target_tensors(t):
    """every device tensor of one BoxList, in a fixed order (boxes, then the fields by name; masks: the BinaryMaskList's tensor)"""
    out = [t.bbox]
    for k in sorted(t.extra_fields):
        v = t.extra_fields[k]
        if torch.is_tensor(v):
            out.append(v)
        elif hasattr(v, "instances") and hasattr(v.instances, "masks"):    # SegmentationMask(BinaryMaskList)
            out.append(v.instances.masks)
        elif hasattr(v, "masks") and torch.is_tensor(v.masks):
            out.append(v.masks)
        else:
            raise TypeError("GraphedTrainStep: target field %r of type %s holds no tensor this class knows how to refill"
                            % (k, type(v).__name__))
    return out


def _tensors_of(images, targets):
    out = [images.tensors]
    for t in targets:
        out.extend(_target_tensors(t))
    return out


def _signature(images, targets):
    sig = [tuple(images.tensors.shape), tuple(tuple(s) for s in images.image_sizes)]
    for t in targets:
        sig.append((tuple(t.size), t.mode, tuple(sorted(t.extra_fields)), tuple((tuple(x.shape), x.dtype) for x in _target_tensors(t))))
    return tuple(sig)


def _clone_batch(images, targets):
    """a private copy of the batch: the static inputs of one graph"""
    import copy
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.image_list import ImageList
    im = ImageList(images.tensors.clone(), list(images.image_sizes))
    tg = []
    for t in targets:
        c = BoxList(t.bbox.clone(), t.size, t.mode)
        for k, v in t.extra_fields.items():
            if torch.is_tensor(v):
                v = v.clone()
            elif hasattr(v, "instances") and hasattr(v.instances, "masks"):
                v2 = copy.copy(v)
                v2.instances = copy.copy(v.instances)
                v2.instances.masks = v.instances.masks.clone()
                v = v2
            elif hasattr(v, "masks") and torch.is_tensor(v.masks):
                v2 = copy.copy(v)
                v2.masks = v.masks.clone()
                v = v2
            c.add_field(k, v)
        tg.append(c)
    return im, tg


class GraphedTrainStep(object):
    """`step` (an engine.ddp_step.TrainStep on a single-process model) captured per input signature and replayed.

    warmup: eager iterations run on a side stream before the first capture (allocator, MIOpen workspaces, lazily built
    caches), as torch.cuda.graph's documentation asks; max_graphs: signatures beyond that many run eagerly."""

    def __init__(self, step, warmup=3, max_graphs=4):
        if hasattr(step.model, "buckets") or type(step.model).__name__ == "BucketedDataParallel":
            raise RuntimeError("GraphedTrainStep: the data-parallel wrapper is not capturable (RCCL calls and stream waits "
                               "are enqueued from autograd hooks); use it on a single-process model")
        os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # (must be set before the HIP runtime starts to take effect)
        # nothing can be read back inside a capture: the mask head takes its static number of slots per image
        from maskrcnn_benchmark.modeling.roi_heads.mask_head import mask_head as _mh
        if _mh.SLOT_MODE == "dynamic":
            _mh.SLOT_MODE = "fixed"
            log.info("GraphedTrainStep: mask head slots fixed (a captured step cannot read the positive counts back)")
        self.eager = step
        # at least one eager iteration before the first capture: the optimizer creates its momentum buffers on its first step
        # (host-side `is None` test + zeros_like) — captured, that zero-fill would be replayed, i.e. reset the momentum every iteration
        self.warmup = max(1, int(warmup))
        self.max_graphs = int(max_graphs)
        self._graphs = collections.OrderedDict()
        self._pool = None
        self._warm = False
        dev = next(step.model.parameters()).device
        self.device = dev
        opt = step.optimizer
        if getattr(opt, "lr_tensors", "absent") == "absent":
            raise RuntimeError("GraphedTrainStep needs an optimizer with device-side learning rates (OverlappedSGD)")
        opt.lr_tensors = {gi: torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev) for gi, g in enumerate(opt.param_groups)}
        self._lr_host = {gi: None for gi in opt.lr_tensors}
        self.seed = torch.zeros((1,), dtype=torch.int64, device=dev)
        self.replays = 0
        self.eager_steps = 0

    # ------------------------------------------------------------------------------------------------------------------
    def _push_lr(self):
        opt = self.eager.optimizer
        for gi, g in enumerate(opt.param_groups):
            v = float(g["lr"])
            if self._lr_host[gi] != v:              # (the scheduler changes it every iteration during warm-up, rarely after)
                opt.lr_tensors[gi].fill_(v)
                self._lr_host[gi] = v

    def _iteration(self, images, targets):
        """what gets captured: TrainStep's body without its host-side tail (the scheduler)"""
        st = self.eager
        self.seed.add_(1)                            # first node of the graph: a new sampler stream per replay
        if st.amp_dtype is None:
            loss_dict = st.model(images, targets)
        else:
            with torch.autocast(device_type=st.device_type, dtype=st.amp_dtype):
                loss_dict = st.model(images, targets)
        losses = sum(loss for loss in loss_dict.values())
        st.optimizer.zero_grad(set_to_none=True)
        if st.scaler is not None:
            st.scaler.scale(losses).backward()
            st.scaler.step(st.optimizer)
            st.scaler.update()
        else:
            losses.backward()
            st.optimizer.step()
        return loss_dict

    def run_eager(self, images, targets):
        """one iteration WITHOUT a graph, through the same device-side learning rate / seed plumbing"""
        return self._run_eager(images, targets)

    def _run_eager(self, images, targets):
        self._push_lr()
        _C.GRAPH_SEED = self.seed
        try:
            out = self._iteration(images, targets)
        finally:
            _C.GRAPH_SEED = None
        if self.eager.scheduler is not None:
            self.eager.scheduler.step()
        self.eager_steps += 1
        return out

    def _capture(self, images, targets, sig):
        if not self._warm:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self._run_eager(images, targets)
            torch.cuda.current_stream(self.device).wait_stream(side)
            self._warm = True
        s_images, s_targets = _clone_batch(images, targets)
        graph = torch.cuda.CUDAGraph()
        self._push_lr()
        _C.GRAPH_SEED = self.seed
        try:
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(graph, pool=self._pool):
                losses = self._iteration(s_images, s_targets)
        finally:
            _C.GRAPH_SEED = None
        if self._pool is None:
            self._pool = graph.pool()
        entry = (graph, _tensors_of(s_images, s_targets), losses)
        self._graphs[sig] = entry
        log.info("captured a training step into a HIP graph (signature %d of at most %d)", len(self._graphs), self.max_graphs)
        return entry

    def __call__(self, images, targets):
        sig = _signature(images, targets)
        entry = self._graphs.get(sig)
        if entry is None:
            if len(self._graphs) >= self.max_graphs:
                return self._run_eager(images, targets)
            entry = self._capture(images, targets, sig)
        graph, statics, losses = entry
        for dst, src in zip(statics, _tensors_of(images, targets)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self._push_lr()
        graph.replay()
        if self.eager.scheduler is not None:
            self.eager.scheduler.step()
        self.replays += 1
        return losses
