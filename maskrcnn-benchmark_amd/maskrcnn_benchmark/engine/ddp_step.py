"""The data-parallel training step (reference tools/train_net.py:45-54 + engine/trainer.py:66-99:
DistributedDataParallel wrap, `losses.backward()`, `optimizer.step()`).

MI355X design.  One process per GPU; gradients are all-reduced by RCCL over xGMI in ~25 MB fp32
buckets while the backward pass is still running, and each bucket's SGD update is chained to that
bucket's all-reduce as a future callback: it executes on the communication side stream as soon as the
bucket's averaged gradient exists, overlapping the remaining backward compute; the main stream joins
the side stream when the backward pass ends.  With one process there is no communication and the same
multi-tensor update runs on the main stream after backward.

The wrapper is `BucketedDataParallel` below, not torch's DistributedDataParallel: this model needs none
of DDP's generality (every parameter receives a gradient in every step, no buffer is trained — every
BatchNorm is frozen —, one backward per step), and DDP's per-step host work (the reducer's bucket
bookkeeping, the forward-time graph walk, gradients accumulated INTO pre-assigned bucket views: one
add kernel per parameter) costs 2.0-2.5 ms of a 39 ms step at ANY bucket size (profiles/r03m_*).
Here a step costs one Python hook per parameter (a counter), one multi-tensor copy + one collective +
one fused update per bucket.

The update is SGD with momentum exactly as torch.optim.SGD computes it (dampening 0, no Nesterov):
    g <- g + wd * p ;  buf <- momentum * buf + g (buf <- g on the first step) ;  p <- p - lr * buf
with the reference's per-parameter hyper-parameters (biases: lr x BIAS_LR_FACTOR, WEIGHT_DECAY_BIAS;
solver/build.py:7-20), held in the optimizer's param groups so LR schedulers keep working.
"""
import logging
import os

import torch
import torch.distributed as dist

from maskrcnn_benchmark import _C

_FUSED_SGD = hasattr(torch, "_fused_sgd_") and os.environ.get("DETOPS_FUSED_SGD", "1") != "0"


class OverlappedSGD(torch.optim.Optimizer):
    """SGD(momentum) whose update can be applied to an arbitrary subset of its parameters
    (`step_params`) — the unit the DDP bucket callback works on — using multi-tensor kernels."""

    def __init__(self, params, lr, momentum=0.0):
        super(OverlappedSGD, self).__init__(params, dict(lr=lr, momentum=momentum, weight_decay=0.0))
        self._group_of = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                self._group_of[p] = gi
        self.deferred = False  # True while BucketedDataParallel applies the updates during backward
        # group index -> fp32 [1] device tensor holding that group's learning rate: set by engine/graph_step.py, whose
        # captured step must not freeze the scheduler's value of the capture step into the kernel arguments
        self.lr_tensors = None
        # torch.amp.GradScaler.step() then hands `grad_scale` / `found_inf` over as attributes and does NOT read
        # found_inf back to the host: the fused kernel unscales and skips on the device (no sync in the fp16 step)
        # (only the fused path takes grad_scale / found_inf: momentum != 0 in every group, fp32 parameters on the GPU)
        self._step_supports_amp_scaling = _FUSED_SGD and all(g["momentum"] != 0 for g in self.param_groups) and all(
            p.is_cuda and p.dtype == torch.float32 for g in self.param_groups for p in g["params"])

    @torch.no_grad()
    def step_params(self, params, grads=None):
        """Update `params`; `grads` (same order) overrides `p.grad` — the bucket callback passes the
        bucket's gradient views, which hold the averaged values whatever `p.grad` points at."""
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        by_group = {}
        for i, p in enumerate(params):
            gr = p.grad if grads is None else grads[i]
            if gr is None:
                continue
            entry = by_group.setdefault(self._group_of[p], ([], []))
            entry[0].append(p)
            entry[1].append(gr)
        for gi, (ps, grads_g) in by_group.items():
            g = self.param_groups[gi]
            lr, mom, wd = g["lr"], g["momentum"], g["weight_decay"]
            grads = list(grads_g)
            fused = mom != 0 and _FUSED_SGD and all(p.is_cuda and p.dtype == torch.float32 for p in ps)
            if (grad_scale is not None or found_inf is not None) and not fused:
                raise RuntimeError("loss scaling is only wired into the fused device update (fp32 parameters on the GPU)")
            if fused:
                # one multi-tensor pass (read g, p, buf; write buf, p) instead of four (weight decay, momentum
                # scale, momentum add, parameter add): 0.46 -> 0.2 ms for the 176 MB of R-50-FPN parameters
                # Momentum buffers start as ZEROS and every call is a "later" step: momentum * 0 + g = g is bit for bit
                # the first step of torch.optim.SGD (buf = g), and a step the loss scaler skips (found_inf: the kernel
                # returns without touching anything) leaves a valid buffer behind — an uninitialised buffer marked as
                # existing would feed garbage into the first step that is NOT skipped.
                bufs = []
                for p, gr in zip(ps, grads):
                    st = self.state[p]
                    if st.get("momentum_buffer") is None:   # absent, or None in a torch.optim.SGD checkpoint saved before its first step
                        st["momentum_buffer"] = torch.zeros_like(gr, dtype=p.dtype)
                    bufs.append(st["momentum_buffer"])
                if self.lr_tensors is not None:
                    lr = self.lr_tensors[gi]
                torch._fused_sgd_(ps, grads, bufs, weight_decay=wd, momentum=mom, lr=lr, dampening=0.0, nesterov=False,
                                  maximize=False, is_first_step=False, grad_scale=grad_scale, found_inf=found_inf)
                continue
            if wd != 0:
                grads = torch._foreach_add(grads, ps, alpha=wd)
            if mom != 0:
                bufs, fresh = [], []
                for p, gr in zip(ps, grads):
                    st = self.state[p]
                    if st.get("momentum_buffer") is None:
                        st["momentum_buffer"] = torch.clone(gr).detach()
                        fresh.append(True)
                    else:
                        fresh.append(False)
                    bufs.append(st["momentum_buffer"])
                old = [b for b, f in zip(bufs, fresh) if not f]
                if old:
                    torch._foreach_mul_(old, mom)
                    torch._foreach_add_(old, [gr for gr, f in zip(grads, fresh) if not f])
                grads = bufs
            torch._foreach_add_(ps, grads, alpha=-lr)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if not self.deferred:
            self.step_params([p for g in self.param_groups for p in g["params"]])
        return loss


def make_overlapped_sgd(cfg, model):
    """reference solver/build.py:7-20 hyper-parameters, two param groups (weights / biases)."""
    S = cfg.SOLVER
    weights, biases = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (biases if "bias" in name else weights).append(p)
    groups = []
    if weights:
        groups.append({"params": weights, "lr": S.BASE_LR, "weight_decay": S.WEIGHT_DECAY})
    if biases:
        groups.append({"params": biases, "lr": S.BASE_LR * S.BIAS_LR_FACTOR, "weight_decay": S.WEIGHT_DECAY_BIAS})
    return OverlappedSGD(groups, S.BASE_LR, momentum=S.MOMENTUM)


class _Bucket(object):
    """One gradient bucket: `flat` (the all-reduced gradients) with a view per parameter.  Layout: the optimizer's first
    group (weights) first, then the others (biases), every slot starting on a 64-float (256-byte) boundary — `split` is
    where the biases begin; padding belongs to nobody and stays zero.  `flat_p` / `flat_m` (native update path only):
    the parameters themselves and their momentum in the SAME layout, so that the bucket's SGD update is one streaming
    pass over three flat arrays (csrc/optim.hip)."""
    __slots__ = ("params", "flat", "views", "pending", "event", "offsets", "split", "flat_p", "flat_m")

    ALIGN = 64

    def __init__(self, params, is_bias=None):
        if is_bias is not None:
            params = [p for p in params if not is_bias(p)] + [p for p in params if is_bias(p)]
        self.params = params
        self.offsets, at, self.split = [], 0, None
        for p in params:
            if is_bias is not None and self.split is None and is_bias(p):
                self.split = at
            self.offsets.append(at)
            at += -(-p.numel() // self.ALIGN) * self.ALIGN
        if self.split is None:
            self.split = at
        self.flat = torch.zeros(at, dtype=params[0].dtype, device=params[0].device)
        # a view has its PARAMETER's strides (a channels-last convolution weight stays channels-last inside its slot): the
        # fused multi-tensor SGD of the process-group path needs parameter, gradient and momentum laid out alike
        self.views = [self._view(self.flat, o, p) for o, p in zip(self.offsets, params)]
        self.pending = len(params)
        self.event = None       # direct-RCCL path: "this bucket is packed" (recorded on the main stream, reused every step)
        self.flat_p = self.flat_m = None

    @staticmethod
    def _view(flat, offset, p):
        piece = flat[offset:offset + p.numel()]
        if p.is_contiguous() or not p.is_contiguous(memory_format=torch.channels_last):
            return piece.view(p.shape)
        return piece.as_strided(p.shape, p.stride())

    def contiguous_views(self):
        """the native flat-bucket path moves the parameters themselves into row-major slots: the gradient views follow"""
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, self.params)]


class BucketedDataParallel(torch.nn.Module):
    """Data parallelism for a model whose every parameter gets a gradient in every step.

    Parameters are dealt into flat buckets of `bucket_cap_mb` in REVERSE registration order (about the order the
    backward pass finishes them).  A post-accumulate hook per parameter counts its bucket down; a finished bucket
    (buckets go out strictly in index order, so every rank issues the same collective sequence) is packed with one
    multi-tensor copy, averaged over the ranks by one asynchronous all-reduce, and — `optimizer.deferred` — updated
    in the all-reduce's completion callback on the communication stream.  `p.grad` is re-pointed at the bucket view,
    i.e. after the step it holds the AVERAGED gradient like under torch DDP.  When the backward pass ends, buckets
    that did not fill (a parameter without gradient on this rank: its slice is sent as zeros) are flushed and the
    main stream joins the callbacks.  `module` is the wrapped model (checkpoints strip the prefix as for DDP).

    Communication (`comm`): "direct" — RCCL called through engine/rccl_comm.py on ONE side stream (HIGH priority by
    default: its own hardware queue; DETOPS_DDP_PRIO=low|normal|main for A/B runs) that also runs the bucket updates (two HIP streams in the step, one event per bucket; the default on the GPU: "auto" tries
    it, self-tests one all-reduce and falls back); "pg" — torch.distributed's ProcessGroupNCCL (its own stream, a work
    object + future per bucket, the update in the future's callback on a pool stream) and the only path for gloo / CPU
    tensors.  `self.comm_mode` says which one runs.

    Not supported, on purpose: gradient accumulation over several backward passes, trained buffers, parameters
    of a rank that change `requires_grad` after wrapping."""

    def __init__(self, module, optimizer=None, bucket_cap_mb=25, process_group=None, overlap_optimizer=True,
                 error_on_unused=False, comm=None):
        super(BucketedDataParallel, self).__init__()
        self.module = module
        # (mixed precision: the per-stage half-weight casts of layers/half_weights.py deliver a stage's fp32 weight gradients as
        #  soon as the stage's backward is over — the bucket hooks below see them at the same points as autocast's per-layer casts)
        self.process_group = process_group if process_group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.process_group)
        self.optimizer = optimizer if isinstance(optimizer, OverlappedSGD) else None
        # A parameter without a gradient on THIS rank may have one on another: its slice is sent as zeros and the averaged
        # gradient is then treated like any other — weight decay and momentum apply (torch.optim.SGD skips a parameter
        # whose grad is None, which a rank cannot know about the other ranks without a read-back).  error_on_unused=True
        # raises instead, like torch DDP without find_unused_parameters.
        self.error_on_unused = bool(error_on_unused)
        self.overlap_optimizer = (bool(overlap_optimizer) and self.optimizer is not None
                                  and os.environ.get("DETOPS_DDP_OVERLAP", "1") != "0")
        if self.optimizer is not None:
            self.optimizer.deferred = self.overlap_optimizer
        # RCCL averages inside the collective; gloo (CPU tests) has no AVG: pre-divide there
        self._avg = dist.ReduceOp.AVG if dist.get_backend(self.process_group) == "nccl" else None
        self._sync_module_states()
        cap = max(int(bucket_cap_mb * 1024 * 1024), 1)
        self.buckets, cur, cur_bytes = [], [], 0
        group_of = self.optimizer._group_of if self.optimizer is not None else {}
        is_bias = (lambda q: group_of.get(q, 0) != 0) if self.optimizer is not None else None
        for p in reversed([p for p in module.parameters() if p.requires_grad]):
            if cur and (cur_bytes + p.numel() * p.element_size() > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(_Bucket(cur, is_bias))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
        if cur:
            self.buckets.append(_Bucket(cur, is_bias))
        self._next, self._armed, self._futures = 0, False, []
        self.exposed_wait_events = None     # a list: _finish_backward appends an event pair around the main stream's join
        spec = os.environ.get("DETOPS_DDP_STANDIN", "")
        self._standin = tuple(int(v) for v in spec.split(":")) if spec else None
        self._setup_comm(comm if comm is not None else os.environ.get("DETOPS_DDP_COMM", "auto"))
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(b))
                       for b in self.buckets for p in b.params]

    def _setup_comm(self, want):
        """want: auto | direct | pg | side-nocoll (measurement only: the direct path without the collective)"""
        self.comm_mode, self.comm_note, self._rccl, self._side = "pg", None, None, None
        self._native_update = False
        dev = self.buckets[0].flat.device if self.buckets else None
        # (`_C.on_device`, not `.is_cuda`: the CPU suite drives this path too, over the emulated kernels and a stand-in for
        #  librccl.so on gloo — tests/test_ddp_cpu.py)
        on_gpu = dev is not None and _C.on_device(self.buckets[0].flat) and all(
            b.flat.device == dev and b.flat.dtype == torch.float32 for b in self.buckets)
        if want == "pg" or not on_gpu or dist.get_backend(self.process_group) != "nccl":
            if want in ("direct", "side-nocoll"):
                raise RuntimeError("comm=%r needs fp32 parameters on one GPU and an RCCL process group" % want)
            return
        # Phase 1 — everything that can fail on ONE rank alone (library load, symbol resolution, side-stream creation) runs
        # BEFORE any collective and is folded into one flag; the ranks then agree (MIN) on whether all of them can go on.
        # A rank that cannot never skips a collective its peers are waiting in.
        local_err, side, prio, rccl_comm = None, None, 0, None
        try:
            from . import rccl_comm
            if not rccl_comm.available():
                raise RuntimeError("librccl.so (or one of its entry points) is not loadable")
            # "high" by default (round 6, profiles/r06_ddp_contention.txt): with GPU_MAX_HW_QUEUES=2 a NORMAL-priority stream
            # shares the compute stream's hardware queue — whatever runs on it serialises with the step (measured with a
            # contention stand-in: +8 x its duration); an explicit priority gets its own queue and overlaps
            prio_want = os.environ.get("DETOPS_DDP_PRIO", "high")
            if prio_want == "low":
                side, prio = rccl_comm.low_priority_stream(dev)
            elif prio_want == "high":
                side, prio = torch.cuda.Stream(dev, priority=-1), -1
            elif prio_want == "main":     # measurement only: no second stream at all
                side, prio = torch.cuda.current_stream(dev), 0
            else:                         # "normal": shares a hardware queue with the compute stream unless GPU_MAX_HW_QUEUES >= 8
                side, prio = torch.cuda.Stream(dev), 0
        except Exception as e:  # noqa: BLE001 — local set-up only; reported in comm_note
            local_err = e
        ok = torch.tensor([0.0 if local_err is not None else 1.0], device=dev)
        if self.world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.process_group)
        if float(ok) < 1.0:
            if want in ("direct", "side-nocoll"):
                raise RuntimeError("comm=%r: the direct RCCL path cannot be set up on every rank (%s)" % (
                    want, "this rank: %s: %s" % (type(local_err).__name__, local_err) if local_err is not None else "another rank failed"))
            self.comm_note = "direct RCCL unavailable (%s)" % (
                "%s: %s" % (type(local_err).__name__, local_err) if local_err is not None else "another rank could not set it up")
        else:
            # Phase 2 — the collective part: every rank enters it (agreed above).  RuntimeError is what RcclComm raises for a
            # failed RCCL call or self-test; anything else is a programming error and propagates.
            try:
                self._side = side
                if want != "side-nocoll":
                    self._rccl = rccl_comm.RcclComm(dev, self.process_group)
                    self._rccl.selftest()
                for b in self.buckets:
                    b.event = torch.cuda.Event()
                self.comm_mode = "direct" if want != "side-nocoll" else "side-nocoll"
                self.comm_note = "side stream priority %d" % prio
            except (RuntimeError, OSError) as e:
                if want == "direct" or want == "side-nocoll":
                    raise
                self._drop_direct()
                self.comm_note = "direct RCCL unavailable (%s: %s)" % (type(e).__name__, e)
        if self._side is not None and os.environ.get("DETOPS_DDP_NATIVE", "1") != "0":
            self._make_flat_parameters()
            if self._native_update:
                self.comm_note += ", flat buckets: native pack + SGD kernels"
        if self.world > 1:   # every rank must take the same path: the collectives of the two paths do not match up
            flag = torch.tensor([1.0 if self.comm_mode == "pg" else 0.0], device=dev)
            dist.all_reduce(flag, group=self.process_group)
            if 0 < float(flag) and self.comm_mode != "pg":
                self._drop_direct()
                self.comm_note = "another rank fell back to ProcessGroupNCCL"
        logging.getLogger("maskrcnn_benchmark.ddp").info(
            "BucketedDataParallel: %d bucket(s), world %d, comm_mode=%s (%s)", len(self.buckets), self.world, self.comm_mode, self.comm_note)

    def _drop_direct(self):
        """leave the direct path: destroy the communicator (if one was created), forget the side stream, fall back to the
        process group's collectives"""
        if self._rccl is not None:
            try:
                self._rccl.destroy()
            except Exception:  # noqa: BLE001
                pass
        if self._native_update:
            raise RuntimeError("BucketedDataParallel: cannot leave the direct path after the parameters moved into flat buckets")
        self.comm_mode, self._rccl, self._side = "pg", None, None

    def close(self):
        """Release what the wrapper owns outside torch's process group: wait for the side stream, destroy the wrapper's own
        RCCL communicator and drop the stream.  Call before `dist.destroy_process_group()` (tools/train_net.py, bench.py);
        idempotent.  The wrapped module stays usable without the wrapper (its parameters keep their values)."""
        dev = self.buckets[0].flat.device if self.buckets else None
        if dev is not None and dev.type == "cuda":
            torch.cuda.synchronize(dev)
        for h in getattr(self, "_hooks", []):
            h.remove()
        self._hooks = []
        if self._rccl is not None:
            self._rccl.destroy()
            self._rccl = None
        self._side = None

    @torch.no_grad()
    def _make_flat_parameters(self):
        """Native update path (direct mode, overlapped fp32 SGD with momentum, at most two hyper-parameter groups): every
        bucket's parameters and momentum buffers MOVE into flat arrays laid out like its gradient array — `p.data` and
        the optimizer's `momentum_buffer` become views of them (state_dict / load_state_dict / checkpoints are unaffected:
        they go through the views).  A parameter re-allocated afterwards (`model.to(...)`, `.data = ...`) would leave its
        bucket: `_launch` checks the addresses."""
        opt = self.optimizer
        if opt is None or not self.overlap_optimizer or len(opt.param_groups) > 2:
            return
        moms = set(g["momentum"] for g in opt.param_groups)
        if len(moms) != 1 or 0 in moms or any(g.get("nesterov") or g.get("dampening") for g in opt.param_groups):
            return
        owned = set(id(q) for g in opt.param_groups for q in g["params"])
        if any(id(q) not in owned for b in self.buckets for q in b.params):
            return      # a trained parameter the optimizer does not know: the torch path reports it (KeyError in step_params)
        for b in self.buckets:
            b.flat_p, b.flat_m = torch.zeros_like(b.flat), torch.zeros_like(b.flat)
            b.contiguous_views()
            for p, o in zip(b.params, b.offsets):
                pv = b.flat_p[o:o + p.numel()].view(p.shape)
                pv.copy_(p.data)
                p.data = pv
                mv = b.flat_m[o:o + p.numel()].view(p.shape)
                old = opt.state[p].get("momentum_buffer")
                if old is not None:
                    mv.copy_(old)
                opt.state[p]["momentum_buffer"] = mv
        self._native_update = True

    @torch.no_grad()
    def _adopt_momentum(self, b):
        """the optimizer's momentum buffers of this bucket -> views of its flat momentum array (values kept)"""
        for p, o in zip(b.params, b.offsets):
            mv = b.flat_m[o:o + p.numel()].view(p.shape)
            old = self.optimizer.state[p].get("momentum_buffer")
            if old is None:
                mv.zero_()
            elif old.data_ptr() != mv.data_ptr():
                mv.copy_(old)
            self.optimizer.state[p]["momentum_buffer"] = mv

    def forward(self, *args, **kwargs):
        if self._armed:             # a backward pass that raised never reached _finish_backward: start clean
            self._reset()
        return self.module(*args, **kwargs)

    def _reset(self):
        del self._futures[:]
        for b in self.buckets:
            b.pending = len(b.params)
        self._next, self._armed = 0, False

    def _sync_module_states(self):
        """rank 0's parameters and buffers everywhere (what DDP's constructor does), one broadcast per dtype"""
        tensors = [t.detach() for t in list(self.module.parameters()) + list(self.module.buffers())]
        by_kind = {}
        for t in tensors:
            by_kind.setdefault((t.dtype, t.device), []).append(t)
        for group in by_kind.values():
            flat = torch.cat([t.reshape(-1) for t in group])
            src = 0 if self.process_group is None else dist.get_global_rank(self.process_group, 0)   # broadcast takes a GLOBAL rank
            dist.broadcast(flat, src, group=self.process_group)
            at = 0
            with torch.no_grad():
                for t in group:
                    t.copy_(flat[at:at + t.numel()].view(t.shape))
                    at += t.numel()

    def _make_hook(self, bucket):
        def hook(param):
            if not self._armed:
                self._armed = True
                torch.autograd.Variable._execution_engine.queue_callback(self._finish_backward)
            bucket.pending -= 1
            if bucket.pending == 0:
                self._advance()
        return hook

    def _advance(self):
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    @torch.no_grad()
    def _launch(self, b):
        grads = [p.grad for p in b.params]
        missing = any(g is None for g in grads)
        if missing and self.error_on_unused:    # only on the flush path
            raise RuntimeError("BucketedDataParallel(error_on_unused=True): %d parameter(s) of a bucket received no "
                               "gradient in this backward pass" % sum(g is None for g in grads))
        if self._side is not None:              # one launch per 48 tensors, pointer table in the kernel arguments
            if missing:
                for v, g in zip(b.views, grads):
                    if g is None:
                        v.zero_()
            have = [(g, o) for g, o in zip(grads, b.offsets) if g is not None]
            _C.pack_into(b.flat, [g for g, _ in have], [o for _, o in have])
        elif missing:
            for v, g in zip(b.views, grads):
                v.copy_(g) if g is not None else v.zero_()
        else:
            torch._foreach_copy_(b.views, grads)
        for p, v in zip(b.params, b.views):
            p.grad = v
        if self._side is not None:
            # direct path: "packed" event on the main stream -> the side stream averages the bucket over the ranks and
            # updates its parameters, in stream order; nothing returns to the host
            side = self._side
            b.event.record(torch.cuda.current_stream(b.flat.device))
            side.wait_event(b.event)
            if self._rccl is not None:
                self._rccl.all_reduce_avg_(b.flat, side)
            if self._standin is not None:
                # measurement only (DETOPS_DDP_STANDIN="workgroups:microseconds"): hold CUs on the side stream the way an
                # N > 1 ring all-reduce kernel would for this bucket — the 1-rank collective above is a no-op copy
                _C.check(_C.lib.detops_debug_occupy(self._standin[0], self._standin[1], side.cuda_stream), "debug_occupy")
            if self.overlap_optimizer and self.optimizer.deferred:
                self.optimizer.last_update_stream = side.cuda_stream
                if self._native_update:
                    # every parameter (and momentum buffer) of the bucket must still BE its slice of the flat arrays: a
                    # partial `.data = ...` / load_state_dict(assign=True) / optimizer state load would otherwise leave a
                    # parameter that the kernel no longer updates (host loop over <= a few hundred pointers per bucket)
                    base_p, base_m, state = b.flat_p.data_ptr(), b.flat_m.data_ptr(), self.optimizer.state
                    adopt = False
                    for q, o in zip(b.params, b.offsets):
                        if q.data_ptr() != base_p + 4 * o:
                            raise RuntimeError("BucketedDataParallel: a parameter was re-allocated after wrapping (model.to(...), "
                                               ".data = ..., load_state_dict(assign=True)?): its storage is no longer the bucket's")
                        mb = state[q].get("momentum_buffer")
                        if mb is None or mb.data_ptr() != base_m + 4 * o:
                            adopt = True
                    if adopt:
                        self._adopt_momentum(b)     # optimizer.load_state_dict() replaced (some of) the state tensors
                    gs = self.optimizer.param_groups
                    gw, gb = gs[0], gs[-1]
                    _C.sgd_momentum_flat_(b.flat_p, b.flat, b.flat_m, b.split, gw["lr"], gw["weight_decay"], gb["lr"],
                                          gb["weight_decay"], gw["momentum"], stream=side)
                else:
                    with torch.cuda.stream(side):
                        self.optimizer.step_params(b.params, b.views)
            return
        if self._avg is None:
            if self.world > 1:
                b.flat.div_(self.world)
            work = dist.all_reduce(b.flat, group=self.process_group, async_op=True)
        else:
            work = dist.all_reduce(b.flat, op=self._avg, group=self.process_group, async_op=True)
        fut = work.get_future()
        if self.overlap_optimizer and self.optimizer.deferred:
            opt = self.optimizer

            def apply(f, b=b):
                if b.flat.is_cuda:  # which stream the update kernels are enqueued on (read by the tests)
                    opt.last_update_stream = torch.cuda.current_stream(b.flat.device).cuda_stream
                opt.step_params(b.params, b.views)
                return b.flat       # a tensor: the child future records this stream's event for wait()

            fut = fut.then(apply)
        self._futures.append(fut)

    def _finish_backward(self):
        for b in self.buckets[self._next:]:
            self._launch(b)
        for f in self._futures:
            f.wait()                # device: the current stream waits for the callbacks' stream; host does not block
        if self._side is not None and self.buckets:
            main = torch.cuda.current_stream(self.buckets[0].flat.device)
            if self.exposed_wait_events is not None:
                # diagnosis (bench.py's post-pass): how long the compute stream stands still here = the part of the bucket
                # all-reduces + updates the backward pass did NOT hide
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(main)
                main.wait_stream(self._side)
                b.record(main)
                self.exposed_wait_events.append((a, b))
            else:
                main.wait_stream(self._side)   # device-side join
        self._reset()


def wrap_data_parallel(model, optimizer=None, device_ids=None, bucket_cap_mb=25, process_group=None,
                       overlap_optimizer=True, force=False):
    """BucketedDataParallel over RCCL (or gloo in tests) with the optimizer overlapped into the
    gradient all-reduce.  Returns the wrapped model (the bare model for a single process, unless
    `force`: then a 1-rank process group runs the very same bucket -> all-reduce -> update-in-the-
    callback path, which is how the hook is exercised on a single MI355X).  `device_ids` is accepted
    for the reference's call shape (tools/train_net.py:48-53); the module already lives on its device."""
    if not (dist.is_available() and dist.is_initialized()):
        if force:
            raise RuntimeError("force=True needs an initialised process group (world size 1 is fine)")
        return model
    if dist.get_world_size() == 1 and not force:
        return model
    return BucketedDataParallel(model, optimizer, bucket_cap_mb=bucket_cap_mb, process_group=process_group,
                                overlap_optimizer=overlap_optimizer)


class TrainStep(object):
    """One training iteration: forward (optionally under autocast) -> sum of losses -> backward
    (+ overlapped all-reduce/optimizer) -> optimizer.  Returns the loss dict (device tensors; no
    host sync)."""

    def __init__(self, model, optimizer, scheduler=None, dtype="float32", device_type="cuda"):
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.amp_dtype = {"float32": None, "float16": torch.float16, "bfloat16": torch.bfloat16}[dtype]
        self.device_type = device_type
        self.scaler = None
        if self.amp_dtype is torch.float16:
            # the gradient all-reduce still overlaps the backward pass; the update waits for the whole step's
            # found_inf (a skipped step must skip EVERY parameter), then runs fused on the device, no host sync
            if getattr(optimizer, "deferred", False):
                optimizer.deferred = False
            self.scaler = torch.amp.GradScaler(device_type)

    def __call__(self, images, targets):
        if self.amp_dtype is None:
            loss_dict = self.model(images, targets)
        else:
            with torch.autocast(device_type=self.device_type, dtype=self.amp_dtype):
                loss_dict = self.model(images, targets)
        losses = sum(loss for loss in loss_dict.values())
        self.optimizer.zero_grad(set_to_none=True)   # gradients are stolen, not accumulated: no fill, no add kernels
        if self.scaler is not None:
            self.scaler.scale(losses).backward()
            self.scaler.step(self.optimizer)
            self.scaler.update()
        else:
            losses.backward()
            self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        return loss_dict
