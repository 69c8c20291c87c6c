"""RCCL called directly on a HIP stream of OUR choosing (reference tools/train_net.py:45-54 hands the gradients to
torch's DistributedDataParallel, whose ProcessGroupNCCL owns its streams, work objects, futures and watchdog).

`BucketedDataParallel` (engine/ddp_step.py) needs exactly one thing from the communication library per bucket: "average
this flat fp32 buffer over the ranks, in stream order, on the side stream the update runs on".  ProcessGroupNCCL wraps
that in an internal stream per device, an event pair and a work object per collective, a future whose callback runs on
yet another pool stream, and a watchdog thread polling the events — at world size 1, where RCCL itself does nothing,
that machinery alone cost 1.5-1.8 ms of a 38 ms step (profiles/r03n_data_parallel_overhead.txt, r05b_*).  Here the
communicator is created once (`ncclCommInitRank`, the unique id travels through the already initialised
torch.distributed group) and `ncclAllReduce(..., ncclAvg, comm, stream)` is enqueued on the ONE side stream that also
runs the bucket's SGD update: two HIP streams in the whole step, one event per bucket, no host thread besides the
launcher.

Everything here is plumbing around librccl.so (the copy torch already loaded: two RCCL instances in one process would
each bring their own topology discovery and IPC handles).  `selftest()` all-reduces a small buffer once at set-up and
checks the sum — the caller falls back to ProcessGroupNCCL when anything fails, so a multi-GPU run never depends on an
untested path.
"""
import ctypes
import os

import torch
import torch.distributed as dist

_NCCL_FLOAT32, _NCCL_SUM, _NCCL_AVG = 7, 0, 4


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def _load():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = ctypes.CDLL(path if os.path.exists(path) else "librccl.so")
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy"):
        getattr(lib, name).restype = ctypes.c_int
    return lib


def available():
    """Can this process call RCCL directly?  (library present, the entry points resolve.)  Checked on every rank BEFORE the
    collective `ncclCommInitRank`, so that a rank that cannot must not leave the others waiting inside it."""
    try:
        lib = _load()       # raises OSError (no library) or AttributeError (an entry point does not resolve)
        return all(hasattr(lib, n) for n in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy"))
    except Exception:       # noqa: BLE001 — a probe: ANY local failure means "this rank cannot", never an escape past the ranks' agreement
        return False


def low_priority_stream(device):
    """A HIP stream of the LOWEST priority the device offers (torch.cuda.Stream(priority=) only reaches 'normal' and
    above): the command processor serves the main stream's dispatches first, so the side stream's all-reduce and update
    kernels fill what the backward pass leaves free instead of halving the CUs of whatever runs beside them
    (ROIAlign backward took 180 instead of 98 us next to a normal-priority update, profiles/r04z_bench_forceddp.json).
    Returns (torch.cuda.ExternalStream, priority)."""
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    least, greatest = ctypes.c_int(0), ctypes.c_int(0)
    rc = hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest))
    if rc != 0:
        raise RuntimeError("hipDeviceGetStreamPriorityRange failed: %d" % rc)
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(handle), ctypes.c_uint(1), ctypes.c_int(least.value))   # 1 = hipStreamNonBlocking
    if rc != 0 or not handle.value:
        raise RuntimeError("hipStreamCreateWithPriority failed: %d" % rc)
    return torch.cuda.ExternalStream(handle.value, device=device), least.value


class RcclComm(object):
    """One RCCL communicator over the ranks of `process_group` (default WORLD), used from ONE thread."""

    def __init__(self, device, process_group=None):
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.group = process_group if process_group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        self.lib = _load()
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        if self.world > 1:
            # bytes(uid): the whole 128-byte struct (a c_char array FIELD reads as a C string and stops at the first NUL)
            box = [bytes(uid) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0), group=self.group)
            if not isinstance(box[0], (bytes, bytearray)) or len(box[0]) != ctypes.sizeof(_UniqueId):
                raise RuntimeError("RCCL unique id: expected %d bytes from rank 0" % ctypes.sizeof(_UniqueId))
            ctypes.memmove(ctypes.byref(uid), box[0], ctypes.sizeof(_UniqueId))
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def all_reduce_avg_(self, flat, stream):
        """flat (contiguous fp32 on this device) <- mean over the ranks, enqueued on `stream` (a torch.cuda stream)"""
        if flat.dtype != torch.float32 or not flat.is_contiguous() or flat.device != self.device:
            raise RuntimeError("all_reduce_avg_: a contiguous fp32 tensor on %s is required" % (self.device,))
        self._check(self.lib.ncclAllReduce(flat.data_ptr(), flat.data_ptr(), flat.numel(), _NCCL_FLOAT32, _NCCL_AVG,
                                           self.comm, stream.cuda_stream), "ncclAllReduce")

    def selftest(self):
        """one blocking all-reduce: rank r contributes r + 1 everywhere; the mean must be (world + 1) / 2"""
        stream = torch.cuda.current_stream(self.device)
        t = torch.full((4096,), float(self.rank + 1), dtype=torch.float32, device=self.device)
        self.all_reduce_avg_(t, stream)
        stream.synchronize()
        want = (self.world + 1) / 2.0
        if not bool(((t - want).abs() <= 1e-6 * want).all()):
            raise RuntimeError("RCCL self-test: got %r, expected %r" % (float(t[0]), want))

    def destroy(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()
