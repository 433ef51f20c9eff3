"""Gene-axis sharding across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI) as the exchange plumbing.

Column subproblems are independent given the row factors (src/optimize.cpp:215-230), so rank g owns a contiguous
gene slab of X / masks / C and runs the column pass on it alone.  The row update needs gene-global sums; every
quantity entering a level's normal equations is linear in per-slab partial sums, so each rank reduces its slab to
the L_i x (KP^2 + KP) per-level equations and ONE sum-all-reduce per covariate makes them global; every rank then
solves the same tiny systems redundantly, which keeps the row factors bit-identical everywhere.  The loss needs one
more all-reduce of 6 doubles per checkpoint (the collective BASELINE.json's north star names).
"""
import ctypes as C

import numpy as np


def shard_range(p, rank, world):
    """Contiguous gene slab [lo, hi) of rank `rank`: sizes differ by at most one gene."""
    base, rem = divmod(int(p), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class HostAllreduce:
    """All-reduce of a HOST buffer given by address (CPU / gloo rehearsal of the exchange protocol)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.calls = []

    def __call__(self, ptr, count, stream=0):
        import torch
        arr = np.ctypeslib.as_array((C.c_double * count).from_address(ptr))
        t = torch.from_numpy(arr)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self.calls.append(count)


class _DevBuf:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2,
                                         "strides": None}


class DeviceAllreduce:
    """All-reduce of a DEVICE buffer given by address, through torch.distributed (RCCL), ordered against the HIP stream
    the library passes in: the stream is wrapped in a torch.cuda.ExternalStream and made current for the call, so
    ProcessGroupNCCL makes its collective wait for the stream's prior kernels and the stream wait for the collective.
    No host synchronisation: the host keeps enqueueing the next kernels while the GPU works.

    Zero-copy when torch accepts the pointer through __cuda_array_interface__ (checked once); otherwise staged through
    a torch-owned buffer with synchronous device-to-device copies.
    """

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.device = torch.device("cuda", device)
        self.zero_copy = None
        self.stage = None
        self.hip = None
        self.ext = {}
        self.calls = []

    def _alias(self, ptr, count):
        return self.torch.as_tensor(_DevBuf(ptr, count), device=self.device)

    def _probe(self, ptr, count):
        torch = self.torch
        try:
            t = self._alias(ptr, count)
            ok = t.data_ptr() == ptr and t.dtype == torch.float64 and t.numel() == count
        except Exception:
            ok = False
        self.zero_copy = bool(ok)
        if not ok:
            self.hip = C.CDLL("libamdhip64.so")
            self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def _stream(self, stream):
        if stream not in self.ext:
            self.ext[stream] = self.torch.cuda.ExternalStream(stream, device=self.device)
        return self.ext[stream]

    def __call__(self, ptr, count, stream=0):
        torch = self.torch
        if self.zero_copy is None:
            self._probe(ptr, count)
        if self.zero_copy and stream:
            with torch.cuda.stream(self._stream(stream)):
                self.dist.all_reduce(self._alias(ptr, count), op=self.dist.ReduceOp.SUM, group=self.group)
        elif self.zero_copy:
            torch.cuda.synchronize(self.device)
            self.dist.all_reduce(self._alias(ptr, count), op=self.dist.ReduceOp.SUM, group=self.group)
            torch.cuda.synchronize(self.device)
        else:
            if stream:
                self.hip.hipStreamSynchronize(stream)
            if self.stage is None or self.stage.numel() < count:
                self.stage = torch.empty(max(count, 1 << 16), dtype=torch.float64, device=self.device)
            if self.hip.hipMemcpy(self.stage.data_ptr(), ptr, count * 8, 3) != 0:
                raise RuntimeError("hipMemcpy D2D failed")
            self.dist.all_reduce(self.stage[:count], op=self.dist.ReduceOp.SUM, group=self.group)
            torch.cuda.synchronize(self.device)
            if self.hip.hipMemcpy(ptr, self.stage.data_ptr(), count * 8, 3) != 0:
                raise RuntimeError("hipMemcpy D2D failed")
        self.calls.append(count)


class StagedHostAllreduce:
    """All-reduce of a DEVICE buffer through the HOST: wait for the library's stream, copy the buffer to pinned-size host
    memory, all-reduce it with a CPU backend (gloo), copy it back.  Slow by construction; it exists so that the sharded
    HIP path (per-level equations and loss terms crossing ranks between kernels) can be run with several processes
    on ONE GPU, where RCCL refuses duplicate devices (tests/test_gpu_sharded.py)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.calls = []

    def __call__(self, ptr, count, stream=0):
        import torch
        if self.hip.hipStreamSynchronize(stream) != 0:
            raise RuntimeError("hipStreamSynchronize failed")
        host = np.empty(count, dtype=np.float64)
        if self.hip.hipMemcpy(host.ctypes.data, ptr, count * 8, 2) != 0:      # hipMemcpyDeviceToHost
            raise RuntimeError("hipMemcpy D2H failed")
        self.dist.all_reduce(torch.from_numpy(host), op=self.dist.ReduceOp.SUM, group=self.group)
        if self.hip.hipMemcpy(ptr, host.ctypes.data, count * 8, 1) != 0:      # hipMemcpyHostToDevice
            raise RuntimeError("hipMemcpy H2D failed")
        self.calls.append(count)


def broadcast_comm_id(rank, group=None):
    """The RCCL unique id of the in-library communicator: made by rank 0 (insider_hip_comm_unique_id), handed to every
    rank over torch.distributed (any backend).  Without a process group (world 1) rank 0's own id is returned."""
    from . import _lib
    buf = (C.c_char * _lib.COMM_ID_BYTES)()
    if rank == 0:
        _lib.check(_lib.load().insider_hip_comm_unique_id(buf, _lib.COMM_ID_BYTES))
    box = [bytes(buf)]
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast_object_list(box, src=0, group=group)
    except ImportError:
        pass
    return box[0]


def attach(ds, gene_offset, rank, world, device=None, group=None, force=False, staged=False, mode=None):
    """Mark an InsiderData handle as one gene slab of a `world`-rank job and install the cross-rank all-reduce.

    ``mode``: "rccl" (default) = the library's own RCCL communicator, ncclAllReduce enqueued on the library's stream
    (insider_hip_comm_init; the unique id travels over torch.distributed once); "torch" = callback into
    torch.distributed on the library's stream (DeviceAllreduce); "staged" (or ``staged=True``) = through the host with
    the process group's CPU backend (several ranks on one GPU, where RCCL refuses duplicate devices).
    ``force``: install (and call) the all-reduce even for world == 1 (plumbing rehearsal on a single GPU)."""
    if world <= 1 and not force:
        ds.set_shard(gene_offset, 0, 1, None)
        return None
    mode = "staged" if staged else (mode or "rccl")
    if mode == "rccl":
        ds.set_shard(gene_offset, rank, world, None)
        ds.comm_init(broadcast_comm_id(rank, group), rank, world)
        ar = "rccl"
    else:
        ar = StagedHostAllreduce(group) if mode == "staged" else DeviceAllreduce(device if device is not None else 0, group)
        ds.set_shard(gene_offset, rank, world, ar)
    if force:
        ds.set_option("force_allreduce", 1)
    return ar


def _make_unique_id():
    from . import _lib
    buf = (C.c_char * _lib.COMM_ID_BYTES)()
    _lib.check(_lib.load().insider_hip_comm_unique_id(buf, _lib.COMM_ID_BYTES))
    return bytes(buf)


def _vote_min(flag, group=None, cuda=True):
    """MIN over the ranks of an integer flag (torch.distributed; the tensor lives where the backend wants it)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(flag)], dtype=torch.int32, device="cuda" if cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def attach_voted(ds, gene_offset, rank, world, device=0, group=None, fallback="torch", log=None):
    """`attach(mode="rccl")` for a job whose ranks must END UP ON THE SAME EXCHANGE whatever happens to one of them: the
    in-library RCCL communicator when EVERY rank can join it, else the `fallback` mode ("torch": torch.distributed callback on
    the library's stream; "staged": through the host) on every rank together.  ncclCommInitRank is a blocking collective: a
    rank that failed before entering it would leave the others waiting for ever, so there are two votes (sum-MIN all-reduces
    over torch.distributed): (1) after everything a rank does alone — shard set-up, the unique id (rank 0; its bytes travel
    in a broadcast every rank takes part in, an empty id = rank 0 failed) — "ready to join?"; only when all are does anyone
    call insider_hip_comm_init(); (2) after it, "joined?": a communicator that came up on some ranks only is dropped
    everywhere (installing the callback replaces it, include/insider_hip.h).
    INSIDER_FAIL_COMM_RANK=<r> makes rank r fail in phase 1 (rehearsal of the first vote without a broken node).
    Returns (exchange, report): exchange = "rccl" or the callback object; report = the votes, for the bench line."""
    import os
    from . import _lib
    cuda = True
    try:
        import torch.distributed as dist
        cuda = dist.get_backend(group) == "nccl"
    except Exception:
        pass
    report = {"attempted": "rccl", "ready_min": None, "joined_min": None, "path": None, "errors": []}
    say = log or (lambda msg: None)
    # ---- phase 1: what a rank does alone -----------------------------------------------------------------------------------
    ready, uid = 1, b""
    try:
        if os.environ.get("INSIDER_FAIL_COMM_RANK") == str(rank):
            raise RuntimeError(f"injected failure on rank {rank} (INSIDER_FAIL_COMM_RANK)")
        ds.set_shard(gene_offset, rank, world, None)
    except Exception as e:
        ready = 0
        report["errors"].append(f"prepare: {e!r}")
        say(f"in-library RCCL: rank {rank} cannot prepare ({e!r})")
    try:      # every rank takes part in the broadcast, ready or not; rank 0's failure travels as an empty id
        if rank == 0:
            try:
                uid = _make_unique_id()
            except Exception as e:
                ready = 0
                report["errors"].append(f"unique id: {e!r}")
        box = [uid]
        import torch.distributed as dist
        dist.broadcast_object_list(box, src=0, group=group)
        uid = box[0]
    except Exception as e:
        ready = 0
        report["errors"].append(f"id broadcast: {e!r}")
    if len(uid) != _lib.COMM_ID_BYTES:
        ready = 0
    report["ready_min"] = _vote_min(ready, group, cuda)
    joined = 0
    if report["ready_min"] == 1:
        # ---- phase 2: the collective join ----------------------------------------------------------------------------------
        try:
            ds.comm_init(uid, rank, world)
            joined = 1
        except Exception as e:
            report["errors"].append(f"comm_init: {e!r}")
            say(f"in-library RCCL: rank {rank} could not join ({e!r})")
        report["joined_min"] = _vote_min(joined, group, cuda)
    if report["ready_min"] == 1 and report["joined_min"] == 1:
        report["path"] = "rccl"
        return "rccl", report
    say(f"rank {rank}: all ranks fall back to the {fallback} exchange together (ready {report['ready_min']}, joined {report['joined_min']})")
    ar = StagedHostAllreduce(group) if fallback == "staged" else DeviceAllreduce(device, group)
    ds.set_shard(gene_offset, rank, world, ar)       # replaces a communicator that came up on this rank only
    report["path"] = fallback
    return ar, report
