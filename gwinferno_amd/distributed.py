"""Multi-GPU evaluation: one process per GPU, events and injections sharded, ONE collective.

The reference has no distributed path; the likelihood nevertheless shards
naturally: events are independent (pipeline/analysis.py:78-86 reduces along axis 1 only) and the
injection sum is associative (:126-134).  Rank r scans a contiguous block of events and a slice of
the injections and publishes a small partial record (8 + n_norms + 2 n_theta doubles: sums of
per-event log-sums / variances, min log n_eff, the injection (M, S1, S2) triple, gradient numerators).
The only exchange is an all-gather of those records -- RCCL over xGMI when the process group's
backend is "nccl", gloo in the CPU tests -- after which every rank assembles the identical result
with the same summation order (deterministic, rank-symmetric).  The payload is ~1 KiB, so the
collective is latency-bound; there is exactly one per evaluation.
"""
import numpy as np

from .engine import EvalResult


def rccl_library_path():
    """The librccl the running PyTorch uses (so the engine and torch.distributed share one copy)."""
    import os

    import torch

    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


def init_engine_communicator(engine, group=None):
    """Create the engine's own RCCL communicator: rank 0 draws the ncclUniqueId, torch.distributed
    (any backend) broadcasts its 128 bytes, every rank calls ncclCommInitRank inside the engine."""
    import ctypes as C

    import torch.distributed as dist

    from . import _native as N

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    path = rccl_library_path()
    box = [None]
    if rank == 0:
        buf = C.create_string_buffer(128)
        st = N.load_library().gwi_comm_unique_id(path.encode() if path else None, buf)
        box[0] = bytes(buf.raw) if st == 0 else f"gwi_comm_unique_id failed ({N.STATUS_NAMES.get(st, st)})"
    dist.broadcast_object_list(box, src=0, group=group)  # (a failure on rank 0 travels too: nobody is left waiting in the broadcast)
    if not isinstance(box[0], bytes):
        raise N.NativeEngineError(str(box[0]))
    engine.comm_init(box[0], rank, world, rccl_path=path)


def init_shared_memory_exchange(engine, group=None):
    """Single-node exchange without a collective launch (``gwi_shm_comm_init``): rank 0 names a POSIX shared-memory
    segment, torch.distributed (any backend) hands the name around, every rank attaches, rank 0 unlinks the name once all
    have (the segment lives until the last rank unmaps it).  Afterwards ``engine.evaluate_sharded`` / the ``configure``
    closure publish this rank's ~1 KiB record there and poll the other ranks' stamps: the records end up in host memory
    anyway, so this costs a few cache-line transfers between host cores where an all-gather costs a launch plus the
    collective's small-message latency.  All ranks must be processes of ONE node."""
    import os
    import uuid

    import torch.distributed as dist

    from . import _native as N

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [None]
    if rank == 0:
        box[0] = f"/gwi_{os.getpid()}_{uuid.uuid4().hex[:12]}"
    dist.broadcast_object_list(box, src=0, group=group)
    engine.shm_comm_init(box[0], rank, world)
    dist.barrier(group=group)
    if rank == 0:
        N.load_library().gwi_shm_comm_unlink(box[0].encode())
    return box[0]


class ShardedLikelihood:
    """Wraps this rank's :class:`NativePopulationLikelihood` (built with ``rank=``/``world=``)."""

    def __init__(self, engine, total_inj, group=None, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.engine = engine
        self.total_inj = float(total_inj)
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        backend = dist.get_backend(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        self.device = device
        n = engine.partial_len
        self._send = torch.zeros(n, dtype=torch.float64, device=device)
        self._recv = torch.zeros(self.world * n, dtype=torch.float64, device=device)
        self._host = torch.zeros(n, dtype=torch.float64).pin_memory() if device.type == "cuda" else torch.zeros(n, dtype=torch.float64)

    def gather_records(self, record):
        """All-gather one partial record per rank -> (world, len) float64 array (same on all ranks)."""
        t = self.torch
        self._host.copy_(t.from_numpy(np.ascontiguousarray(record)))
        self._send.copy_(self._host, non_blocking=True)
        self.dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
        return self._recv.cpu().numpy().reshape(self.world, -1)

    def evaluate(self, theta, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False, want_grad=True):
        rec, lb, ln, lv = self.engine.eval_partial(theta)
        records = self.gather_records(rec)
        res = self.engine.combine(records, self.total_inj, nobs=self.engine.n_ev_global, marginalize_selection=marginalize_selection, min_neff_cut=min_neff_cut,
                                  max_variance_cut=max_variance_cut, want_grad=want_grad)
        # local per-event sites (this rank's events), global constant applied
        shift = res.summary.log_norm_const - np.log(self.engine.n_pe)
        return EvalResult(log_likelihood=res.log_likelihood, grad=res.grad, summary=res.summary, log_bfs=lb + shift, log_neffs=ln, variances=lv, norms=res.norms)
