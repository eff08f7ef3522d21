"""Data-parallel pretraining: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" in CPU tests).  The reference has no distributed code; semantics here = the reference run on the
GLOBAL batch: every rank back-propagates the SUM-loss gradient of its local batch, then ONE all-reduce (sum) of the
buffer [flat gradient | loss statistics (sum|y-p|, kept count, KL sum, -)] makes both the gradient and the global
kept-count available everywhere; the optimiser kernel divides the reconstruction-path gradient by the global count
(masked-MAE is a mean over kept cells of the whole batch, lib/metrics.py:18) and leaves the KL path (a sum) unscaled.
Masks follow the reference's GLOBAL-batch semantics (GPTST.py:316-321, 351-404: one top-k / one class selection over
the whole batch): every rank draws the same global noise (same Philox seed), runs the bit-exact integer selection on the
global batch and keeps its own rows.  The random phase needs no communication; the adaptive phase all-gathers the
per-cell cluster labels (int32, 261 KB per 32-sample rank) between the two hipGraphs of a step (step.py: part 1 ends with the
guide classifier, part 2 starts with the mask); the per-class counts are the histogram of the gathered labels (taken on the device).
``PretrainStep(global_mask=False)`` falls back to per-rank masks (same ratio, no label exchange).
"""
import os

import torch
import torch.distributed as dist


class DataParallel:
    """native=True: the per-step collectives (gradient all-reduce, label gather) run on the C-ABI communicator (NativeComm: RCCL enqueues on
    torch's current stream) and are therefore CAPTURABLE — the step, its collectives and the optimiser are then ONE hipGraph per phase
    (step.py).  torch.distributed stays for the rendezvous, the initial weight broadcast and the host-side barrier / max of the bench."""

    def __init__(self, backend=None, native=False):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        backend = os.environ.get("GPTST_DIST_BACKEND", backend)       # tests: gloo to run several ranks on one GPU
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
        self.native = None
        if native:
            # The C-ABI communicator binds RCCL at run time.  Whether it can (library found, symbols present) is probed on EVERY rank and the
            # answer is agreed on with one all-reduce BEFORE the unique id travels: a rank that cannot must not leave the others waiting in
            # the broadcast / ncclCommInitRank — the job then runs its collectives through torch.distributed between graph replays instead.
            import ctypes
            import sys
            from . import _C
            ok = 1
            try:
                _C.lib().call("gptst_comm_available")               # side-effect free (binds RCCL; no ncclGetUniqueId bootstrap root)
            except Exception as e:                                    # noqa: BLE001
                ok, why = 0, "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:160] if str(e) else "")
            if self.world > 1:
                t = torch.tensor([ok], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                agreed = int(t.item())
            else:
                agreed = ok
            if agreed:
                self.native = NativeComm(rank=self.rank, world=self.world)
            elif self.rank == 0 or not ok:
                print("gpt-st_amd: the C-ABI RCCL communicator is not available%s -> torch.distributed collectives between graph replays"
                      % ((" here (" + why + ")") if not ok else " on another rank"), file=sys.stderr)
        self._slots = None

    @property
    def capturable(self):
        """the per-step collectives are plain stream-ordered enqueues (no torch.distributed call): they can sit inside a hipGraph"""
        return self.native is not None

    def rccl_ranks(self):
        """ranks RCCL reports for the native communicator (ncclCommCount), or the torch.distributed world size under nccl; None on gloo"""
        if self.native is not None:
            return self.native.count()
        return dist.get_world_size() if dist.get_backend() == "nccl" else None

    def allreduce_(self, buf):
        """In-place sum over ranks of the packed [gradient | statistics] buffer (one collective per step)."""
        if self.native is not None:
            self.native.allreduce_(buf)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        return buf

    def gather_labels(self, local, out=None):
        """Concatenate the ranks' label vectors in rank order (= batch-major order of the global batch)."""
        if out is None:
            out = torch.empty(self.world * local.numel(), dtype=local.dtype, device=local.device)
        if self.native is not None and self.native.allgather_i32(local, out):
            return out                                  # one RCCL launch on the C-ABI communicator
        if self.native is not None:
            # (ncclAllGather not available) all-gather as an all-reduce of a slot buffer: every rank fills its own slot with its labels as exactly representable floats
            # (the C-ABI communicator reduces fp32 only; labels < HS <= 64)
            if self._slots is None or self._slots.numel() != out.numel():
                self._slots = torch.zeros(out.numel(), dtype=torch.float32, device=local.device)
            self._slots.zero_()
            self._slots[self.rank * local.numel():(self.rank + 1) * local.numel()].copy_(local)
            self.native.allreduce_(self._slots)
            out.copy_(self._slots)
            return out
        if self.world == 1:
            out.copy_(local)
            return out
        dist.all_gather_into_tensor(out, local.contiguous())
        return out

    def sum_counts_(self, counts):
        """In-place sum over ranks of the per-class cell counts (int32)."""
        if self.world > 1:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        return counts

    def rows_of(self, flat_global, per_rank):
        """This rank's contiguous slice of a batch-major global vector."""
        return flat_global[self.rank * per_rank:(self.rank + 1) * per_rank]

    def broadcast_(self, t, src=0):
        dist.broadcast(t, src=src)
        return t

    def barrier(self):
        dist.barrier()

    def max_over_ranks(self, x):
        t = torch.tensor([float(x)], dtype=torch.float64, device="cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])


def combine_local_gradients(local_bufs, nA, nB):
    """Reference semantics of the packed-buffer reduction, written out for tests: given each rank's
    [sum-loss gradient | stats] buffer, return the global-mean gradient the optimiser applies."""
    total = torch.stack(local_bufs).sum(0)
    n = total.numel() - 8
    g, stats = total[:n].clone(), total[n:]
    g[:nA] /= max(float(stats[1]), 1.0)
    return g, stats


COMM_TIMER = None     # bench.py sets this to a list: (what, bytes, start_event, end_event) per collective enqueued through NativeComm


def _timed(what, nbytes, fn):
    if COMM_TIMER is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    COMM_TIMER.append((what, nbytes, e0, e1))
    return r


class NativeComm:
    """The C-ABI communicator (csrc/comm.hip: RCCL with an explicit stream).  The 128-byte unique id is created on rank 0 and handed to
    the other ranks through an existing torch.distributed group (any backend) — the only use of torch.distributed here; the collectives
    themselves are plain enqueues on a HIP stream, so they can be captured inside a hipGraph together with the kernels around them."""

    def __init__(self, rank=None, world=None, group=None, src=0):
        """rank / world: this process's rank in the communicator and its size (default: the job's RANK / WORLD_SIZE).
        group / src: a torch.distributed group that holds exactly the communicator's ranks and the GLOBAL rank of its rank 0 — for the
        sub-communicators of a mesh (mesh_comms below); the 128-byte id travels over that group."""
        import ctypes
        from . import _C
        self.lib = _C.lib()
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        uid = (ctypes.c_char * 128)()
        if self.rank == 0:
            self.lib.call("gptst_comm_unique_id", uid)
        if self.world > 1:
            t = torch.tensor(list(bytes(uid)), dtype=torch.uint8)
            if dist.get_backend(group) == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=src, group=group)
            uid = (ctypes.c_char * 128).from_buffer_copy(bytes(t.cpu().tolist()))
        self.h = ctypes.c_void_p()                       # communicator handle (r04: a process may hold several)
        self.lib.call("gptst_comm_init", self.rank, self.world, uid, ctypes.byref(self.h))

    def allreduce_(self, buf):
        """in-place sum over the ranks, enqueued on torch's current stream"""
        assert buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()
        _timed("allreduce_f32", 4 * buf.numel(),
               lambda: self.lib.call("gptst_allreduce_f32", self.h, buf.data_ptr(), buf.numel(), torch.cuda.current_stream().cuda_stream))
        return buf

    def allgather_i32(self, local, out):
        """out[r*n:(r+1)*n] <- rank r's int32 `local`, enqueued on torch's current stream; False when the library cannot (no ncclAllGather)"""
        assert local.is_cuda and local.dtype == torch.int32 and local.is_contiguous() and out.dtype == torch.int32 and out.is_contiguous()
        assert out.numel() == self.world * local.numel()
        if getattr(self, "_no_allgather", False):
            return False
        rc = _timed("allgather_i32", 4 * out.numel(),
                    lambda: self.lib.value("gptst_allgather_i32", self.h, local.data_ptr(), out.data_ptr(), local.numel(), torch.cuda.current_stream().cuda_stream))
        if rc == -4:                                    # GPTST_ECOMM: symbol missing
            self._no_allgather = True
            return False
        if rc != 0:
            raise RuntimeError("gptst_allgather_i32 failed: %d" % rc)
        return True

    def count(self):
        """ncclCommCount of the communicator"""
        import ctypes
        n = ctypes.c_int(0)
        self.lib.call("gptst_comm_count", self.h, ctypes.byref(n))
        return int(n.value)

    def close(self):
        if self.h:
            self.lib.call("gptst_comm_destroy", self.h)
            self.h = None


def mesh_shape(world, n_shard):
    """(data-parallel size, node-shard size) of a world laid out as a row-major (dp, shard) grid: rank = dp_index * n_shard + shard_index."""
    assert n_shard >= 1 and world % n_shard == 0, (world, n_shard)
    return world // n_shard, n_shard


def mesh_groups(world, n_shard):
    """The rank lists of the mesh's sub-communicators: ([ranks of each node-shard group (a row)], [ranks of each data-parallel group (a column)])."""
    n_dp, n_shard = mesh_shape(world, n_shard)
    rows = [[d * n_shard + s for s in range(n_shard)] for d in range(n_dp)]
    cols = [[d * n_shard + s for d in range(n_dp)] for s in range(n_shard)]
    return rows, cols


def mesh_comms(n_shard, rank=None, world=None):
    """SURVEY 8(e) "Combination": a B x N mesh — node shards inside a row (the cluster-aggregation all-reduces of shard.py), batch data
    parallelism down a column (the gradient all-reduce / label gather of this module).  -> (shard_comm, dp_comm): two C-ABI communicators of
    this rank (NativeComm), created over torch.distributed sub-groups (every rank calls new_group for every group, as torch requires)."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    rows, cols = mesh_groups(world, n_shard)
    mine = [None, None]
    for k, groups in enumerate((rows, cols)):
        for ranks in groups:
            pg = dist.new_group(ranks) if world > 1 else None
            if rank in ranks:
                mine[k] = NativeComm(rank=ranks.index(rank), world=len(ranks), group=pg, src=ranks[0])
    return mine[0], mine[1]
