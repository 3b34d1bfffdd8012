"""Data parallelism for the SAR-Net step: one process per GPU, RCCL over xGMI via torch.distributed.

Replaces keras.utils.multi_gpu_model (model.py:193-194): each rank owns a full replica and a contiguous slice of
the global batch (same slicing as multi_gpu_model); BatchNorm statistics stay per-GPU (per-tower in the reference);
the flat fp32 gradient buffer (engine.ParamStore.grads) is summed across ranks and the mean is taken by the Adam
kernel's grad_scale = 1/world.  The buffer is reduced in contiguous buckets: parameters are laid out in
forward order, so the tail (heads + GRUs) is complete early in backward and the head (conv stack) last.
Everything here is device-agnostic torch.distributed code so the N>1 path is covered by gloo tests on CPU."""
import os
import torch
import torch.distributed as dist


def forced():
    """SAR_DP_FORCE=1: run the data-parallel step (five graphs, bucketed asynchronous all-reduce) even in a world of ONE rank --
    how a 1-GPU box exercises the real RCCL path (communicator, RCCL's stream under graph replays, work.wait()); the reduction of a
    one-rank group is the identity, so the step must agree with the single-graph step."""
    return os.environ.get("SAR_DP_FORCE", "0") == "1"


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or forced()) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("SAR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            # CU budget of the collective (round 6).  The step's conv launches are one block per CU on all 256 CUs, so every CU an
            # RCCL channel sits on sends that launch's displaced blocks into a second round: tools/cu_contention.py, cfg3 on one GPU -- the backward
            # segment the 33.6 MB all-reduce hides under (507 us) is +8 % beside 8 channel-like workgroups, +20 % beside 16, +34 % beside 64.  Eight
            # channels move the 59 MB a rank sends and receives in that all-reduce at 8 ranks in ~0.33 ms at 22 GB/s per channel: inside the segment.
            # NCCL_MAX_NCHANNELS in the environment overrides; no N > 1 box was available to any round to confirm the choice on links.
            os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
        elif torch.cuda.is_available():              # gloo with GPU tensors (debug on a 1-GPU box): all ranks share a device
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(batch, rank, world):
    """Contiguous equal slices of the global batch (what multi_gpu_model does with its Lambda slices)."""
    out = {}
    for k, v in batch.items():
        n = len(v); per = n // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out


def bucket_bounds(n, split):
    """Two buckets over a flat buffer of n elements: [split, n) first (ready early), then [0, split)."""
    split = max(0, min(n, int(split)))
    return [(split, n), (0, split)] if 0 < split < n else [(0, n)]


def allreduce_flat(flat, bounds, group=None, async_op=False):
    """Sum-all-reduce the given [lo, hi) ranges of a flat tensor in place.  Returns the work handles."""
    works = []
    for lo, hi in bounds:
        if hi > lo:
            works.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op))
    return works


def broadcast_flat(flat, src=0, group=None):
    dist.broadcast(flat, src=src, group=group)


class RcclComm:
    """The C-ABI collective (include/sarnet.h: sar_comm_* / sar_allreduce_bucket_f32) as DataParallel uses it when
    SAR_DP_COLLECTIVE=cabi: a communicator of its own (unique id made by rank 0, broadcast through the already-initialised
    torch.distributed group), all-reduces enqueued on a dedicated stream behind the work submitted so far, `wait()` = the
    current stream waits for that stream -- the same overlap contract as torch.distributed's asynchronous work handles."""

    class _Work:
        def __init__(self, stream):
            self.stream = stream

        def wait(self):
            torch.cuda.current_stream().wait_stream(self.stream)

    def __init__(self):
        import ctypes
        from . import _lib
        self.fn = _lib.lib().fn
        rank, world = dist.get_rank(), dist.get_world_size()
        ident = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            if self.fn["sar_comm_unique_id"](buf) != 0:
                raise RuntimeError("sar_comm_unique_id failed (is librccl loadable?)")
            ident = [buf.raw]
        dist.broadcast_object_list(ident, src=0)
        self.handle = ctypes.c_void_p()
        if self.fn["sar_comm_init"](ctypes.byref(self.handle), ident[0], world, rank) != 0:
            raise RuntimeError("sar_comm_init failed (rank %d of %d)" % (rank, world))
        self.stream = torch.cuda.Stream()

    def allreduce(self, t, scratch16=None):
        """scratch16 (a bf16 tensor of t's length): the bucket travels as bf16 (sar_allreduce_bucket_bf16), the sum lands in t as fp32."""
        self.stream.wait_stream(torch.cuda.current_stream())
        if scratch16 is not None:
            if self.fn["sar_allreduce_bucket_bf16"](self.handle, t.data_ptr(), scratch16.data_ptr(), t.numel(), self.stream.cuda_stream) != 0:
                raise RuntimeError("sar_allreduce_bucket_bf16 failed")
        elif self.fn["sar_allreduce_bucket_f32"](self.handle, t.data_ptr(), t.numel(), self.stream.cuda_stream) != 0:
            raise RuntimeError("sar_allreduce_bucket_f32 failed")
        return self._Work(self.stream)

    def close(self):
        if self.handle:
            torch.cuda.synchronize()
            self.fn["sar_comm_destroy"](self.handle); self.handle = None


class _Bf16Work:
    """Work handle of a bucket that travelled as bf16 through torch.distributed: wait() = wait for the collective, then widen the sum
    back into the fp32 gradient range (stream-ordered, in front of whatever reads the gradients next)."""

    def __init__(self, work, dst32, src16):
        self.work, self.dst32, self.src16 = work, dst32, src16

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.dst32.copy_(self.src16)


class DataParallel:
    def __init__(self, store, split_name_prefix="CNN_LIN", trunk_splits=()):
        """Buckets of the flat gradient buffer in the order backward completes them: [first post-trunk parameter, n) (heads,
        pooling, GRUs), then the conv trunk from its end towards its start, cut at the parameters named in `trunk_splits`
        (ResNetTrunk.bucket_split_names(): stage 3, stage 2), the rest last."""
        self.store = store
        # (SAR_DP_DISABLE=1: a replica that takes no part in any collective although a process group exists -- bench.py times the plain
        # single-GPU step of the same box beside the data-parallel one)
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced()) and os.environ.get("SAR_DP_DISABLE", "0") != "1"
        self.world = dist.get_world_size() if self.enabled else 1
        self.rank = dist.get_rank() if self.enabled else 0
        split = store.n
        for r in store.refs:
            if r.name.startswith(split_name_prefix):
                split = r.offset; break
        self.bounds = bucket_bounds(store.n, split)
        offs = {r.name: r.offset for r in store.refs}
        cuts = sorted((offs[n] for n in trunk_splits if n in offs and 0 < offs[n] < split), reverse=True)
        if cuts and len(self.bounds) == 2:
            lo_hi, hi = [], split
            for c in cuts:
                lo_hi.append((c, hi)); hi = c
            lo_hi.append((0, hi))
            self.bounds = [self.bounds[0]] + lo_hi
        # SAR_DP_COLLECTIVE=cabi: the gradient buckets go through the library's own RCCL entry point instead of torch.distributed's
        self.cabi = RcclComm() if (self.enabled and os.environ.get("SAR_DP_COLLECTIVE") == "cabi" and store.grads.is_cuda) else None
        # SAR_DP_GRAD_DTYPE=bf16 (off by default): the buckets cross the links as bf16 -- half the bytes, 17 MB instead
        # of 34 at F = 32 -- and the sum is widened back into the fp32 gradient buffer that Adam reads.  Every hop of a ring all-reduce then
        # rounds its partial sum to 8 mantissa bits: a bandwidth / precision trade for link-bound configurations, to be judged on hardware.
        self.grad16 = None
        if self.enabled and os.environ.get("SAR_DP_GRAD_DTYPE", "fp32") == "bf16":
            self.grad16 = torch.empty(store.n, dtype=torch.bfloat16, device=store.grads.device)
        if self.enabled:
            self.broadcast_params()

    def broadcast_params(self):
        if self.enabled:
            broadcast_flat(self.store.params, 0)
            for r in self.store.state_refs:
                dist.broadcast(r.w, src=0)

    def allreduce_grads(self):
        if self.enabled:
            for lo, hi in self.bounds:
                [w.wait() for w in self._reduce(lo, hi, False)]

    def _reduce(self, lo, hi, async_op):
        """Sum-all-reduce of grads[lo:hi] by the configured route; returns the work handles still to be waited for."""
        g = self.store.grads[lo:hi]
        if self.cabi is not None:
            w = self.cabi.allreduce(g, None if self.grad16 is None else self.grad16[lo:hi])
            if not async_op:
                w.wait()
            return [w] if async_op else []
        if self.grad16 is not None:
            t = self.grad16[lo:hi]
            t.copy_(g)                                                  # fp32 -> bf16, round to nearest even
            w = _Bf16Work(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op), g, t)
            if not async_op:
                w.wait()
            return [w] if async_op else []
        return [w for w in allreduce_flat(self.store.grads, [(lo, hi)], async_op=async_op) if w is not None]

    def allreduce_bucket(self, i, async_op=True):
        """Launch the sum-all-reduce of bucket i (0 = tail: heads/GRUs, complete first in backward; 1.. = conv stack, late
        stages first).
        With RCCL the collective is enqueued on RCCL's stream behind the work already submitted to the current stream, so
        it overlaps whatever the caller submits next; `work.wait()` makes the current stream wait for it (no host block).
        A model without post-trunk parameters has a single bucket: index 1 is then empty."""
        if not self.enabled or i >= len(self.bounds):
            return []
        lo, hi = self.bounds[i]
        return self._reduce(lo, hi, async_op) if hi > lo else []

    def allreduce_range(self, lo, hi, async_op=True):
        """Sum-all-reduce of grads[lo:hi] (several adjacent buckets as one collective); same stream semantics as allreduce_bucket."""
        if not self.enabled or hi <= lo:
            return []
        return self._reduce(lo, hi, async_op)
