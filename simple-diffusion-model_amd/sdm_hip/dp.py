"""Bucketed gradient all-reduce for data-parallel training (one process per GPU, RCCL over xGMI).

The U-Net backward finishes its top-level blocks in a fixed order and the flat gradient buffer is laid out in
that order, so a bucket is a contiguous slice that becomes final when its last block reports done.  Each
bucket's sum-all-reduce is issued asynchronously at that moment (it overlaps the rest of the backward); the
optimizer consumes buckets in order, waiting only for the bucket it is about to update.

WHO moves the bytes is an *exchange* object (``make_exchange``):

  backend "torch" (default)  torch.distributed collectives on the caller's process group: ``nccl`` (== RCCL on ROCm) on
                             GPUs, ``gloo`` on CPU -- importable and tested with gloo, world_size 2, without a GPU.
  backend "sdm"              the communicator owned by libsdm_hip.so (sdm_dp_init / sdm_dp_allreduce_bucket /
                             sdm_dp_finalize, csrc/sdm_dp.hip): the exchange is enqueued from C on the stream it is
                             given, RCCL errors come back through the library's error convention (SdmError).  The
                             128-byte unique id is made on rank 0 and broadcast over the torch process group, which is
                             the launcher's rendezvous only.

and HOW (``wire``):

  "f32"   one fp32 sum-all-reduce per bucket (RCCL's ring / tree).
  "bf16"  the mesh exchange for the switchless, fully connected xGMI topology: the bucket is cut into `world` shards,
          every shard travels as bf16 straight to the rank that owns it (all-to-all: W - 1 point-to-point transfers per
          rank, one per link), is summed there in FP32 over the ranks in rank order (the owner's own contribution
          unrounded), rounded once, and all-gathered as bf16.  Half the bytes of the fp32 exchange on every link, two
          roundings per element whatever the world size, deterministic.  TorchExchange implements the same arithmetic
          with two all-gathers (any backend, incl. gloo): the CPU twin the tests pin the shard / rounding rules with.

No multi-GPU curve has been measured with either (no multi-GPU node in this build's pool): correctness by construction,
world-2 gloo tests, a world-1 RCCL test on the one GPU.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


def make_buckets(block_ranges, bucket_bytes, elem_bytes=4):
    """block_ranges: [(name, start, end)] contiguous and ordered.  Returns [(start, end, first_block, last_block)]
    where consecutive blocks are merged until a bucket holds at least ``bucket_bytes``."""
    buckets, cur, prev_end = [], None, None
    for name, s, e in block_ranges:
        if prev_end is not None and s != prev_end:
            raise ValueError("block ranges must be contiguous and in completion order")
        prev_end = e
        if cur is None:
            cur = [s, e, name, name]
        else:
            cur[1], cur[3] = e, name
        if (cur[1] - cur[0]) * elem_bytes >= bucket_bytes:
            buckets.append(tuple(cur))
            cur = None
    if cur is not None:
        buckets.append(tuple(cur))
    return buckets


def shard_elems(n, world):
    """Elements per shard of an n-element bucket: ceil(n / world) rounded up to 8 (16 bytes of bf16) -- csrc/sdm_dp.hip:dp_shard."""
    s = -(-n // world)
    return -(-s // 8) * 8


class _Done:
    """Handle of an exchange that is already ordered on the caller's stream (or finished): wait() is a no-op."""

    def wait(self):
        return True


class TorchExchange:
    """torch.distributed on ``group`` (nccl == RCCL, or gloo)."""
    backend = "torch"

    def __init__(self, group=None, wire="f32"):
        if wire not in ("f32", "bf16"):
            raise ValueError("wire must be 'f32' or 'bf16'")
        self.group, self.wire = group, wire
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def all_reduce(self, buf):
        """Sum ``buf`` (a contiguous fp32 slice) over the ranks, in place; returns a handle with wait()."""
        if self.world == 1 and self.wire == "f32":
            return None
        if self.wire == "f32":
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        # mesh arithmetic (see the module docstring) through two all-gathers
        W, r, n = self.world, self.rank, buf.numel()
        sh = shard_elems(n, W)
        mine = torch.zeros(W * sh, dtype=torch.float32, device=buf.device)
        mine[:n] = buf
        packed = mine.to(torch.bfloat16)
        if W > 1:
            every = [torch.empty_like(packed) for _ in range(W)]
            dist.all_gather(every, packed, group=self.group)
        else:
            every = [packed]
        acc = torch.zeros(sh, dtype=torch.float32, device=buf.device)
        for w in range(W):                                   # rank order; the owner's own term in fp32
            acc += mine[r * sh:(r + 1) * sh] if w == r else every[w][r * sh:(r + 1) * sh].float()
        red = acc.to(torch.bfloat16)
        if W > 1:
            parts = [torch.empty_like(red) for _ in range(W)]
            dist.all_gather(parts, red, group=self.group)
        else:
            parts = [red]
        buf.copy_(torch.cat(parts)[:n].float())
        return _Done()

    def check(self):
        pass

    def close(self):
        pass


class SdmExchange:
    """The communicator owned by libsdm_hip.so.  ``group`` is only the rendezvous the unique id travels over."""
    backend = "sdm"

    def __init__(self, group=None, wire="f32", device=None):
        from . import lib, ops
        if wire not in ("f32", "bf16"):
            raise ValueError("wire must be 'f32' or 'bf16'")
        self.lib, self.ops, self.wire = lib, ops, wire
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        lib.call("sdm_dp_available")                         # SdmError with the reason when RCCL cannot be loaded
        uid = C.create_string_buffer(128)
        if self.rank == 0:
            lib.call("sdm_dp_unique_id", uid, 128)
        if self.world > 1:
            box = [uid.raw]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = C.create_string_buffer(box[0], 128)
        if device is not None:
            torch.cuda.set_device(device)
        h = C.c_void_p()
        lib.call("sdm_dp_init", uid, 128, self.rank, self.world, C.byref(h))
        self.handle = h

    def reserve(self, max_elems):
        self.lib.call("sdm_dp_reserve", self.handle, int(max_elems))

    def all_reduce(self, buf):
        """Enqueued on the CURRENT torch stream: whatever the caller enqueues there next sees the sums."""
        assert buf.dtype == torch.float32 and buf.is_contiguous()
        self.lib.call("sdm_dp_allreduce_bucket", self.handle, buf.data_ptr(), buf.numel(), int(self.wire == "bf16"), 0,
                      self.ops._stream())
        return _Done()

    def check(self):
        """Raise SdmError if RCCL has recorded an asynchronous failure (a lost peer)."""
        self.lib.call("sdm_dp_async_error", self.handle)

    def close(self):
        if self.handle is not None:
            self.lib.call("sdm_dp_finalize", self.handle)
            self.handle = None


def make_exchange(group=None, backend=None, wire=None, device=None):
    """backend / wire default to the environment (SDM_DP_BACKEND = torch | sdm, SDM_DP_WIRE = f32 | bf16)."""
    backend = backend or os.environ.get("SDM_DP_BACKEND", "torch")
    wire = wire or os.environ.get("SDM_DP_WIRE", "f32")
    if backend == "torch":
        return TorchExchange(group, wire)
    if backend == "sdm":
        return SdmExchange(group, wire, device)
    raise ValueError("SDM_DP_BACKEND / backend must be 'torch' or 'sdm'")


class BucketedAllReduce:
    def __init__(self, flat, block_ranges, bucket_mb=64, group=None, exchange=None):
        self.flat, self.group = flat, group
        self.buckets = make_buckets(block_ranges, int(bucket_mb * (1 << 20)), flat.element_size())
        self.closing = {b[3]: i for i, b in enumerate(self.buckets)}
        self.exchange = exchange if exchange is not None else TorchExchange(group, "f32")
        self.world = self.exchange.world
        self.pending = []
        if hasattr(self.exchange, "reserve") and self.exchange.wire == "bf16":
            self.exchange.reserve(max(e - s for s, e, _, _ in self.buckets))

    def begin_step(self):
        self.pending = []

    def block_done(self, name):
        """Call when every gradient of top-level block ``name`` has been written (in completion order)."""
        i = self.closing.get(name)
        if i is None:
            return
        s, e = self.buckets[i][0], self.buckets[i][1]
        # (a one-rank run still goes through the library-owned communicator and the bf16 wire when those are selected:
        # the paths a multi-GPU run takes are then the ones a one-GPU test exercises)
        live = self.world > 1 or self.exchange.wire == "bf16" or self.exchange.backend == "sdm"
        work = self.exchange.all_reduce(self.flat[s:e]) if live else None
        self.pending.append((i, work))

    def drain(self):
        """Yield (bucket_index, start, end) in issue order, each after its all-reduce has landed."""
        for i, work in self.pending:
            if work is not None:
                work.wait()
            yield i, self.buckets[i][0], self.buckets[i][1]
        self.pending = []
