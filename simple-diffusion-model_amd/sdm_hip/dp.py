"""Bucketed gradient all-reduce for data-parallel training (one process per GPU, RCCL over xGMI).

The U-Net backward finishes its top-level blocks in a fixed order and the flat gradient buffer is laid out in
that order, so a bucket is a contiguous slice that becomes final when its last block reports done.  Each
bucket's sum-all-reduce is issued asynchronously at that moment (it overlaps the rest of the backward); the
optimizer consumes buckets in order, waiting only for the bucket it is about to update.

Pure torch.distributed: importable (and tested with gloo, world_size 2) without a GPU.
"""
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


class BucketedAllReduce:
    def __init__(self, flat, block_ranges, bucket_mb=64, group=None):
        self.flat, self.group = flat, group
        self.buckets = make_buckets(block_ranges, int(bucket_mb * (1 << 20)), flat.element_size())
        self.closing = {b[3]: i for i, b in enumerate(self.buckets)}
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.pending = []

    def begin_step(self):
        self.pending = []

    def block_done(self, name):
        """Call when every gradient of top-level block ``name`` has been written (in completion order)."""
        i = self.closing.get(name)
        if i is None:
            return
        s, e = self.buckets[i][0], self.buckets[i][1]
        work = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True) if self.world > 1 else None
        self.pending.append((i, work))

    def drain(self):
        """Yield (bucket_index, start, end) in issue order, each after its all-reduce has landed."""
        for i, work in self.pending:
            if work is not None:
                work.wait()
            yield i, self.buckets[i][0], self.buckets[i][1]
        self.pending = []
