"""Data-parallel gradient exchange (sdm_hip/dp.py) with world_size 2 on gloo / CPU: the bucket plan covers the
flat buffer exactly once in completion order, and after block_done()/drain() every rank holds the SUM of the
ranks' gradients (the trainer then scales by 1/world inside the fused Adam)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "simple-diffusion-model_amd"))
from sdm_hip.dp import BucketedAllReduce, make_buckets  # noqa: E402

BLOCKS = [("out_layers", 0, 1000), ("up_layers.1", 1000, 70000), ("up_layers.0", 70000, 90000), ("middle_layer", 90000, 90010),
          ("down_layers.1", 90010, 150000), ("down_layers.0", 150000, 150500), ("in_layer", 150500, 151000),
          ("cond_emb", 151000, 160000)]


def test_bucket_plan_covers_buffer_once():
    b = make_buckets(BLOCKS, 128 * 1024)
    assert b[0][0] == 0 and b[-1][1] == 160000
    for x, y in zip(b, b[1:]):
        assert x[1] == y[0]
    assert all((e - s) * 4 >= 128 * 1024 for s, e, _, _ in b[:-1])
    assert make_buckets(BLOCKS, 1 << 40) == [(0, 160000, "out_layers", "cond_emb")]
    with pytest.raises(ValueError):
        make_buckets([("a", 0, 10), ("b", 11, 20)], 1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat = torch.arange(160000, dtype=torch.float32) * (rank + 1)
        red = BucketedAllReduce(flat, BLOCKS, bucket_mb=0.125)
        red.begin_step()
        seen = []
        for name, s, e in BLOCKS:            # "backward": blocks finish in order; later blocks are still being written
            flat[s:e] += 0.5 * (rank + 1)
            red.block_done(name)
        for i, s, e in red.drain():
            seen.append((s, e))
        expect = (torch.arange(160000, dtype=torch.float32) + 0.5) * sum(r + 1 for r in range(world))
        ok = bool(torch.equal(flat, expect)) and seen[0][0] == 0 and seen[-1][1] == 160000
        q.put((rank, ok, len(seen)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert len({n for _, _, n in res}) == 1
