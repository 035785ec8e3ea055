"""RCCL sanity on one GPU (world_size 1): the exact call pattern of the data-parallel trainer -- nccl backend bound to
the device, barrier, async all-reduce of a slice of a flat fp32 buffer issued under a side stream, work.wait() there."""
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29613")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dist.barrier()
    flat = torch.arange(32 << 20, dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        work = dist.all_reduce(flat[1 << 20:17 << 20], op=dist.ReduceOp.SUM, async_op=True)
        work.wait()
        flat[1 << 20:17 << 20].mul_(0.5)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    assert float(flat[(1 << 20) + 5]) == ((1 << 20) + 5) * 0.5
    dist.barrier()
    dist.destroy_process_group()
    print("rccl selftest ok")


if __name__ == "__main__":
    main()
