"""Multi-GPU plumbing: one process per GPU, batches shard along the batch axis.

Every effect treats batch items independently (per-item controls, filters and impulse responses;
dasp_pytorch/functional.py:189-208, :330-336, :542-548), so the data path needs **no collective**.
The only exchange in the reference's use case is the gradient all-reduce of the *networks* that
predict the controls (examples/style_transfer.py:331-380 trains them with plain Adam on one GPU):
`allreduce_gradients` is the data-parallel version of that step -- gradients flattened into a few
large buckets (RCCL over xGMI is per-link bound on point-to-point links, so few large messages beat
many small ones), summed, averaged, and scattered back.

torch.distributed backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """Initialise the default process group if WORLD_SIZE > 1. Returns (rank, world)."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def shard_bounds(n_items, world, rank):
    """Contiguous batch shard [lo, hi) of rank `rank`: sizes differ by at most one, every item is owned once."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(t, world=None, rank=None):
    """The slice of a (B, ...) tensor this rank owns."""
    if world is None:
        rank, _, world = env_world()
    lo, hi = shard_bounds(t.shape[0], world, rank)
    return t[lo:hi]


def max_over_ranks(value, device="cpu"):
    """Max of a python float over all ranks (bench timing: the slowest rank defines the step time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_gradients(params, bucket_bytes=64 << 20, average=True):
    """Sum (and average) .grad of `params` over all ranks with bucketed flat all-reduces.
    Gradients that are None are treated as zeros so that every rank issues the same collectives."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    params = [p for p in params if p.requires_grad]
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nbytes = p.numel() * p.element_size()
        if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    handles = []
    for bucket in buckets:   # launch every bucket asynchronously, then unpack in order
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        handles.append((bucket, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)))
    for bucket, flat, h in handles:
        h.wait()
        if average:
            flat.div_(world)
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
    return len(buckets)
