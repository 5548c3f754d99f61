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


def init(backend=None, device=None, force=False):
    """Initialise the default process group if WORLD_SIZE > 1 - or, with force=True, at world size 1 as well (a one-rank RCCL / gloo group:
    the collectives then run through the backend instead of being short-cut, which is how the 1-GPU box exercises the RCCL path).
    Returns (rank, world)."""
    rank, local, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
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
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)       # (a one-rank group goes through the backend as well)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_gradients(params, bucket_bytes=64 << 20, average=True, force=False):
    """Sum (and average) .grad of `params` over all ranks with bucketed flat all-reduces, after backward() has returned (the simple form;
    GradientBuckets below overlaps the same collectives with the backward pass and needs no flatten / copy-back).
    Gradients that are None are treated as zeros so that every rank issues the same collectives. force=True runs the collectives on a
    one-rank group too (test hook: the RCCL path on a 1-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
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


class GradientBuckets:
    """The data-parallel gradient exchange of BASELINE config 5, overlapped with the backward pass ("flatten into buckets;
    overlap with the tail of backward").

    * Storage: the gradients of `params` live in a few flat buffers (one per bucket, `bucket_bytes` each, filled in REVERSE parameter
      order - the order in which a backward pass produces them); every `p.grad` is a view into its bucket, so autograd accumulates
      straight into the buffer that is all-reduced: no torch.cat before the collective, no copy back after it.
    * Overlap: a post-accumulate-grad hook per parameter counts a bucket's outstanding gradients; the moment a bucket is complete (and every
      earlier bucket has been launched: all ranks issue the collectives in the same order) its all-reduce is launched asynchronously -
      RCCL runs it on its own stream while autograd keeps computing the earlier layers' gradients. `finish()` after `backward()` launches
      whatever is left (parameters that got no gradient this step count as zeros), waits, and averages.
    * Use `zero_grad()` of this object instead of the optimizer's (`set_to_none` would detach the views).
    * Inside a HIP-graph capture the hooks stay silent (a collective must not be captured); call `finish()` after the replay - the exchange
      then follows the backward pass instead of overlapping it.
    * One step = zero_grad() -> backward() -> finish(). The launch state belongs to the step: `zero_grad()` first waits for collectives a
      previous step left in flight (a backward() that was never finished) and starts the count afresh, and a second backward() in the same
      step - before finish() - raises from its first hook instead of all-reducing half-accumulated buckets (round 4, advisor: three
      un-finished warm-up passes left `_next` at the end, and the first real step's gradients were never exchanged). Passes whose
      gradients are only to be accumulated locally (warm-up, gradient accumulation) run under `no_sync()`: the hooks are silent and the
      next `finish()` exchanges the sum.

    Inactive (hooks do nothing, `finish()` returns 0) when no process group is initialised or it has one rank - unless force=True, which
    runs the collectives on a one-rank group as well (how a 1-GPU box exercises the RCCL path)."""

    def __init__(self, params, bucket_bytes=16 << 20, average=True, force=False):
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.buckets = []                       # [(flat, [(param, view)])]
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._bucket_of = {id(p): i for i, (_, members) in enumerate(self.buckets) for p, _ in members}
        self._handles = []
        self._silent = 0
        self._begin_step()
        self.launched_in_backward = 0          # buckets whose all-reduce was issued from a hook (i.e. overlapped) in the last step
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _begin_step(self):
        self._left = [len(members) for _, members in self.buckets]
        self._next = 0                          # buckets [0, _next) have been launched in this step
        self._seen = set()                      # parameters whose gradient arrived in this step

    def _close(self, members):
        flat = torch.zeros(sum(p.numel() for p in members), dtype=members[0].dtype, device=members[0].device)
        views, off = [], 0
        for p in members:
            v = flat[off:off + p.numel()].view_as(p)
            if p.grad is not None:
                v.copy_(p.grad)
            p.grad = v
            views.append((p, v))
            off += p.numel()
        self.buckets.append((flat, views))

    @property
    def bytes(self):
        return sum(f.numel() * f.element_size() for f, _ in self.buckets)

    def _drain(self):
        """Wait for every collective in flight and average its bucket; returns how many there were."""
        for flat, h in self._handles:
            h.wait()
            if self.average and self.world > 1:
                flat.div_(self.world)
        n = len(self._handles)
        self._handles = []
        return n

    def zero_grad(self):
        """Start a step: collectives a previous, un-finished step still has in flight are waited for BEFORE the buffers are written."""
        self._drain()
        self._begin_step()
        for flat, members in self.buckets:
            flat.zero_()
            for p, v in members:
                if p.grad is not v:              # someone set it to None / replaced it: attach the view again
                    p.grad = v

    def no_sync(self):
        """Context manager: backward passes inside it only accumulate into the buckets (no collective is launched from the hooks, no
        second-backward check); the next finish() exchanges the accumulated sum. For warm-up passes and gradient accumulation."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self._silent += 1
            try:
                yield self
            finally:
                self._silent -= 1
        return cm()

    def _launch(self, i):
        flat = self.buckets[i][0]
        self._handles.append((flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)))

    def _on_grad(self, p):
        if not self.active or self._silent or (p.is_cuda and torch.cuda.is_current_stream_capturing()):
            return
        if id(p) in self._seen:
            raise RuntimeError("GradientBuckets: a parameter received a second gradient before finish() - backward() ran twice in one step. "
                               "Call finish() (or zero_grad()) between steps, or run accumulation / warm-up passes under no_sync().")
        self._seen.add(id(p))
        self._left[self._bucket_of[id(p)]] -= 1
        while self._next < len(self.buckets) and self._left[self._next] <= 0:
            self._launch(self._next)
            self._next += 1

    def finish(self):
        """Call after backward(): launches the buckets the hooks did not, waits for all of them, averages. Returns the number of buckets
        exchanged. What is launched here follows from this step's own count (`_next`, reset by zero_grad() / the previous finish())."""
        if not self.active:
            return 0
        overlapped = self._next
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1
        n = self._drain()
        self._begin_step()
        self.launched_in_backward = overlapped
        return n

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
