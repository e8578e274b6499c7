"""Multi-GPU layout of the rollout path: one process per GPU, independent candidate
trajectories sharded in contiguous blocks, the GP pack replicated.

The rollout itself needs no exchange.  Two collectives exist around it, both via
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests):

* ``broadcast_kinv``  - the inverse kernel matrices from rank 0 when data / hypers change, so that
  every replica of the pack is built from the same bits;
* ``gather_results`` - ONE fused all_gather of [cost | grad] per evaluation (B (1 + H da) doubles,
  latency bound), which is what a solver on any rank needs back.
"""
import torch


def shard_range(n_items, world, rank):
    """Contiguous block [lo, hi) of rank ``rank``; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_items, world):
    return [shard_range(n_items, world, r)[1] - shard_range(n_items, world, r)[0] for r in range(world)]


def broadcast_kinv(kinv, dist, src=0):
    dist.broadcast(kinv, src=src)
    return kinv


def gather_results(cost, grad, dist, sizes=None):
    """Fused all_gather of per-rank results.  cost (b,), grad (b, H, da) or None.
    Equal shard sizes use one all_gather_into_tensor; ragged shards pad to the largest."""
    world = dist.get_world_size()
    b = cost.shape[0]
    # explicit width: reshape(0, -1) is ambiguous for the empty block of a rank that owns no trajectory (world > B)
    gw = 1
    if grad is not None:
        for n in grad.shape[1:]:
            gw *= int(n)
    flat = cost.reshape(b, 1) if grad is None else torch.cat((cost.reshape(b, 1), grad.reshape(b, gw)), dim=1)
    width = flat.shape[1]
    if sizes is None:
        sizes = [b] * world
    bmax = max(sizes)
    if b < bmax:
        flat = torch.cat((flat, flat.new_zeros(bmax - b, width)), dim=0)
    flat = flat.contiguous()
    out = flat.new_empty((world * bmax, width))
    if flat.is_cuda and dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, flat)
    else:                                   # gloo: list form
        parts = [flat.new_empty((bmax, width)) for _ in range(world)]
        dist.all_gather(parts, flat)
        out = torch.cat(parts, dim=0)
    rows = torch.cat([out[r * bmax:r * bmax + sizes[r]] for r in range(world)], dim=0) if min(sizes) < bmax else out
    cost_all = rows[:, 0].contiguous()
    grad_all = None if grad is None else rows[:, 1:].reshape(rows.shape[0], *grad.shape[1:]).contiguous()
    return cost_all, grad_all


def sharded_rollout(rollout_fn, x0, U, dist, want_grad=True):
    """Evaluate a GLOBAL batch of candidate trajectories across the ranks of ``dist``: every rank calls this with the
    same (x0, U); rank r evaluates its contiguous block with ``rollout_fn(x0_block, U_block) -> dict(cost, grad)`` and
    all ranks receive the full (cost, grad).  x0: (B, ds) or (ds,); U: (B, H, da)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    B = U.shape[0]
    lo, hi = shard_range(B, world, rank)
    sizes = shard_sizes(B, world)
    x0_blk = x0 if getattr(x0, "ndim", 1) == 1 else x0[lo:hi]
    if hi > lo:
        r = rollout_fn(x0_blk, U[lo:hi])
        cost, grad = r["cost"], (r["grad"] if want_grad else None)
    else:                                   # more ranks than trajectories: contribute an empty block
        cost = U.new_zeros((0,)) if isinstance(U, torch.Tensor) else torch.zeros(0, dtype=torch.float64)
        grad = None if not want_grad else torch.zeros((0,) + tuple(U.shape[1:]), dtype=torch.float64, device=cost.device)
    return gather_results(cost, grad, dist, sizes)
