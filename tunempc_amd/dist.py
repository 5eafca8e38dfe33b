"""Multi-GPU path: batches of independent tuning problems shard over the ranks of one node (one process
per GPU, torch.distributed; backend 'nccl' == RCCL over xGMI on ROCm, 'gloo' for the CPU tests).
Tuning problems are fully independent (the reference handles exactly one at a time, convexifier.py:36),
so there is no data-path collective during the solve; the only exchange is ONE all-gather of the
results (Hc [+ kappa/status]) at the end.  Stages of one problem are chained through
P_k/P_{k+1} (convexifier.py:335-336) and are never split across GPUs."""
import numpy as np


def shard_range(nb, rank, world):
    """Contiguous, balanced slice of a batch of nb problems owned by `rank`."""
    return (nb * rank) // world, (nb * (rank + 1)) // world


def all_gather_results(local, nb_total, group=None):
    """All-gather per-rank result tensors (dict of torch tensors whose dim 0 is the local batch) into
    full-batch tensors on every rank.  Shards may be ragged by one problem: they are padded to the
    maximum shard size for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_range(nb_total, r, world)[1] - shard_range(nb_total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    out = {}
    for key, t in local.items():
        pad = t
        if t.shape[0] < mx:
            pad = torch.cat([t, torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)], 0)
        pad = pad.contiguous()
        full = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(full, pad, group=group)
        parts = [full[r * mx: r * mx + sizes[r]] for r in range(world)]
        out[key] = torch.cat(parts, 0)
    return out


def convexify_batch_sharded(A, B, H, solve_fn, group=None, keys=('Hc', 'kappa', 'status'), extra=None):
    """A, B, H: full-batch torch tensors (every rank holds or can generate the full batch; only its slice is
    read).  solve_fn(A_loc, B_loc, H_loc, **extra_loc) -> dict of torch tensors (local batch leading).  `extra`: dict of
    further per-problem inputs sliced the same way -- the equality-/active-constraint Jacobians and row counts of
    Step 1 with G and of Step 2 (G [nb,p,ng,n], C [nb,p,nc,n], ncnt [nb,p]).  Returns the gathered dict (full batch on
    every rank); add 'Fg' / 'F' to `keys` to gather the multipliers as well."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nb = A.shape[0]
    lo, hi = shard_range(nb, rank, world)
    loc = solve_fn(A[lo:hi], B[lo:hi], H[lo:hi], **{k: v[lo:hi] for k, v in (extra or {}).items()})
    return all_gather_results({k: loc[k] for k in keys}, nb, group)
