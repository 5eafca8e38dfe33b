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


def all_gather_results(local, nb_total, group=None, cache=None):
    """All-gather per-rank result tensors (dict of torch tensors whose dim 0 is the local batch) into
    full-batch tensors on every rank.  Shards may be ragged by one problem: they are padded to the
    maximum shard size for the collective and trimmed afterwards.
    `cache` (a dict the caller keeps between calls): the gather and padding buffers are allocated once per key and re-used -- a
    serving loop gathers into the same world x shard buffer every step instead of allocating 2 GB per step at the bench shape
    (8 ranks x 268 MB of Hc).  With equal shards the returned tensors ARE those buffers: valid until the next call with the
    same cache."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_range(nb_total, r, world)[1] - shard_range(nb_total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    even = min(sizes) == mx
    out = {}

    def buf(kind, key, shape, t):
        if cache is None:
            return torch.empty(shape, dtype=t.dtype, device=t.device)
        k = (kind, key, tuple(shape), t.dtype, str(t.device))
        b = cache.get(k)
        if b is None:
            for old in [c for c in cache if c[0] == kind and c[1] == key]:      # a key changes shape: drop the stale buffer
                del cache[old]
            b = cache[k] = torch.empty(shape, dtype=t.dtype, device=t.device)
        return b

    for key, t in local.items():
        pad = t
        if t.shape[0] < mx:
            pad = buf('pad', key, (mx,) + tuple(t.shape[1:]), t)
            pad[:t.shape[0]].copy_(t)
            pad[t.shape[0]:].zero_()
        pad = pad.contiguous()
        full = buf('full', key, (world * mx,) + tuple(t.shape[1:]), t)
        dist.all_gather_into_tensor(full, pad, group=group)
        if even:
            out[key] = full
        else:
            out[key] = torch.cat([full[r * mx: r * mx + sizes[r]] for r in range(world)], 0)
    return out


def balanced_assignment(cost, world):
    """Straggler-aware split ("dynamic chunking of the local batch to even out iteration counts"): problems sorted by a cost proxy,
    most expensive first, and dealt to the ranks in snake order (0, 1, .., w-1, w-1, .., 1, 0, ...) -- the expensive members are spread over the ranks
    instead of sitting in one contiguous shard, and every rank gets nb // world or nb // world + 1 problems.  Returns a list of index arrays (ascending
    within a rank).  The iteration count of a member grows with the conditioning of its Hessians: `sbeta = max|eig H| / min|eig H|` (convexifier.py:383-399,
    one batched eigen-scan, tmpc_eig_scan_host) is the proxy bench-style callers have for free."""
    cost = np.asarray(cost, dtype=np.float64)
    order = np.argsort(-cost, kind='stable')
    ranks = [[] for _ in range(world)]
    for pos, idx in enumerate(order):
        rnd, off = divmod(pos, world)
        ranks[off if rnd % 2 == 0 else world - 1 - off].append(int(idx))
    return [np.sort(np.asarray(r, dtype=np.int64)) for r in ranks]


def convexify_batch_sharded(A, B, H, solve_fn, group=None, keys=('Hc', 'kappa', 'status'), extra=None, cost=None):
    """A, B, H: full-batch torch tensors (every rank holds or can generate the full batch; only its slice is
    read).  solve_fn(A_loc, B_loc, H_loc, **extra_loc) -> dict of torch tensors (local batch leading).  `extra`: dict of
    further per-problem inputs sliced the same way -- the equality-/active-constraint Jacobians and row counts of
    Step 1 with G and of Step 2 (G [nb,p,ng,n], C [nb,p,nc,n], ncnt [nb,p]).  Returns the gathered dict (full batch on
    every rank); add 'Fg' / 'F' to `keys` to gather the multipliers as well.
    cost [nb] (optional): a per-problem cost proxy -- the shards are then dealt by `balanced_assignment` instead of cut contiguously, and the gathered
    tensors are put back into the caller's order."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nb = A.shape[0]
    if cost is not None:
        parts = balanced_assignment(cost, world)
        mine = torch.as_tensor(parts[rank], device=A.device)
        sel = lambda t: t.index_select(0, mine.to(t.device))
        loc = solve_fn(sel(A), sel(B), sel(H), **{k: sel(v) for k, v in (extra or {}).items()})
        sizes = [len(q) for q in parts]
        mx = max(sizes)
        out = {}
        for key in keys:
            t = loc[key]
            pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[:t.shape[0]] = t
            full = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(full, pad.contiguous(), group=group)
            res = torch.empty((nb,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            for r in range(world):
                res[torch.as_tensor(parts[r], device=t.device)] = full[r * mx: r * mx + sizes[r]]
            out[key] = res
        return out
    lo, hi = shard_range(nb, rank, world)
    loc = solve_fn(A[lo:hi], B[lo:hi], H[lo:hi], **{k: v[lo:hi] for k, v in (extra or {}).items()})
    return all_gather_results({k: loc[k] for k in keys}, nb, group)
