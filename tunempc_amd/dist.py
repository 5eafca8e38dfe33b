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


def all_gather_results(local, nb_total, group=None, cache=None, sizes=None):
    """All-gather per-rank result tensors (dict of torch tensors whose dim 0 is the local batch) into
    full-batch tensors on every rank.  Shards may be ragged by one problem: they are padded to the
    maximum shard size for the collective and trimmed afterwards.
    `cache` (a dict the caller keeps between calls): the gather and padding buffers are allocated once per key and re-used -- a
    serving loop gathers into the same world x shard buffer every step instead of allocating 2 GB per step at the bench shape
    (8 ranks x 268 MB of Hc).  With equal shards the returned tensors ARE those buffers: valid until the next call with the
    same cache.  sizes: the shard length of every rank when the split is not the contiguous one of shard_range (balanced assignment)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if sizes is None:
        sizes = [shard_range(nb_total, r, world)[1] - shard_range(nb_total, r, world)[0] for r in range(world)]
    assert len(sizes) == world and sum(sizes) == nb_total
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


def convexify_batch_sharded(A, B, H, solve_fn, group=None, keys=('Hc', 'kappa', 'status'), extra=None, cost=None, cache=None):
    """A, B, H: full-batch torch tensors (every rank holds or can generate the full batch; only its slice is
    read).  solve_fn(A_loc, B_loc, H_loc, **extra_loc) -> dict of torch tensors (local batch leading).  `extra`: dict of
    further per-problem inputs sliced the same way -- the equality-/active-constraint Jacobians and row counts of
    Step 1 with G and of Step 2 (G [nb,p,ng,n], C [nb,p,nc,n], ncnt [nb,p]).  Returns the gathered dict (full batch on
    every rank); add 'Fg' / 'F' to `keys` to gather the multipliers as well.
    cost [nb] (optional): a per-problem cost proxy -- the shards are then dealt by `balanced_assignment` instead of cut contiguously, and the gathered
    tensors are put back into the caller's order.  The assignment every rank uses is the one RANK 0 computes from its `cost` (one small broadcast), so
    ranks whose proxies differ in the last bit -- or at all -- cannot duplicate or drop a problem; `cost` must have one finite entry per problem.
    cache: as in all_gather_results (padding / gather / result buffers kept between calls)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    nb = A.shape[0]
    if cost is not None:
        # one assignment for everybody: rank 0 validates ITS cost and computes the permutation; an ok-flag travels with it (nb + 1 int64), so a bad proxy raises
        # on every rank together instead of leaving the others in the collective.  Only rank 0's cost is ever used: the other ranks neither check
        # theirs nor run the O(nb) deal.
        src = dist.get_global_rank(group, 0) if group is not None else 0
        msg = torch.zeros(nb + 1, dtype=torch.int64, device=A.device)
        err = None
        if rank == 0:
            c0 = np.asarray(cost.detach().cpu() if hasattr(cost, 'detach') else cost, dtype=np.float64).reshape(-1)
            if c0.shape != (nb,):
                err = 'cost must hold one entry per problem: expected shape ({},), got {}'.format(nb, c0.shape)
            elif not np.isfinite(c0).all():
                err = 'cost must be finite (a NaN would make the split meaningless)'
            else:
                msg[0] = 1
                msg[1:] = torch.as_tensor(np.concatenate(balanced_assignment(c0, world)), dtype=torch.int64)
        if world > 1:
            dist.broadcast(msg, src=src, group=group)
        if int(msg[0].item()) != 1:
            raise ValueError(err or "rank 0's cost proxy was rejected (wrong shape or not finite): no rank solved anything")
        flat = msg[1:]
        # snake dealing gives every rank nb // world problems and the first (even last round) or last (odd) nb % world ranks one more: known without the cost
        base, rem = divmod(nb, world)
        sizes = [base + (1 if ((r < rem) if base % 2 == 0 else (r >= world - rem)) else 0) for r in range(world)]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        idx = [flat[offs[r]:offs[r + 1]] for r in range(world)]
        seen = torch.zeros(nb, dtype=torch.int32, device=A.device)
        seen[flat] += 1
        if not bool((seen == 1).all()):
            raise RuntimeError('balanced assignment is not a permutation of the batch')
        sel = lambda t: t.index_select(0, idx[rank].to(t.device))
        loc = solve_fn(sel(A), sel(B), sel(H), **{k: sel(v) for k, v in (extra or {}).items()})
        gathered = all_gather_results({k: loc[k] for k in keys}, nb, group, cache, sizes=sizes)      # shards are joined in rank order: the order of `flat`
        out = {}
        for key in keys:
            t = gathered[key]
            if cache is None:
                res = torch.empty_like(t)
            else:
                ck = ('res', key, tuple(t.shape), t.dtype, str(t.device))
                res = cache.get(ck)
                if res is None:
                    for old_ in [c for c in cache if c[0] == 'res' and c[1] == key]:
                        del cache[old_]
                    res = cache[ck] = torch.empty_like(t)
            res[flat.to(t.device)] = t                       # every row is written: flat is a permutation (checked above)
            out[key] = res
        return out
    lo, hi = shard_range(nb, rank, world)
    loc = solve_fn(A[lo:hi], B[lo:hi], H[lo:hi], **{k: v[lo:hi] for k, v in (extra or {}).items()})
    return all_gather_results({k: loc[k] for k in keys}, nb, group, cache)
