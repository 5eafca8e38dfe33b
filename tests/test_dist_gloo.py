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


def test_shard_range_partitions():
    from tunempc_amd.dist import shard_range
    for nb in (1, 7, 512, 4096):
        for world in (1, 2, 4, 8):
            edges = [shard_range(nb, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == nb
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
