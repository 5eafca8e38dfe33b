#!/usr/bin/env python
"""Benchmark of the convexify() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1 directly; N>1 under torch.distributed.run)

A "step" is one complete batched convexify (pre-check, scaling, SDP Step 1 to tolerance, reconstruction,
status check) over one synthetic batch of `--batch` tuning problems per GPU, inputs resident in HBM.
Workload: BASELINE.json configs[3] = synthetic (nx+nu)=32 (nx=24, m=8), p=64; the published batch of 4096
is sharded over 8 GPUs, i.e. 512 problems per GPU (weak scaling: the per-GPU batch is fixed).  With N>1
ranks, each step ends with one RCCL all-gather of Hc/kappa/status.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X FP64 matrix peak (AMD CDNA4 datasheet figure; not in MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3  # fp32-in / fp32-accumulate MFMA (v_mfma_f32_16x16x4_f32; /opt/skills/guides/cdna_hip_programming.md: 155 TF measured)


def factor_flops_per_problem(p, d):
    """Algorithmic fp64 flops of one block-cyclic-tridiagonal Cholesky (k_factor), unpadded d."""
    d3 = float(d) ** 3
    if p == 1:
        return d3 / 3.0
    full = max(p - 2, 0) * (1.0 / 3 + 1 + 1 + 1 + 1 + 2) * d3     # chol, trsm O, trsm F, syrk O, syrk F, gemm F O'
    return full + (1.0 / 3 + 1 + 1) * d3 + d3 / 3.0                 # stage p-2 (no fill row), stage p-1 (chol only)


TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'r6_traffic.json')
KERNEL_SOURCES = sorted(os.path.join(ROOT, 'tunempc_amd', 'csrc', f) for f in os.listdir(os.path.join(ROOT, 'tunempc_amd', 'csrc')) if f.endswith(('.h', '.hip')))      # every kernel source (round 3 hashed two of them)


def _normalised(path):
    """source text without // comments and whitespace: comment edits do not make a measurement stale, code edits do"""
    import re
    txt = open(path).read()
    txt = re.sub(r'//[^\n]*', '', txt)
    return re.sub(r'\s+', '', txt).encode()


def kernel_sources_sha():
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(_normalised(f))
    return h.hexdigest()[:16]


def hbm_traffic_per_launch(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r6_traffic.json, written by scripts/pmc_traffic.py from separate
    rocprofv3 --pmc runs of this same command; PMC counters cannot be collected from inside this process).  None when the file is missing, has no
    entry for the kernel, or was measured on OTHER kernel sources (content hash): a stale number is not reported."""
    try:
        j = json.load(open(TRAFFIC_JSON))
        if j.get('sources_sha') != kernel_sources_sha():
            return None
        return float(j['kernels'][kernel]['hbm_bytes_per_launch'])
    except Exception:
        return None


def host_cpu_info():
    """What the host offers THIS process: the cores it may run on (affinity), the CPU-time quota of its cgroup, model and NUMA layout.
    `os.cpu_count()` alone counts the machine's cores, whatever the container was given."""
    info = {"os_cpu_count": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["affinity"] = info["os_cpu_count"]
    quota = None
    try:                                   # cgroup v2
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = float(q) / float(per)
    except Exception:
        try:                               # cgroup v1
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    info["cgroup_cpu_quota"] = quota
    model, sockets, numa = None, set(), None
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name') and model is None:
                model = ln.split(':', 1)[1].strip()
            if ln.startswith('physical id'):
                sockets.add(ln.split(':', 1)[1].strip())
        numa = len([d for d in os.listdir('/sys/devices/system/node') if d.startswith('node') and d[4:].isdigit()])
    except Exception:
        pass
    info.update(model=model, sockets=len(sockets) or None, numa_nodes=numa)
    eff = info["affinity"]
    if quota:
        eff = max(1, min(eff, int(quota + 0.5)))
    info["effective_cores"] = eff
    return info


def cpu_baseline(p, nx, mb, tol):
    """The C++/OpenMP restatement of the same structured algorithm (oracle/cpu_ipm: LAPACK/BLAS on the d x d blocks, one problem
    per OpenMP thread), timed on the host cores at the BENCH shape: one problem on one thread, a mid-point (8 threads), and one
    problem per core on all cores this process may use (affinity and cgroup quota, not os.cpu_count()); bounded to about a minute of
    wall time.  Same seeded generator as the GPU batch.  Not PICOS+MOSEK."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import cpu_ipm
    from tunempc_amd import synthetic
    hw = host_cpu_info()
    cores = max(1, min(hw["effective_cores"], cpu_ipm.max_threads(), 64))
    A, B, H = synthetic.gen_batch(100000, cores, p, nx, mb)

    def run(nt):
        t0 = time.perf_counter()
        o = cpu_ipm.convexify_batch(A[:nt], B[:nt], H[:nt], tol=tol, threads=nt)
        return time.perf_counter() - t0, o

    t1, o1 = run(1)
    points = {1: p / t1}
    mid = 8 if cores >= 16 else None
    if mid:
        tm, _ = run(mid)
        points[mid] = mid * p / tm
    if cores > 1:
        tc, oc = run(cores)
    else:
        tc, oc = t1, o1
    points[cores] = cores * p / tc
    eff = points[cores] / (cores * points[1])
    d = nx * (nx + 1) // 2
    ws_mb = 3 * p * ((d + 15) // 16 * 16) ** 2 * 8 / 1e6
    note = ""
    if eff < 0.5:
        note = (f"; parallel efficiency {eff:.2f} < 0.5: every thread streams its own {ws_mb:.0f} MB of Schur blocks per factorisation "
                f"({cores * ws_mb / 1e3:.1f} GB over {cores} threads against the shared L3 / memory channels)"
                + (f", and os.cpu_count() = {hw['os_cpu_count']} exceeds what this process may use" if hw['os_cpu_count'] > cores else ""))
    return {"value": points[cores], "unit": "stage-convexifications/s", "cores": cores, "kind": "port",
            "single_thread_value": points[1], "parallel_efficiency": eff,
            "by_threads": {str(k): v for k, v in sorted(points.items())},
            "host": hw,
            "sample": f"oracle/cpu_ipm (C++/OpenMP restatement of the structured IPM, OpenBLAS on the {d}-wide blocks) at the bench "
                      f"shape nx={nx}, m={mb}, p={p} on {hw['model']} ({hw['sockets']} socket(s), {hw['numa_nodes']} NUMA node(s); "
                      f"os.cpu_count {hw['os_cpu_count']}, affinity {hw['affinity']}, cgroup quota {hw['cgroup_cpu_quota']}): {cores} problems on "
                      f"{cores} threads in {tc:.1f} s ({int(oc['iters'].max())} IPM iterations max, {int((oc['status'] == 0).sum())}/{cores} Optimal); "
                      f"1 problem on 1 thread in {t1:.1f} s ({p / t1:.2f} stage-conv/s, {int(o1['iters'][0])} iterations)"
                      + (f"; {mid} problems on {mid} threads: {points[mid]:.1f} stage-conv/s" if mid else "") + note}


def small_configs(HipConvexifier, synthetic):
    """BASELINE configs[1] and configs[2] on the GPU (host-buffer entry, median of 7 solves after a warm-up) with oracle/cpu_ipm on the host cores beside
    them (at batch 1 the margin over a CPU is thin and nobody had measured it on the GPU box)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    out = {}
    try:
        import cpu_ipm
    except Exception:      # noqa: BLE001
        cpu_ipm = None
    hw = host_cpu_info()
    cores = max(1, min(hw["effective_cores"], 64))
    # (round 6: the reference's steady-state examples, p = 1 -- examples/convex_lqr.py:40-46 nx = 3, m = 1; examples/cstr/main.py nx = 4, m = 2;
    # examples/evaporation_process/main.py:110-111 nx = 2, m = 2 -- as synthetic problems of their shapes, batch 1)
    # (the LQR example is convex: its H passes the pre-check, as in the reference -- convexifier.py:83-85; the two economic steady states have an indefinite H:
    #  sigP = 3 makes the synthetic member of their shape indefinite too, so that the leg times an interior-point solve)
    for key, (seed, nb, p, nx, mb, sigP) in {"configs[1] unicycle-shaped p=30 n=5 batch=1": (200000, 1, 30, 4, 1, 1.0),
                                              "configs[2] evaporation-shaped p=50 n=4 batch=256": (200100, 256, 50, 2, 2, 1.0),
                                              "LQR-shaped p=1 n=4 batch=1 (convex: pre-check only)": (200200, 1, 1, 3, 1, 1.0),
                                              "CSTR-shaped p=1 n=6 batch=1": (200300, 1, 1, 4, 2, 3.0),
                                              "evaporation-shaped p=1 n=4 batch=1": (200400, 1, 1, 2, 2, 3.0)}.items():
        A, B, H = synthetic.gen_batch(seed, nb, p, nx, mb, sigP=sigP)
        h = HipConvexifier(p, nx, mb, chunk=nb)
        h.convexify_batch(A, B, H)
        h.profile()                          # (clears the counters)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); o = h.convexify_batch(A, B, H); ts.append(time.perf_counter() - t0)
        tg = float(np.median(ts))
        # which path the library took, as the library reports it (profile slot 15: problems solved by the one-launch kernel; members that were convex already never reach either)
        names = {True: "k_ipm_small: the whole interior-point loop in one launch, one workgroup per problem (tmpc_persist.h)", False: "launch sequence (per-phase kernels, stages spread over the CUs)"}
        took = h.profile()['persistent_problems'] > 0
        rec = {"gpu_ms_per_solve": 1e3 * tg, "gpu_stage_conv_per_s": nb * p / tg, "status_optimal": int((o['status'] == 0).sum()), "batch": nb,
               "ipm_iterations_mean": float(o['iters'].mean()), "path": names[took] if o['iters'].max() > 0 else "pre-check only: every member convex already (convexifier.py:83-85), no interior-point iteration"}
        # the other path, for the record (TMPC_TUNE_PERSISTENT 0 / 2): what the default rule chose against what it did not
        try:
            h.set_tuning(persistent=0 if took else 2)
            h.convexify_batch(A, B, H)
            h.profile()
            t2 = []
            for _ in range(5):
                t0 = time.perf_counter(); h.convexify_batch(A, B, H); t2.append(time.perf_counter() - t0)
            other = h.profile()['persistent_problems'] > 0
            if o['iters'].max() > 0 and other != took:
                rec["other_path_ms_per_solve"] = 1e3 * float(np.median(t2))
            else:
                rec["other_path_ms_per_solve"] = None      # the shape is not eligible for the other path (or nothing iterated): nothing to compare
        except Exception as e:      # noqa: BLE001
            rec["other_path_error"] = f"{type(e).__name__}: {e}"
        h.close()
        if cpu_ipm is not None:
            th = 1 if nb == 1 else cores
            cpu_ipm.convexify_batch(A[:1], B[:1], H[:1], threads=1)
            tc = []
            for _ in range(3 if nb == 1 else 1):          # (the batch-256 leg takes ~9 s per pass on 16 threads)
                t0 = time.perf_counter(); cpu_ipm.convexify_batch(A, B, H, threads=th); tc.append(time.perf_counter() - t0)
            tcm = float(np.median(tc))
            rec.update(cpu_ms_per_solve=1e3 * tcm, cpu_stage_conv_per_s=nb * p / tcm, cpu_threads=th, gpu_over_cpu=tcm / tg)
        out[key] = rec
    out["note"] = "host-buffer entry (H2D + D2H inside), median of 7; CPU: oracle/cpu_ipm (C++/OpenMP port, one problem per thread), median of 3 (batch 1) / one pass (batch 256), same seeds"
    return out


def large_block_configs(HipConvexifier, synthetic):
    """Stage blocks beyond the tuned kernels' 32 (generic per-stage kernels, register-staged block factorisation): the plain model at n = 48 and the Step 2 model
    at n = 48 with 24 + 24 rows of G_k / C_k per stage (the shape the round-3 review names).  Host-buffer entry, median of 3 solves after a warm-up."""
    out = {}
    rng = np.random.default_rng(7)
    for key, (seed, nb, p, nx, mb, ng, nc) in {"plain p=16 nx=40 n=48 batch=32": (200200, 32, 16, 40, 8, 0, 0),
                                                "step2 p=8 nx=36 n=48 rows=24+24 batch=16": (200300, 16, 8, 36, 12, 24, 24)}.items():
        A, B, H = synthetic.gen_batch(seed, nb, p, nx, mb)
        h = HipConvexifier(p, nx, mb, chunk=nb, ng=ng, nc=nc)
        if ng or nc:
            J = rng.standard_normal((nb, p, ng + nc, nx + mb)); ncnt = np.full((nb, p), nc, np.int32)
            run = lambda: h.convexify_step2_batch(A, B, H, J, ncnt, 1e-2)
        else:
            run = lambda: h.convexify_batch(A, B, H)
        run()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); o = run(); ts.append(time.perf_counter() - t0)
        tg = float(np.median(ts))
        out[key] = {"gpu_ms_per_solve": 1e3 * tg, "gpu_stage_conv_per_s": nb * p / tg, "status_optimal": int((o['status'] == 0).sum()), "batch": nb,
                    "ipm_iterations_mean": float(o['iters'].mean()), "schur_block": nx * (nx + 1) // 2 + (ng + nc + 2 if (ng or nc) else 0)}
        # roofline of the leg: one more solve with the phase events on -- the factorisation flops of the whole solve against the fp64 matrix peak, and what the peak
        # would allow for this batch (the time of the factorisation flops alone at 100 % of it)
        try:
            from tunempc_amd._lib import FLAG_PROFILE
            h.set_options(flags=FLAG_PROFILE)
            h.profile(); run(); pf = h.profile()
            dblk = out[key]["schur_block"]
            fl = pf['problem_factorisations'] * factor_flops_per_problem(p, dblk)
            out[key]["roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_F64_MFMA_TFLOPS,
                                    "achieved": fl / (pf['factor_ms'] * 1e-3) / 1e12 if pf['factor_ms'] > 0 else None,
                                    "frac": fl / (pf['factor_ms'] * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS if pf['factor_ms'] > 0 else None,
                                    "kernel": "block factorisation, all levels (register-staged k_cr_potrf / k_cr_trsm + k_cr_update_dma: blocks wider than 320)",
                                    "factorisations_per_problem": pf['problem_factorisations'] / nb, "factor_ms_per_solve": pf['factor_ms'],
                                    "whole_solve_tflops": fl / tg / 1e12,
                                    "stage_conv_per_s_at_peak": nb * p / (fl / (PEAK_F64_MFMA_TFLOPS * 1e12)),
                                    "note": "stage_conv_per_s_at_peak: the rate if the factorisation flops of this batch ran at 100 % of the datasheet peak and nothing else took time"}
        except Exception as e:      # noqa: BLE001
            out[key]["roofline_error"] = f"{type(e).__name__}: {e}"
        h.close()
        # the CPU port beside it, on a bounded sample of the same stage shape (same rows): one problem per thread at a shorter period (the cost of a solve is linear
        # in the period: ~12 s per problem at blocks of 820).  Round 6: oracle/cpu_ipm carries the models with rows (cpu_ipm_con.h).
        try:
            sys.path.insert(0, os.path.join(ROOT, 'oracle'))
            import cpu_ipm
            hw = host_cpu_info()
            cores = max(1, min(hw["effective_cores"], cpu_ipm.max_threads(), 64))
            ps = 4
            As, Bs, Hs = synthetic.gen_batch(seed + 50, cores, ps, nx, mb)
            if ng or nc:
                Js = np.random.default_rng(8).standard_normal((cores, ps, ng + nc, nx + mb))
                t0 = time.perf_counter(); oc = cpu_ipm.convexify_con_batch(As, Bs, Hs, Js, ng=ng, ncnt=np.full((cores, ps), nc, np.int32), rho=1e-2, threads=cores); tc = time.perf_counter() - t0
            else:
                t0 = time.perf_counter(); oc = cpu_ipm.convexify_batch(As, Bs, Hs, threads=cores); tc = time.perf_counter() - t0
            out[key].update(cpu_stage_conv_per_s=cores * ps / tc, cpu_threads=cores, gpu_over_cpu=(nb * p / tg) / (cores * ps / tc),
                            cpu_baseline={"value": cores * ps / tc, "unit": "stage-convexifications/s", "cores": cores, "kind": "port",
                                          "sample": f"oracle/cpu_ipm ({'Step 2 model, stage-local elimination' if (ng or nc) else 'plain model'}), {cores} problems of the same stage shape "
                                                    f"{'and rows ' if (ng or nc) else ''}with period {ps} on {cores} threads in {tc:.1f} s, {int((oc['status'] == 0).sum())}/{cores} Optimal"})
        except Exception as e:      # noqa: BLE001
            out[key]["cpu_error"] = f"{type(e).__name__}: {e}"
    out["note"] = "32 < n <= 64: csrc/tmpc_big.h (one thread per matrix entry, matrices in global memory) + the register-staged factorisation kernels; not the tuned path of the headline"
    return out


# Everything that libraries print on stdout while the bench runs (RCCL prints its version banner there at communicator creation) goes
# to stderr: stdout carries exactly one line, the JSON record.
REAL_STDOUT = os.dup(1)
os.dup2(2, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=512, help='problems per GPU per step')
    ap.add_argument('--p', type=int, default=64)
    ap.add_argument('--nx', type=int, default=24)
    ap.add_argument('--mb', type=int, default=8)
    ap.add_argument('--tol', type=float, default=0.0, help='0 = library default')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--lowp-switch', type=float, default=-1.0, help='TMPC_TUNE_LOWP_SWITCH of the handle (single-precision Schur-complement updates while mu / kappa > value; 0: off); < 0: library default')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help="collective backend; 'gloo' (results staged through the host) lets the multi-rank branch run where RCCL cannot")
    ap.add_argument('--same-device', action='store_true', help='every rank on cuda:0 (with --backend gloo: the N > 1 code path on a one-GPU box; not a scaling measurement)')
    ap.add_argument('--digest', action='store_true', help='add the sha256 of the gathered Hc / kappa / status of the last step to the line (tests)')
    ap.add_argument('--no-small', action='store_true', help='skip the small-configuration legs (configs[1] latency, configs[2] rate)')
    ap.add_argument('--no-tight', action='store_true', help='skip the tight-accuracy leg (one extra step, ~10x a default step)')
    ap.add_argument('--no-extra', action='store_true', help='skip the unprofiled and host-buffer legs after the timed region')
    ap.add_argument('--distinct', type=int, default=512, help='distinct synthetic problems generated per rank (tiled to --batch when smaller)')
    ap.add_argument('--balance', action='store_true', help='N > 1: deal the global batch to the ranks by a cost proxy (sbeta = max|eig H| / min|eig H| per problem, one '
                    'batched eigen-scan) instead of giving every rank the problems it generated: tunempc_amd.dist.convexify_batch_sharded(cost=...)')
    ap.add_argument('--stragglers', type=int, default=0, help='make the first K problems of rank 0 hard targets (cond Hhat = 10^4.5: ~3x the iterations): what --balance is for')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tunempc_amd._lib import HipConvexifier, FLAG_PROFILE
    from tunempc_amd import synthetic
    from tunempc_amd.dist import all_gather_results, convexify_batch_sharded

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_dist = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 ranks with python -m torch.distributed.run '
                         f'--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...')
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    host_coll = args.backend == 'gloo'            # gloo gathers host tensors
    cdev = torch.device('cpu') if host_coll else dev
    if use_dist:
        if host_coll:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    p, nx, mb, n = args.p, args.nx, args.mb, args.nx + args.mb
    nbl = args.batch
    # synthetic data: `distinct` different seeded problems per rank, tiled to the per-GPU batch
    nd = min(args.distinct, nbl)
    reps = (nbl + nd - 1) // nd
    tile = lambda x: torch.from_numpy(np.tile(x, (reps, 1, 1, 1))[:nbl].copy()).to(dev)

    def rank_problems(r):
        A_, B_, H_ = synthetic.gen_batch(100000 + r * 10000, nd, p, nx, mb)
        if r == 0:
            for b in range(min(args.stragglers, nd)):       # hard targets at the head of rank 0's share (a contiguous split leaves them all on one rank)
                A_[b], B_[b], H_[b] = synthetic.gen_problem(990000 + b, p, nx, mb, cond_exp=4.5)[:3]
        return A_, B_, H_

    balance = bool(args.balance and use_dist and world > 1)
    if balance:                                   # every rank holds the global batch (it could not deal it otherwise); it solves the share the assignment gives it
        parts = [rank_problems(r) for r in range(world)]
        gA, gB, gH = (torch.cat([tile(q[i]) for q in parts]) for i in range(3))
        dA, dB, dH = gA[rank * nbl:(rank + 1) * nbl], gB[rank * nbl:(rank + 1) * nbl], gH[rank * nbl:(rank + 1) * nbl]
    else:
        A, B, H = rank_problems(rank)
        dA, dB, dH = tile(A), tile(B), tile(H)
    h = HipConvexifier(p, nx, mb, chunk=(nbl if (args.same_device and nbl < 512) else 0), flags=FLAG_PROFILE)
    if args.tol > 0:
        h.set_options(tol=args.tol, flags=FLAG_PROFILE)
    if args.lowp_switch >= 0:
        h.set_tuning(lowp_switch=args.lowp_switch)
    out = None
    gather_cache = {}                      # the all-gather lands in the same buffers every step (tunempc_amd/dist.py)
    cost = None
    if balance:                            # cost proxy of every problem of the global batch: sbeta from one eigen-scan of the stage Hessians (tmpc_eig_scan_host)
        ev = h.eig_scan(gH.cpu().numpy())                   # [nb, p, 4]: min eig, max eig, min |eig|, max |eig| per stage
        cost = ev[:, :, 3].max(1) / ev[:, :, 2].min(1)

    def step():
        nonlocal out
        if balance:
            def solve_fn(a, b, hh):
                nonlocal out
                out = h.convexify_batch_device(a.contiguous(), b.contiguous(), hh.contiguous(), None)
                return {k: (out[k].cpu() if host_coll else out[k]) for k in ('Hc', 'kappa', 'status')}
            src = (gA.cpu(), gB.cpu(), gH.cpu()) if host_coll else (gA, gB, gH)
            g = convexify_batch_sharded(*src, lambda a, b, hh: solve_fn(a.to(dev), b.to(dev), hh.to(dev)), cost=cost, cache=gather_cache)
            return g
        out = h.convexify_batch_device(dA, dB, dH, out)
        if use_dist:
            g = all_gather_results({k: (out[k].cpu() if host_coll else out[k]) for k in ('Hc', 'kappa', 'status')}, nbl * world, cache=gather_cache)
            return g
        return out

    for _ in range(args.warmup):
        step()
    h.profile()   # reset accumulators
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tmax = torch.tensor([el], dtype=torch.float64, device=cdev)
    el_rank = [el]
    if use_dist:
        # every rank's own clock between the two barriers is the same wall interval; what differs is the time it spent SOLVING: the profile's wall time of the handle
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    el = float(tmax.item())
    prof = h.profile()
    status = out['status'].cpu().numpy(); iters = out['iters'].cpu().numpy(); kappa = out['kappa'].cpu().numpy(); info0 = out['info'].cpu().numpy()
    ok = int((status == 0).sum())
    # load balance across ranks: the slowest member of every rank's shard sets that rank's step time
    it_rank = torch.tensor([float(iters.max()), float(iters.mean()), float(prof['total_ms']) / max(args.steps, 1), float(status.size)], dtype=torch.float64, device=cdev)
    if use_dist:
        it_all = [torch.empty_like(it_rank) for _ in range(world)]
        dist.all_gather(it_all, it_rank)
        it_all = [t.cpu().tolist() for t in it_all]
    else:
        it_all = [it_rank.cpu().tolist()]

    # the same solve WITHOUT the profiling events (the product default), and through the host-buffer entry (H2D + D2H inside), N = 1 only
    extra_rates = {}
    if world == 1 and not args.no_extra:
        ne = max(1, min(args.steps, 3))
        h.set_options(tol=args.tol if args.tol > 0 else None, flags=0)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(ne):
            step()
        torch.cuda.synchronize()
        e0 = time.perf_counter() - t0
        extra_rates["unprofiled"] = {"value": nbl * p * ne / e0, "ms_per_step": 1e3 * e0 / ne, "steps": ne,
                                     "note": "same device-resident step without TMPC_FLAG_PROFILE (no hipEvent pairs around the launches): the product default"}
        hA, hB, hH = (x.cpu().numpy() for x in (dA, dB, dH))
        h.convexify_batch(hA[:8], hB[:8], hH[:8])
        nh = max(1, min(args.steps, 2))
        t0 = time.perf_counter()
        for _ in range(nh):
            oh = h.convexify_batch(hA, hB, hH)
        e1 = time.perf_counter() - t0
        extra_rates["host_buffer"] = {"value": nbl * p * nh / e1, "ms_per_step": 1e3 * e1 / nh, "steps": nh,
                                      "note": "tmpc_convexify_batch_host: pageable numpy buffers in, numpy buffers out (H2D of A, B, H and D2H of Hc, dHc, P, "
                                              "scalars inside the timed region); SURVEY.md 8d's host-buffer to host-buffer metric -- never `value`",
                                      "status_optimal": int((oh['status'] == 0).sum())}
        # the documented non-default option that trades the reproducible point for speed (include/tunempc_hip.h, TMPC_FLAG_FAST_EXIT)
        from tunempc_amd._lib import FLAG_FAST_EXIT
        h.set_options(tol=args.tol if args.tol > 0 else None, flags=FLAG_FAST_EXIT)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(ne):
            step()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t0
        st2 = out['status'].cpu().numpy(); it2 = out['iters'].cpu().numpy()
        extra_rates["fast_exit_option"] = {"value": nbl * p * ne / e2, "ms_per_step": 1e3 * e2 / ne, "steps": ne,
                                           "ipm_iterations_mean": float(it2.mean()), "status_optimal": int((st2 == 0).sum()),
                                           "note": "NOT the headline setting: TMPC_FLAG_FAST_EXIT stops every member after its first full centering step -- feasible, "
                                                   "kappa within the same gap, but not the converged central-path point (Hc ~1e-3..1e-2 off it, not reproducible to 1e-8)"}
        # BASELINE configs[1] (unicycle-shaped, batch 1: latency) and configs[2] (evaporation-shaped, batch 256: rate), each with the CPU port beside it
        if not args.no_small:
            try:
                extra_rates["small_configs"] = small_configs(HipConvexifier, synthetic)
            except Exception as e:      # noqa: BLE001
                extra_rates["small_configs"] = None
                extra_rates["small_configs_error"] = f"{type(e).__name__}: {e}"
            try:
                extra_rates["large_blocks"] = large_block_configs(HipConvexifier, synthetic)
            except Exception as e:      # noqa: BLE001
                extra_rates["large_blocks"] = None
                extra_rates["large_blocks_error"] = f"{type(e).__name__}: {e}"
        # the opt-in tight-accuracy mode (include/tunempc_hip.h: tmpc_set_tight): every member continued from its centred point to
        # mu_t = 2^-37 kappa with double-double block linear algebra + dd dual-Newton polish (VALU kernels, no matrix cores)
        if not args.no_tight:
            try:
                h.set_options(tol=args.tol if args.tol > 0 else None, flags=FLAG_PROFILE)
                h.set_tight(True)
                h.profile()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                e3 = time.perf_counter() - t0
                pt = h.profile()
                st3 = out['status'].cpu().numpy(); it3 = out['iters'].cpu().numpy(); k3 = out['kappa'].cpu().numpy(); in3 = out['info'].cpu().numpy()
                # the double-double factorisations of the step = everything the profile counts beyond the default solve's share (the main run above, per step)
                dflt_fac_ms = prof['factor_ms'] / max(args.steps, 1); dflt_nfac = prof['problem_factorisations'] / max(args.steps, 1)
                dd_ms = max(pt['factor_ms'] - dflt_fac_ms, 1e-9); dd_nfac = max(pt['problem_factorisations'] - dflt_nfac, 0.0)
                dd_tf = dd_nfac * factor_flops_per_problem(p, nx * (nx + 1) // 2) / (dd_ms * 1e-3) / 1e12
                tight_cpu = None
                if not args.no_cpu_baseline:
                    try:                       # the CPU port in ITS tight mode, bounded sample: one problem per thread at period 8 instead of 64 (cost linear in the period)
                        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
                        import cpu_ipm
                        hw = host_cpu_info()
                        cores = max(1, min(hw["effective_cores"], cpu_ipm.max_threads(), 64))
                        ps = 8
                        As, Bs, Hs = synthetic.gen_batch(100000, cores, ps, nx, mb)
                        tc0 = time.perf_counter(); oc = cpu_ipm.convexify_batch(As, Bs, Hs, tol=2.0 ** -37, threads=cores, tight=True); tc = time.perf_counter() - tc0
                        tight_cpu = {"value": cores * ps / tc, "unit": "stage-convexifications/s", "cores": cores, "kind": "port",
                                     "sample": f"oracle/cpu_ipm tight mode (double-double block algebra + polish), {cores} problems of the bench stage shape with period {ps} "
                                               f"on {cores} threads in {tc:.1f} s, {int((oc['status'] == 0).sum())}/{cores} Optimal"}
                    except Exception as e:      # noqa: BLE001
                        tight_cpu = {"error": f"{type(e).__name__}: {e}"}
                extra_rates["tight_mode"] = {"value": nbl * p / e3, "ms_per_step": 1e3 * e3, "steps": 1, "tight_tol": 2.0 ** -37,
                                             "fell_back_to_default": int((in3[:, 10] == 4.0).sum()),
                                             "roofline": {"bound": "fp64 vector ALU (no extended format on the matrix cores)", "achieved": dd_tf, "peak": PEAK_F64_MFMA_TFLOPS / 12.0,
                                                          "unit": "dd-TFLOP/s (a double-double multiply-add = 2 flops = 12 fp64 vector operations with deferred renormalisation)",
                                                          "frac": dd_tf / (PEAK_F64_MFMA_TFLOPS / 12.0), "kernel": "k_dd_potrf + k_dd_trsm + k_dd_update, all levels",
                                                          "dd_factorisations_per_problem": dd_nfac / nbl, "dd_factor_ms": dd_ms},
                                             "cpu_baseline": tight_cpu,
                                             "ipm_iterations_mean": float(it3.mean()), "ipm_iterations_max": int(it3.max()), "status_optimal": int((st3 == 0).sum()),
                                             "mu_target_max": float(in3[:, 6].max()), "certified_gap_on_kappa": float((2 * p * n + 1) * (in3[:, 6] / k3).max()),
                                             "kappa_mean": float(k3.mean()), "kappa_drop_vs_default_mean": float((kappa - k3).mean()),
                                             "note": "NOT the headline setting: opt-in tmpc_set_tight, default solve + continuation to mu_t = 2^-37 kappa (relative gap on kappa "
                                                     "N * 7.3e-12 instead of N * 3e-8) in double-double arithmetic on the vector ALU (tmpc_dd.h)"}
                h.set_tight(False)
                # round 5: the mode on the Step 2 model (rows of G, ragged rows of C, norm terms) at the bench stage shape -- a small batch through the host-buffer entry
                try:
                    nb2, ng2, nc2, rho2 = 64, 2, 3, 1e-2
                    rng2 = np.random.default_rng(5)
                    A2, B2, H2 = synthetic.gen_batch(100000, nb2, p, nx, mb)
                    J2 = rng2.standard_normal((nb2, p, ng2 + nc2, n)); cnt2 = rng2.integers(0, nc2 + 1, size=(nb2, p)).astype(np.int32)
                    for b_ in range(nb2):
                        for k_ in range(p):
                            J2[b_, k_, ng2 + cnt2[b_, k_]:] = 0.0
                    h2 = HipConvexifier(p, nx, mb, ng=ng2, nc=nc2, chunk=nb2)
                    h2.convexify_step2_batch(A2[:4], B2[:4], H2[:4], J2[:4], cnt2[:4], rho2)
                    t0 = time.perf_counter(); od = h2.convexify_step2_batch(A2, B2, H2, J2, cnt2, rho2); ed = time.perf_counter() - t0
                    h2.set_tight(True)
                    h2.convexify_step2_batch(A2[:4], B2[:4], H2[:4], J2[:4], cnt2[:4], rho2)
                    t0 = time.perf_counter(); ot = h2.convexify_step2_batch(A2, B2, H2, J2, cnt2, rho2); et = time.perf_counter() - t0
                    h2.close()
                    step2_cpu = None
                    if not args.no_cpu_baseline:
                        try:                   # the CPU port on the same model (default mode and its tight mode), bounded sample: one problem per thread at period 8
                            sys.path.insert(0, os.path.join(ROOT, 'oracle'))
                            import cpu_ipm
                            hw = host_cpu_info()
                            cores = max(1, min(hw["effective_cores"], cpu_ipm.max_threads(), 64))
                            ps = 8
                            As, Bs, Hs = synthetic.gen_batch(100000, cores, ps, nx, mb)
                            Js = J2[:cores, :ps] if cores <= nb2 else np.tile(J2[:, :ps], ((cores + nb2 - 1) // nb2, 1, 1, 1))[:cores]
                            cs = cnt2[:cores, :ps] if cores <= nb2 else np.tile(cnt2[:, :ps], ((cores + nb2 - 1) // nb2, 1))[:cores]
                            tc0 = time.perf_counter(); oc = cpu_ipm.convexify_con_batch(As, Bs, Hs, Js, ng=ng2, ncnt=cs, rho=rho2, threads=cores); tcd = time.perf_counter() - tc0
                            step2_cpu = {"value": cores * ps / tcd, "unit": "stage-convexifications/s", "cores": cores, "kind": "port", "mode": "default",
                                         "sample": f"oracle/cpu_ipm, Step 2 model (stage-local elimination, cpu_ipm_con.h), {cores} problems of the bench stage shape and rows with period {ps} "
                                                   f"on {cores} threads in {tcd:.1f} s, {int((oc['status'] == 0).sum())}/{cores} Optimal"}
                            tc0 = time.perf_counter(); oc = cpu_ipm.convexify_con_batch(As, Bs, Hs, Js, ng=ng2, ncnt=cs, rho=rho2, tol=2.0 ** -37, threads=cores, tight=True); tct = time.perf_counter() - tc0
                            step2_cpu["tight"] = {"value": cores * ps / tct, "unit": "stage-convexifications/s", "cores": cores, "kind": "port",
                                                  "sample": f"the same sample in the port's tight mode (double-double rows and polish, cpu_ipm_con.h) in {tct:.1f} s, {int((oc['status'] == 0).sum())}/{cores} Optimal"}
                        except Exception as e:      # noqa: BLE001
                            step2_cpu = {"error": f"{type(e).__name__}: {e}"}
                    # cone dimension: the 2 p LMI blocks and alpha, one linear cone per row, an (m + 1)-dimensional arrow block per norm term (rows of G; rows of C where present)
                    ncone = 2 * p * n + 1 + (ng2 + cnt2).sum(1) + p * (ng2 + 1) + (cnt2 + (cnt2 > 0)).sum(1)
                    extra_rates["tight_mode"]["step2_model"] = {
                        "value": nb2 * p / et, "default_value": nb2 * p / ed, "batch": nb2, "rows_G": ng2, "rows_C_max": nc2, "rho": rho2,
                        "ipm_iterations_mean": float(ot['iters'].mean()), "default_ipm_iterations_mean": float(od['iters'].mean()),
                        "status_optimal": int((ot['status'] == 0).sum()), "fell_back_to_default": int((ot['info'][:, 10] == 4.0).sum()),
                        "certified_gap_on_value": float((ncone * ot['info'][:, 6] / ot['kappa']).max()),
                        "cpu_baseline": step2_cpu,
                        "note": "Step 2 model (convexifier.py:116-131) in the tight mode, host-buffer entry (H2D + D2H inside), one handle, one timed call each"}
                except Exception as e:      # noqa: BLE001
                    extra_rates["tight_mode"]["step2_model"] = {"error": f"{type(e).__name__}: {e}"}
            except Exception as e:      # noqa: BLE001
                extra_rates["tight_mode"] = None
                extra_rates["tight_mode_error"] = f"{type(e).__name__}: {e}"
        h.set_options(tol=args.tol if args.tol > 0 else None, flags=FLAG_PROFILE)

    if rank == 0:
        total_units = nbl * world * p * args.steps
        d = nx * (nx + 1) // 2
        d3 = float(d) ** 3
        from tunempc_amd._lib import cr_schedule
        sched = cr_schedule(p)
        upd_levels = int((sched['levels'][:, 3] > 0).sum())              # levels with a k_cr_update launch
        phases = max(prof['factor_launches'], 1.0)                       # factorisation phases (one per IPM iteration and chunk)
        nfac = max(prof['problem_factorisations'], 1.0)                  # problem-factorisations: only problems still iterating are factored
        # the four kernels of the factorisation phase, each against the matrix peak of the arithmetic it runs in; the dominant one (largest share of the timed region)
        # is the `roofline` of the line.  Algorithmic flops per problem-factorisation (unpadded d): block Cholesky p d^3 / 3; triangular solves d^3 per edge,
        # 2 (p - 2) + 1 edges; updates 4 d^3 per elimination with two neighbours, d^3 for the last pair -- in float32 for the problem-factorisations of the early
        # main-phase iterations (TMPC_TUNE_LOWP_SWITCH), in fp64 for the others.
        nlow = float(prof.get('lowp_factorisations', 0.0))
        f_upd = (max(p - 2, 0) * 4.0 + (1.0 if p >= 2 else 0.0)) * d3
        f_trsm = (max(p - 2, 0) * 2.0 + (1.0 if p >= 2 else 0.0)) * d3
        kern = {
            "k_cr_trsm_dma": dict(flops=nfac * f_trsm, ms=prof['trsm_ms'], peak=PEAK_F64_MFMA_TFLOPS, dtype="f64",
                                  what="triangular solves O <- E L^-T of the cyclic-reduction block Cholesky, register-resident right-looking strips on v_mfma_f64_4x4x4_4b"),
            "k_cr_update_dma": dict(flops=(nfac - nlow) * f_upd, ms=prof['update_ms'], peak=PEAK_F64_MFMA_TFLOPS, dtype="f64",
                                    what="symmetric updates and fill edges in fp64: v_mfma_f64_4x4x4_4b on LDS-DMA fed 64 x 64 tiles"),
            "k_cr_update_dma_f32": dict(flops=nlow * f_upd, ms=prof.get('update_f32_ms', 0.0), peak=PEAK_F32_MFMA_TFLOPS, dtype="f32",
                                        what="the same updates on float32 copies of the O blocks with float32 accumulation (v_mfma_f32_16x16x4_f32), early main-phase iterations only"),
            "k_cr_potrf_dma": dict(flops=nfac * p * d3 / 3.0, ms=prof['potrf_ms'], peak=PEAK_F64_MFMA_TFLOPS, dtype="f64",
                                   what="block Cholesky of the diagonal blocks (latency-bound 64 x 64 tile factorisations between MFMA sweeps)"),
        }
        for k_, v_ in kern.items():
            v_["tflops"] = v_["flops"] / (v_["ms"] * 1e-3) / 1e12 if v_["ms"] > 0 else 0.0
            v_["frac"] = v_["tflops"] / v_["peak"]
        dom = max(kern, key=lambda k_: kern[k_]["ms"])
        dom_launches = phases * max(upd_levels, 1) if dom != "k_cr_potrf_dma" else phases * len(sched['levels'])
        achieved = kern[dom]["tflops"]
        upd_ms = prof['update_ms'] + prof.get('update_f32_ms', 0.0)
        phase_tf = nfac * factor_flops_per_problem(p, d) / (prof['factor_ms'] * 1e-3) / 1e12 if prof['factor_ms'] > 0 else 0.0
        line = {
            "metric": "stage-Hessian convexifications/sec at (nx+nu)=32, p=64",
            "value": total_units / el,
            "unit": "stage-convexifications/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: synthetic SPD-perturbed Hessians, (nx+nu)={n} (nx={nx}, m={mb}), p={p}, "
                                   f"{nbl} problems per GPU ({nbl * world} total; published batch 4096 = 512 x 8 GPUs)",
                       "p": p, "nx": nx, "m": mb, "batch_per_gpu": nbl, "global_batch": nbl * world,
                       "distinct_problems_per_gpu": nd, "mu_tol": args.tol if args.tol > 0 else 2.0 ** -25,
                       "certified_gap_on_kappa": float((2 * p * n + 1) * (info0[:, 6] / np.maximum(kappa, 1e-300)).max()),      # N mu_t / kappa: what mu_tol means for kappa (the reference solvers stop at ~1e-8: `tight_mode`)
                       "ipm_iterations_max": int(iters.max()), "ipm_iterations_mean": float(iters.mean()),
                       "ipm_iterations_per_rank": [{"max": int(a), "mean": b} for a, b, _, _ in it_all],
                       "solve_ms_per_step_per_rank": [c for _, _, c, _ in it_all],      # time inside the library per step (the rest of a rank's step is waiting for the slowest at the gather)
                       "problems_per_rank": [int(d_) for _, _, _, d_ in it_all],
                       "balance": ("cost-balanced shards (sbeta proxy, tunempc_amd.dist.balanced_assignment)" if balance else "every rank solves the problems it generated"),
                       "stragglers": int(args.stragglers),
                       "status_optimal": ok, "status_total": int(status.size),
                       "kappa_mean": float(kappa.mean()),
                       "parallelism": (f"batch-sharded x{world}, one all-gather of Hc" + (f" [backend {args.backend}" + (", every rank on cuda:0: the code path, not a scaling measurement]" if args.same_device else "]") if (host_coll or args.same_device) else "")) if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": kern[dom]["peak"], "unit": "TFLOP/s",
                         "frac": achieved / kern[dom]["peak"],
                         "traffic": hbm_traffic_per_launch(dom) if (nbl == 512 and p == 64 and nx == 24 and mb == 8) else None,
                         "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r6_traffic.json)",
                         "kernel": f"{dom} ({kern[dom]['what']})", "dtype": kern[dom]["dtype"],
                         "dominant_by": "largest total time among the kernels of the factorisation phase in the timed region (HIP events around every launch)",
                         "avg_launch_ms": kern[dom]["ms"] / max(dom_launches, 1), "launches": int(dom_launches),
                         "algorithmic_flops_per_launch": kern[dom]["flops"] / max(dom_launches, 1),
                         "kernels": {k_: {"ms_per_step": v_["ms"] / max(args.steps, 1), "tflops": v_["tflops"], "peak": v_["peak"], "frac": v_["frac"], "dtype": v_["dtype"]} for k_, v_ in kern.items()},
                         "single_precision_updates": {"switch_mu_over_kappa": (args.lowp_switch if args.lowp_switch >= 0 else 1e-5), "problem_factorisations": nlow / max(args.steps, 1),
                                                      "of": nfac / max(args.steps, 1),
                                                      "note": "TMPC_TUNE_LOWP_SWITCH: Schur-complement updates (and the O-block reads of the substitutions) of the main-phase iterations with mu > switch * kappa "
                                                              "in float32; Cholesky, triangular solves, every later iteration and the returned point stay fp64 (same iteration counts, Hc within 2e-10 of the "
                                                              "all-fp64 answer: profiles/r6_fp32_*.txt, tests)"},
                         "factorisation_phase": {"kernels": "k_cr_potrf_dma + k_cr_trsm_dma + k_cr_update_dma + k_cr_update_dma_f32 (+ the fused forward sweep of the predictor pass), all levels", "tflops": phase_tf,
                                                 "frac": phase_tf / PEAK_F64_MFMA_TFLOPS, "avg_ms": prof['factor_ms'] / phases,
                                                 "problems_per_phase": nfac / phases,
                                                 "potrf_ms": prof['potrf_ms'] / phases, "trsm_ms": prof['trsm_ms'] / phases,
                                                 "update_ms": upd_ms / phases, "update_f64_ms": prof['update_ms'] / phases, "update_f32_ms": prof.get('update_f32_ms', 0.0) / phases,
                                                 "note": "tflops / frac: algorithmic flops of the whole phase against the FP64 matrix peak (part of them run in float32: a mixed-precision phase has no single roof)"},
                         "peak_note": "datasheet matrix peaks at 2.4 GHz (FP64 78.6, FP32 157.3 TFLOP/s).  Under the fp64 kernels the chip runs at 2.28 GHz (GRBM_GUI_ACTIVE / wall time) with the "
                                      "matrix pipes busy 60 % (k_cr_trsm_dma) / 72 % (k_cr_update_dma) of the SIMD cycles (profiles/r5_final_pmc.txt, r6_final_pmc.txt): they are MFMA-idle a third "
                                      "of the time, not power-capped"},
            "phase_ms": {k: prof[k] for k in ('pre_ms', 'schur_ms', 'factor_ms', 'pass1_ms', 'pass2_ms', 'total_ms')},
        }
        line.update(extra_rates)
        if args.digest:
            import hashlib
            hsh = hashlib.sha256()
            for k in ('Hc', 'kappa', 'status'):
                hsh.update(np.ascontiguousarray(res[k].cpu().numpy()).tobytes())
            line["gathered_digest"] = hsh.hexdigest()
            line["gathered_problems"] = int(res['Hc'].shape[0])
        if not args.no_cpu_baseline and world == 1:
            try:                                                        # the checker's build or run must never cost the GPU measurement its line
                line["cpu_baseline"] = cpu_baseline(p, nx, mb, args.tol if args.tol > 0 else 2.0 ** -25)
            except Exception as e:      # noqa: BLE001
                line["cpu_baseline"] = None
                line["cpu_baseline_error"] = f"{type(e).__name__}: {e}"
        else:
            line["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(REAL_STDOUT, (json.dumps(line) + '\n').encode())      # the ONE line on stdout (libraries' banners went to stderr)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
