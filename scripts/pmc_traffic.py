"""HBM traffic per launch of the dominant kernel from two rocprofv3 --pmc passes of the bench command (FETCH_SIZE in one, WRITE_SIZE in the
other; rocpd sqlite databases) -> profiles/r6_traffic.json, which bench.py reports as roofline.traffic while it is newer than the kernel sources.
Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB-like units of 1024 B... (rocprofv3
reports kilobytes), and FETCH_SIZE counts the 128-byte requests of wide coalesced reads as 64 B on gfx950: doubled.
usage: python scripts/pmc_traffic.py <fetch db dir> <write db dir> [kernel substring]"""
import glob, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(path, counter, kernel):
    db = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select dispatch_id, sum(value) from counters_collection where counter_name = ? and kernel_name like ? group by dispatch_id",
                            (counter, '%' + kernel + '%')))
    return sum(v for _, v in rows) / max(len(rows), 1), len(rows)


def _normalised(path):
    """source text without // comments and whitespace: comment edits do not make a measurement stale, code edits do"""
    import re
    txt = open(path).read()
    txt = re.sub(r'//[^\n]*', '', txt)
    return re.sub(r'\s+', '', txt).encode()


def sources_sha():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'tunempc_amd', 'csrc')
    for f in sorted(x for x in os.listdir(d) if x.endswith(('.h', '.hip'))):      # every kernel source, as bench.py's kernel_sources_sha
        h.update(_normalised(os.path.join(d, f)))
    return h.hexdigest()[:16]


def main():
    kernels = sys.argv[3:] if len(sys.argv) > 3 else ['k_cr_trsm_dma(', 'k_cr_update_dma(', 'k_cr_update_dma_f32(']
    out = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes) on "
                  "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra`; per-kernel summary in profiles/r6_final_pmc.txt",
        "fetch_correction": "x2: gfx950 FETCH_SIZE counts 128-byte requests as 64 B (MI355X_MICROARCH.md, HBM section)",
        "sources_sha": sources_sha(),
        "note": "average over all launches of a kernel (7 levels per factorisation phase, shrinking active sets)",
        "kernels": {},
    }
    for kernel in kernels:
        f, nf = per_launch(sys.argv[1], 'FETCH_SIZE', kernel)
        w, nw = per_launch(sys.argv[2], 'WRITE_SIZE', kernel)
        out["kernels"][kernel.rstrip('(')] = {"launches": nf, "fetch_size_kb_per_launch_raw": f, "write_size_kb_per_launch": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r6_traffic.json'), 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
