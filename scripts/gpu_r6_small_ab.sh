#!/bin/bash
# round 6: A/B of two builds on the small shapes (single-tile blocks): bit-identity of the results (tests/tools/result_digest.py), then bench.small_configs with each
# usage: gpu_r6_small_ab.sh BASE.so   (file name under tunempc_amd/lib/)
export TMPDIR=/tmp; mkdir -p gpurun_out
BASE=$1
timeout 600 python tests/tools/result_digest.py 2>&1 | grep -v "^library" > gpurun_out/digest_new.txt
timeout 600 python tests/tools/result_digest.py tunempc_amd/lib/$BASE 2>&1 | grep -v "^library" > gpurun_out/digest_base.txt
if cmp -s gpurun_out/digest_new.txt gpurun_out/digest_base.txt; then echo "digests IDENTICAL ($(wc -l < gpurun_out/digest_new.txt) lines)"; else echo "digests DIFFER"; diff gpurun_out/digest_new.txt gpurun_out/digest_base.txt | head -20; fi
for lib in "" $BASE "" $BASE; do
TMPC_LIB_ALT=$lib timeout 600 python - <<'PY' 2>&1 | tail -n 6
import json, os, sys
import torch
sys.path.insert(0, '.')
import tunempc_amd._lib as L
alt = os.environ.get('TMPC_LIB_ALT')
if alt:
    path = os.path.abspath(os.path.join('tunempc_amd', 'lib', alt)); L.library_path = lambda: path
import bench
from tunempc_amd import synthetic
bench_cpu = bench.small_configs.__globals__
out = bench.small_configs(L.HipConvexifier, synthetic)
print('library', alt or 'in-tree')
for k, v in out.items():
    if isinstance(v, dict): print(f"  {k[:46]:46s} {v['gpu_ms_per_solve']:8.3f} ms  other path {v.get('other_path_ms_per_solve')}")
PY
done
