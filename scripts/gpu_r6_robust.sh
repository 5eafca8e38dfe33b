#!/bin/bash
# round 6: the robustness sweeps at HEAD (single-precision updates on by default), the wide-block shapes with the updates on and off, the small_configs legs of bench.py
export TMPDIR=/tmp; mkdir -p gpurun_out
ROBUST_OUT=r6_robustness_cond1_3_5.json timeout 900 python scripts/robustness_sweep.py 1,3,5 2>&1 | tail -n 3
ROBUST_OUT=r6_robustness_cond6_7_8.json timeout 900 python scripts/robustness_sweep.py 6,7,8 2>&1 | tail -n 3
ROBUST_WIDE=1 ROBUST_OUT=r6_robustness_wide_lowp.json timeout 1200 python scripts/robustness_sweep.py 1,3,5,7 2>&1 | tail -n 3
ROBUST_WIDE=1 ROBUST_LOWP=0 ROBUST_OUT=r6_robustness_wide_fp64.json timeout 1200 python scripts/robustness_sweep.py 1,3,5,7 2>&1 | tail -n 3
timeout 600 python scripts/robustness_models.py 2>&1 | tail -n 3
timeout 600 python - <<'PY' 2>&1 | tail -n 30
import json, sys, torch
sys.path.insert(0, '.')
import bench
from tunempc_amd._lib import HipConvexifier
from tunempc_amd import synthetic
out = bench.small_configs(HipConvexifier, synthetic)
for k, v in out.items():
    print(k, json.dumps(v)[:400] if isinstance(v, dict) else v)
json.dump(out, open('gpurun_out/r6_small_configs.json', 'w'), indent=1)
PY
