#!/bin/bash
# config sweep of the in-tree library against another build on ONE box (TMPC_LIB), alternating
B=$1; shift
for r in 1 2; do
  echo "== base $B"; TMPC_LIB=$B python scripts/config_sweep.py "$@" 2>&1 | python -c "import sys,json; [print('  ',j['config'][:34], '%.0f/s  %.4f s  fac/iter %.3f ms'%(j['stage_conv_per_s'], j['seconds_host_to_host'], j['factor_ms_per_iter'])) for j in map(json.loads, (l for l in sys.stdin if l.startswith('{')))]"
  echo "== in-tree"; python scripts/config_sweep.py "$@" 2>&1 | python -c "import sys,json; [print('  ',j['config'][:34], '%.0f/s  %.4f s  fac/iter %.3f ms'%(j['stage_conv_per_s'], j['seconds_host_to_host'], j['factor_ms_per_iter'])) for j in map(json.loads, (l for l in sys.stdin if l.startswith('{')))]"
done
