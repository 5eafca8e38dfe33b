#!/bin/bash
# kernel trace of the isolated factorisation bench (scripts/factor_bench.py), one run per value of an environment switch: gpu_kt_factor.sh VAR v1 v2 ...
mkdir -p gpurun_out /tmp/prof; export TMPDIR=/tmp
VAR=${1:-TMPC_NONE}; shift
for V in "${@:-0}"; do
  export $VAR=$V
  rm -rf /tmp/prof/ktf
  rocprofv3 --kernel-trace -d /tmp/prof/ktf -o kt -- python3 scripts/factor_bench.py 433,64,300 > /tmp/prof/ktf.out 2> /tmp/prof/ktf.err
  echo "$VAR=$V"; tail -1 /tmp/prof/ktf.out
  python3 scripts/rocpd_stats.py $(find /tmp/prof/ktf -name '*.db' | head -1) 2>&1 | head -12 | cut -c1-150
done
