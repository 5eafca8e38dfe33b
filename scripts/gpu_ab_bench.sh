#!/bin/bash
# A/B of one environment switch on the bench line: gpu_ab_bench.sh VAR v1 v2 ...  (alternating, three rounds; --no-extra --no-cpu-baseline)
mkdir -p gpurun_out; export TMPDIR=/tmp
VAR=$1; shift
for R in 1 2 3; do
  for V in "$@"; do
    env $VAR=$V timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/abb_${V}_$R.json 2> gpurun_out/abb_${V}_$R.err
    echo "$VAR=$V round $R: $(python scripts/show_bench.py gpurun_out/abb_${V}_$R.json | head -n 1)"
  done
done
