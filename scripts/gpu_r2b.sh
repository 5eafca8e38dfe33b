#!/bin/bash
# round 2, second GPU session: lanes (concurrent half-waves) A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r2b_tests.log
echo "tests rc=$?" >> gpurun_out/r2b_tests.log
for L in 1 2 4; do
  TMPC_LANES=$L timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2b_bench_lanes$L.json 2> gpurun_out/r2b_bench_lanes$L.err
done
tail -4 gpurun_out/r2b_tests.log
for L in 1 2 4; do python - <<PY
import json
j=json.load(open('gpurun_out/r2b_bench_lanes$L.json'))
print('lanes $L', round(j['value'],1), 'ms/step', round(j['ms_per_step'],1), j['config']['ipm_iterations_max'], j['config']['status_optimal'], {k:round(v,1) for k,v in j['phase_ms'].items()}, {k:(round(v,2) if isinstance(v,float) else v) for k,v in j['roofline']['factorisation_phase'].items() if k!='kernels'}, 'upd TF', round(j['roofline']['achieved'],2))
PY
done
