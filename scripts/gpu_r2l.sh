#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for C in 0 30 10; do
  TMPC_CHORD=$C timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2l_bench_chord$C.json 2> gpurun_out/r2l_bench_chord$C.err
python - <<PY
import json
j=json.load(open('gpurun_out/r2l_bench_chord$C.json'))
print('chord $C', round(j['value'],1), 'ms/step', round(j['ms_per_step'],1), j['config']['ipm_iterations_max'], round(j['config']['ipm_iterations_mean'],2), j['config']['status_optimal'], {k:round(v,1) for k,v in j['phase_ms'].items()}, {k:(round(v,2) if isinstance(v,float) else v) for k,v in j['roofline']['factorisation_phase'].items() if k!='kernels'})
PY
done
