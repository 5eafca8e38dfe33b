#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "block_solve or parity_vs_oracle or golden" 2>&1 | grep -E "passed|failed|FAILED|Error" | head
for RR in 0 1; do echo "TRSM_RR=$RR"; TMPC_TRSM_RR=$RR timeout 300 python scripts/factor_bench.py 512,64,300 64,64,300; done
TMPC_TRSM_RR=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2t_bench.json 2> gpurun_out/r2t_bench.err
python - <<PY
import json
j=json.load(open('gpurun_out/r2t_bench.json'))
print(round(j['value'],1), 'ms/step', round(j['ms_per_step'],1), j['config']['ipm_iterations_max'], j['config']['status_optimal'], {k:round(v,1) for k,v in j['phase_ms'].items()}, {k:(round(v,2) if isinstance(v,float) else v) for k,v in j['roofline']['factorisation_phase'].items() if k!='kernels'}, 'upd TF', round(j['roofline']['achieved'],2))
PY
