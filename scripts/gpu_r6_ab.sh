#!/bin/bash
# round 6 A/B on one box: the library of HEAD~ (TMPC_LIB=..._prev.so) against the in-tree build with the single-precision updates off / on
mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/r6_ab.txt; : > $OUT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
one() { # label, lib, extra args
  if [ -n "$2" ]; then export TMPC_LIB=$2; else unset TMPC_LIB; fi
  timeout 300 $B $3 > /tmp/ab.json 2> /tmp/ab.err < /dev/null
  python - "$1" >> $OUT <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/ab.json'))
    r = d['roofline']; f = r['factorisation_phase']
    print(f"{sys.argv[1]:34s} {d['value']:8.0f} stage-conv/s  {d['ms_per_step']:7.1f} ms/step  iters {d['config']['ipm_iterations_mean']:.2f}  optimal {d['config']['status_optimal']}  update {f['update_ms']:.2f} trsm {f['trsm_ms']:.2f} potrf {f['potrf_ms']:.2f} ms/phase  phase {f['tflops']:.1f} TF/s")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('/tmp/ab.err').read()[-300:])
PY
  tail -1 $OUT
}
PREV=$PWD/tunempc_amd/lib/libtunempc_hip_prev.so
for rep in 1 2; do
  one "prev (HEAD~)" $PREV ""
  one "new, lowp off" "" "--lowp-switch 0"
  one "new, lowp 3e-5" "" "--lowp-switch 3e-5"
  one "new, lowp 1e-5" "" "--lowp-switch 1e-5"
done
one "prev (HEAD~)" $PREV ""
