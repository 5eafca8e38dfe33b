"""One-line digest of a bench.py JSON line."""
import json, sys
j = json.load(open(sys.argv[1]))
f = j['roofline'].get('factorisation_phase', {})
print(round(j['value'], 1), 'stage-conv/s  ms/step', round(j['ms_per_step'], 1), 'iters max', j['config'].get('ipm_iterations_max'), 'optimal', j['config'].get('status_optimal'),
      {k: round(v, 1) for k, v in j.get('phase_ms', {}).items()}, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in f.items() if k != 'kernels'},
      'update TF', round(j['roofline']['achieved'], 2))
