mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r2j_dbg.log 2>&1
cat gpurun_out/r2j_dbg.log
