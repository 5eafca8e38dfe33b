mkdir -p gpurun_out
for i in 1 2 3; do
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "FAILED|passed|failed|Error|assert " | head -20
done > gpurun_out/r2j_dbg.log 2>&1
cat gpurun_out/r2j_dbg.log
