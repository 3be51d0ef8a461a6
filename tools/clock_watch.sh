#!/bin/bash
# Sample the GPU's clocks / power while bench.py decodes (diagnostic; read-only queries).
# usage (on the GPU box): bash tools/clock_watch.sh > gpurun_out/clock_watch.txt
python bench.py --steps 60 --warmup 2 --no-cpu-baseline --no-encode --check 0 > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
sleep 1
echo "== idle-ish (bench starting)"; rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -v "^$" | head -40
for i in $(seq 1 60); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "== t=$i"; rocm-smi --showclocks --showpower -u 2>&1 | grep -E "sclk|mclk|fclk|Power|use" | head -12
  cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -8
  sleep 1
done
wait $BP
echo "== bench"; cat gpurun_out/clock_bench.json
amd-smi metric -c -p -u 2>&1 | head -60
