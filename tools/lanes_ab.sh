#!/bin/bash
# A/B of the lane-per-chain kernels on the bench workload (diagnostics): DSA_LANES bit 0 = symbols, bit 1 = prediction
mkdir -p gpurun_out
for f in ${LANES_AB:-0 4}; do
  DSA_LANES=$f python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/ab_lanes_$f.json 2> gpurun_out/ab_lanes_$f.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_lanes_$f.json"))
print("DSA_LANES=$f ms_per_step %.2f" % d["ms_per_step"], {k: round(v,2) for k,v in d["stage_ms"].items()})
PY
done
