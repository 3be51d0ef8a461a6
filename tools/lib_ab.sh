#!/bin/bash
# Same-box A/B of library builds on the bench workload: bash tools/lib_ab.sh build_abl/lib_a.so build_abl/lib_b.so ...
# ("default" = the in-tree library).  Extra environment for every run: AB_ENV="DSA_CHAIN=0".
mkdir -p gpurun_out
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = default ]; then unset DSA_LIB; else export DSA_LIB=$PWD/$lib; fi
  env $AB_ENV python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-encode --check 0 > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { tail -5 gpurun_out/ab_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$tag.json"))
print("$tag ms_per_step %.2f" % d["ms_per_step"], {k: round(v,2) for k,v in d["stage_ms"].items()})
PY
done
