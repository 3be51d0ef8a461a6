#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
A="--steps 6 --warmup 2 --no-cpu-baseline --no-encode --no-end-to-end --no-pool --check 4 --scaling weak"
for n in 4096 2048; do
  for lib in default build_abl/lib_oct3.so build_abl/lib_oct0.so off; do
    if [ "$lib" = default ]; then unset DSA_LIB; unset DSA_OCT_STREAMS; elif [ "$lib" = off ]; then unset DSA_LIB; export DSA_OCT_STREAMS=0; else export DSA_LIB=$R/$lib; unset DSA_OCT_STREAMS; fi
    [ "$lib" != off ] && export DSA_OCT_STREAMS=1
    python bench.py $A --meshes $n > $O/op.json 2> $O/op.err || { tail -3 $O/op.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$O/op.json").read().strip().splitlines()[-1])
print("meshes $n lib $lib ms_per_step %.2f" % d["ms_per_step"], {k: round(v,2) for k,v in d["stage_ms"].items()})
PY
  done
done
