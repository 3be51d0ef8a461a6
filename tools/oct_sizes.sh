#!/bin/bash
# step time by batch size with the octahedral delta on either kernel (DSA_OCT_STREAMS=0|1): where the batch-size rule belongs
R=$PWD; O=$R/gpurun_out; mkdir -p $O
A="--steps 5 --warmup 2 --no-cpu-baseline --no-encode --no-end-to-end --no-pool --check 4 --scaling weak"
for n in 256 512 1024 2048 3072; do
  for v in 0 1; do
    DSA_OCT_STREAMS=$v python bench.py $A --meshes $n > $O/os_${n}_$v.json 2> $O/os_${n}_$v.err || { tail -3 $O/os_${n}_$v.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$O/os_${n}_$v.json").read().strip().splitlines()[-1])
print("meshes $n oct_streams $v ms_per_step %.2f" % d["ms_per_step"], d.get("oracle_check"))
PY
  done
done
