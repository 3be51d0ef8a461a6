#!/bin/bash
# A/B on one box with timelines: bash tools/oct_ab.sh tag:lib[:ENV=V[:ENV=V]] ...   (lib "default" = the in-tree library)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-encode --no-end-to-end --no-pool --check 0 --scaling weak"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  IFS=: read -r tag lib e1 e2 e3 <<< "$spec"
  if [ "$lib" = default ]; then unset DSA_LIB; else export DSA_LIB=$R/$lib; fi
  for kv in $e1 $e2 $e3; do export $kv; done
  python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-encode --no-end-to-end --no-pool --check 8 > $O/oab_$tag.json 2> $O/oab_$tag.err || { tail -5 $O/oab_$tag.err; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/oab_${tag}_trace -- python $R/bench.py $ARGS > $O/oab_${tag}_trace.log 2>&1
  python $R/tools/trace_summary.py $O/oab_${tag}_trace > $O/oab_${tag}_timeline.txt
  rm -rf $O/oab_${tag}_trace
  for kv in $e1 $e2 $e3; do unset ${kv%%=*}; done
  python - <<PY
import json
d=json.loads(open("$O/oab_$tag.json").read().strip().splitlines()[-1])
print("$tag ms_per_step %.2f" % d["ms_per_step"], {k: round(v,2) for k,v in d["stage_ms"].items()}, d.get("oracle_check"))
PY
done
