#!/bin/bash
# bench.py and tools/variant_timing.py with two builds of the library on one box: bash tools/ab_variants.sh libA.so libB.so
for lib in "$@"; do
  if [ "$lib" = default ]; then unset DSA_LIB; else export DSA_LIB=$PWD/$lib; fi
  echo "== $lib"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-encode --no-end-to-end --no-pool --check 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py ms_per_step %.2f' % d['ms_per_step'], {k: round(v,2) for k,v in d['stage_ms'].items()})"
  python tools/variant_timing.py 4096 2>&1 | cut -c1-230
done
