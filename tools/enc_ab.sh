#!/bin/bash
# the encode leg of bench.py with two builds of the library on one box: bash tools/enc_ab.sh libA.so libB.so [meshes]
n=${ENC_MESHES:-2048}
for lib in "$@"; do
  if [ "$lib" = default ]; then unset DSA_LIB; else export DSA_LIB=$PWD/$lib; fi
  python bench.py --steps 2 --warmup 1 --meshes 256 --no-cpu-baseline --no-end-to-end --no-pool --check 0 --encode-meshes $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', json.dumps(d.get('encode'))[:600])"
done
