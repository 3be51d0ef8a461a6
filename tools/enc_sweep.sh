#!/bin/bash
# encode rate of the bench meshes over hardware queues, lanes and chunk sizes: bash tools/enc_sweep.sh   (on the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
: > $O/enc_sweep.txt
for cfg in "4 4 1024" "8 4 1024" "8 8 512" "8 6 704" "12 8 512" "16 12 352" "16 8 512"; do
  set -- $cfg
  for rep in 1 2; do
    echo "hw queues $1, lanes $2, chunk $3: $(GPU_MAX_HW_QUEUES=$1 DSA_ENC_LANES=$2 DSA_ENC_CHUNK=$3 timeout -k 10 120 python3 $R/tools/enc_once.py 4096 2>&1 | tail -1)" >> $O/enc_sweep.txt
  done
done
cat $O/enc_sweep.txt
