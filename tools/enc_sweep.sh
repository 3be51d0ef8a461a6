#!/bin/bash
# encode rate of the bench meshes over meshes-per-wave of the walks at several batch sizes: bash tools/enc_sweep.sh   (on the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
: > $O/enc_sweep.txt
for n2 in 256 512 1024 2048 4096; do
  for wl in 1 2 4 8 16; do
    echo "$n2 meshes, $wl to a wave: $(DSA_ENC_WALK_LANES=$wl timeout -k 10 120 python3 $R/tools/enc_once.py $n2 2>&1 | tail -1)" >> $O/enc_sweep.txt
  done
done
cat $O/enc_sweep.txt
