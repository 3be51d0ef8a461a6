#!/bin/bash
# encode rate of the bench meshes over host threads per fan-out and batch sizes: bash tools/enc_sweep.sh   (on the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
: > $O/enc_sweep.txt
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) | v1 quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) | nproc $(nproc)" >> $O/enc_sweep.txt
for th in 8 16 32 default; do
  if [ $th = default ]; then unset DSA_HOST_THREADS; else export DSA_HOST_THREADS=$th; fi
  for rep in 1 2; do echo "host threads $th, 4096 meshes: $(timeout -k 10 120 python3 $R/tools/enc_once.py 4096 2>&1 | tail -1)" >> $O/enc_sweep.txt; done
done
unset DSA_HOST_THREADS
for n2 in 512 1024 2048 8192; do echo "defaults, $n2 meshes: $(timeout -k 10 120 python3 $R/tools/enc_once.py $n2 2>&1 | tail -1)" >> $O/enc_sweep.txt; done
cat $O/enc_sweep.txt
