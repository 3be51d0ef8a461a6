#!/bin/bash
# Everything profiles/ holds for one round, from one box (run on the GPU box through gpurun):
#   bash tools/profile_round.sh <tag>      ->  gpurun_out/<tag>_{bench.json,kernel_stats.csv,timeline.txt,sq_counters.txt,traffic.json}
# rocprofv3 is given the program itself (python bench.py ...), counters in passes of their own.
set -e
tag=${1:-r04}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-encode --no-end-to-end --no-pool --no-sustained --no-dialects --check 0 --scaling weak"
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_trace -- python $R/bench.py $ARGS > $O/${tag}_trace.log 2>&1
echo "trace done"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/${tag}_sq -- python $R/bench.py $ARGS > $O/${tag}_sq.log 2>&1
echo "sq done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -- python $R/bench.py $ARGS > $O/${tag}_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -- python $R/bench.py $ARGS > $O/${tag}_write.log 2>&1
echo "write done"
cd $R
python tools/trace_summary.py $O/${tag}_trace > $O/${tag}_timeline.txt
cp $(ls $O/${tag}_trace/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
python tools/pmc_summary.py $O/${tag}_sq > $O/${tag}_sq_counters.txt
python tools/pmc_traffic.py $O/${tag}_fetch $O/${tag}_write 4096 65536 4 > $O/${tag}_traffic.json
rm -rf $O/${tag}_trace $O/${tag}_sq $O/${tag}_fetch $O/${tag}_write
tail -c 600 $O/${tag}_bench.json
