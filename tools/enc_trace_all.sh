#!/bin/bash
# every kernel of a warm multi-chunk encode batch, by start time: bash tools/enc_trace_all.sh [meshes]   (on the GPU box; output under gpurun_out/)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; n=${1:-4096}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/enc_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/enc_trace -- python3 $R/tools/enc_once.py $n > $O/enc_trace.log 2>&1
python3 - <<PY > $O/enc_timeline_all.txt
import csv, glob
f = glob.glob('$O/enc_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('dsa::', '') for r in rows]
# the warm call: the second half of the k_enc_table_clear launches
clears = [i for i, nm in enumerate(names) if nm == 'k_enc_table_clear']
idx = clears[len(clears) // 2]
t0 = int(rows[idx]['Start_Timestamp'])
for r, nm in zip(rows[idx:], names[idx:]):
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    if (e - s) > 0.5e6: print("%-24s q %-4s start %8.3f ms  end %8.3f ms  dur %8.3f ms" % (nm, r.get('Queue_Id', '?'), s / 1e6, e / 1e6, (e - s) / 1e6))
mc = glob.glob('$O/enc_trace/**/*memory_copy_trace.csv', recursive=True)
if mc:
    rows = list(csv.DictReader(open(mc[0]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
    for r in rows:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        if s >= 0 and (e - s) > 1e6: print("copy %-18s start %8.3f ms  end %8.3f ms  dur %8.3f ms" % (r.get('Direction', '?'), s / 1e6, e / 1e6, (e - s) / 1e6))
PY
