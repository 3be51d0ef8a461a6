#!/bin/bash
# kernel timeline of one warm encode chunk: bash tools/enc_trace.sh [meshes]   (on the GPU box; output under gpurun_out/)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; n=${1:-512}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/enc_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_trace -- python3 $R/tools/enc_once.py $n > $O/enc_trace.log 2>&1
python3 - <<PY > $O/enc_timeline.txt
import csv, glob
f = glob.glob('$O/enc_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('dsa::', '') for r in rows]
idx = max(i for i, nm in enumerate(names) if nm == 'k_enc_connectivity')
while idx > 0 and names[idx - 1].startswith('k_enc') and int(rows[idx]['Start_Timestamp']) - int(rows[idx - 1]['Start_Timestamp']) < 400e6 and names[idx - 1] != 'k_enc_pack': idx -= 1
t0 = int(rows[idx]['Start_Timestamp'])
for r, nm in zip(rows[idx:], names[idx:]):
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%-24s start %8.3f ms  end %8.3f ms  dur %8.3f ms  vgpr %s grid %s" % (nm, s / 1e6, e / 1e6, (e - s) / 1e6, r.get('VGPR_Count', '?'), r.get('Grid_Size', '?')))
PY
