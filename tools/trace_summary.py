import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last decode: take the last k_locate and everything after
idx = max(i for i, r in enumerate(rows) if r['Kernel_Name'].split('(')[0].replace('dsa::', '') == 'k_locate')
t0 = int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    name = r['Kernel_Name'].split('(')[0].replace('dsa::', '')
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%-28s start %8.3f ms  end %8.3f ms  dur %8.3f ms  vgpr %s lds %s" % (name, s / 1e6, e / 1e6, (e - s) / 1e6, r.get('VGPR_Count', '?'), r.get('LDS_Block_Size', '?')))
