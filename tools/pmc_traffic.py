#!/usr/bin/env python3
"""profiles/rNN_traffic.json from two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <meshes> <triangles> <decodes in the run> > profiles/r02_traffic.json

Per kernel: mean counter value per dispatch.  FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3 derived
metrics: TCC_EA0_RDREQ / WRREQ based); on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Narrow gathers are uncalibrated there, so the doubled
value is an upper estimate for the gather-heavy kernels.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc, calls = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('dsa::', '').replace('void ', '')
        acc[k] += float(r['Counter_Value'])
        calls[k] += 1
    return {k: acc[k] / calls[k] for k in acc}, dict(calls)


fetch, calls = per_kernel(sys.argv[1], 'FETCH_SIZE')
write, _ = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {"meshes_per_gpu": int(sys.argv[3]), "triangles_per_mesh": int(sys.argv[4]),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on bench.py; KB -> bytes; FETCH_SIZE x2 (gfx950)",
       "kernels": {}}
for k in sorted(fetch):
    if not k.startswith('k_'):
        continue
    fb, wb = 2.0 * fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
    out["kernels"][k] = {"fetch_bytes_x2": fb, "write_bytes": wb, "hbm_bytes": fb + wb, "dispatches": calls[k]}
decodes = int(sys.argv[5]) if len(sys.argv) > 5 else 1
for v in out["kernels"].values():
    v["launches_per_decode"] = v["dispatches"] / decodes
out["total_hbm_bytes"] = sum(v["hbm_bytes"] * v["launches_per_decode"] for v in out["kernels"].values())
print(json.dumps(out, indent=1))
