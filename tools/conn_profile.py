"""Connectivity-loop profile of the bench meshes from a -DDSA_LOOP_PROFILE build (DSA_LIB=build_abl/lib_prof.so):
symbols retired by (C R)^k runs, scalar C / R-L-E symbols and their mean cost in shader clocks.  usage: python tools/conn_profile.py [meshes]"""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blob, offs = synth.make_batch(synth.GRID, 128, 256, 1000, n)
ctx = dsa.Context(0); ctx.set_profiling(True)
b = dsa.Batch(ctx, blob=blob, offsets=offs)
for _ in range(2): b.decode()
print({k: round(v, 2) for k, v in b.stage_times().items()})
d = np.array([b.debug_array(i, 4, np.uint32, 20) for i in range(0, n, max(1, n // 64))]).astype(np.int64)
m = np.median(d, axis=0).astype(np.int64)
print("connectivity: loop %d ticks, tail %d, ranks+flags %d, whole %d" % (m[0], m[1], m[4], m[13]))
print("  run symbols %d in %d runs; scalar C %d (mean %d ticks), scalar R/L/E %d (mean %d ticks), fetch mean %d" % (
    m[10] & 0xFFFFF, m[10] >> 20, m[11] & 0xFFFF, m[12], m[11] >> 16, m[18], m[19]))
print("traversal: %d ticks; runs %d covering %d faces, scalar steps %d, failed attempts %d" % (m[6], m[7], m[8], m[9], m[5]))
