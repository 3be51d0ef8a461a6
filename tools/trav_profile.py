"""Traversal-loop profile from a -DDSA_TRAV_PROFILE build (DSA_LIB=build_abl/lib_tprof.so): shader clocks by phase and attempt counts.
usage: python tools/trav_profile.py [meshes]"""
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
print("traversal: %d ticks; runs %d covering %d faces, scalar steps %d, failed attempts %d, fast attempts %d" % (m[6], m[7], m[8], m[9], m[5], m[3]))
print("  fast hits %d, from step history %d, arithmetic membership %d, elements handed over %d, dependent attempts %d, element loads %d" % (m[1], m[13], m[14], m[15], m[2], m[4]))
names = ("fast loads", "element inputs + seed", "dependent hops+loads", "membership + verdict", "retire+progressions")
for k, slot in enumerate((10, 11, 12, 18, 19)):
    print("  %-22s %9d ticks" % (names[k], m[slot] * 16))
print("  %-22s %9d ticks" % ("scalar step", m[0] * 16))
