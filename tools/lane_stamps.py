"""Diagnostics: shader-clock shares of the segments of a k_symbols_lanes step (library built with -DLN_STAMPS,
loaded through DSA_LIB).  Usage: DSA_LIB=build_abl/lib_stamps.so DSA_LANES=1 DSA_SERIAL=1 python tools/lane_stamps.py [meshes]"""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
blob, offs = synth.make_batch(synth.GRID, 128, 256, 1000, n)
ctx = dsa.Context(0); ctx.set_profiling(True)
b = dsa.Batch(ctx, blob=blob, offsets=offs)
for _ in range(2): b.decode()
print({k: round(v, 2) for k, v in b.stage_times().items()})
for i in (0, 64, 128):
    d = b.debug_array(i, 4, np.uint32, 12)
    s1, s2, s3 = (int(d[10]) & 0xFFFF) << 12, (int(d[10]) >> 16) << 12, int(d[11]) << 12
    nv = 33153 * 3
    print("mesh %d position stream: cycles per symbol  top->LUT %.0f  LUT->cum %.0f  search+update+bytes %.0f" % (i, s1 / nv, s2 / nv, s3 / nv))
