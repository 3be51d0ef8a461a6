"""The bench batch re-encoded with valence Edgebreaker symbols (predictive_connectivity=2, what stock encoders write for larger
meshes): step time on the fast kernels, every mesh compared... no: a sample compared with the oracle.  usage: python tools/valence_timing.py [meshes]"""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import time
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth, oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blob, offs = synth.make_batch(synth.GRID, 128, 256, 1000, n, opt=synth.options(predictive_connectivity=2))
ctx = dsa.Context(0); ctx.set_profiling(True)
b = dsa.Batch(ctx, blob=blob, offsets=offs)
for _ in range(2): b.decode()
t0 = time.perf_counter()
for _ in range(3): b.decode()
dt = (time.perf_counter() - t0) / 3
bad = sum(1 for i in range(n) if b.status(i) != 0)
print("valence batch: %d meshes, %.1f ms per step = %.0f meshes/s; failed %d; stages" % (n, dt * 1e3, n / dt, bad), {k: round(v, 2) for k, v in b.stage_times().items()})
info = b.mesh_info(0)
print("mesh 0: status %d detail %d faces %d" % (info.status, info.detail, info.num_faces))
for i in sorted(set(int(x) for x in np.linspace(0, n - 1, 8))):
    ref = oracle.decode(bytes(blob[int(offs[i]):int(offs[i + 1])]))
    m = b.result(i).ConnectedData
    ok = np.array_equal(m.Faces, ref.faces) and all(np.array_equal(a.PortableValues, r.portable) and np.array_equal(a.PointMap, r.point_map) and a.Values.tobytes() == r.values.tobytes() for a, r in zip(m.Attributes, ref.attributes))
    print("mesh %d traversal type %d equal to the oracle: %s" % (i, ref.traversal_type, ok))
d = np.array([b.debug_array(i, 4, np.uint32, 20) for i in range(0, n, max(1, n // 32))]).astype(np.int64)
m = np.median(d, axis=0).astype(np.int64)
print("connectivity ticks %d (whole %d), traversal ticks %d" % (m[0], m[13], m[6]))
