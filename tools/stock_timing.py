"""The bench batch as a stock encoder writes it at its default level, as far as the fast kernels go: valence Edgebreaker symbols,
parallelogram positions, GeometricNormal normals (UVs stay on the parallelogram: TexCoordsPortable needs the general path).
Step time and which kernels decoded the meshes; a sample compared with the oracle.
usage: python tools/stock_timing.py [meshes] [with_uv 0|1]"""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import time
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth, oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for label, opt in (("bench dialect", synth.options()), ("valence", synth.options(predictive_connectivity=2)), ("geometric normals", synth.options(normal_prediction=6)),
                   ("valence + geometric normals", synth.options(predictive_connectivity=2, normal_prediction=6))):
    blob, offs = synth.make_batch(synth.GRID, 128, 256, 1000, n, opt=opt)
    ctx = dsa.Context(0); ctx.set_profiling(True)
    b = dsa.Batch(ctx, blob=blob, offsets=offs)
    for _ in range(2): b.decode()
    t0 = time.perf_counter()
    for _ in range(3): b.decode()
    dt = (time.perf_counter() - t0) / 3
    bad = sum(1 for i in range(n) if b.status(i) != 0)
    paths = np.bincount([b.mesh_info(i).decode_path for i in range(0, n, max(1, n // 256))], minlength=3)
    eq = True
    for i in sorted(set(int(x) for x in np.linspace(0, n - 1, 4))):
        ref = oracle.decode(bytes(blob[int(offs[i]):int(offs[i + 1])]))
        m = b.result(i).ConnectedData
        eq = eq and np.array_equal(m.Faces, ref.faces) and all(np.array_equal(a.PortableValues, r.portable) and np.array_equal(a.PointMap, r.point_map) and a.Values.tobytes() == r.values.tobytes() for a, r in zip(m.Attributes, ref.attributes))
    print("%-28s %d meshes, %.1f ms per step = %.0f meshes/s; failed %d; decode paths (fast, general, retry) %s; sample equal to the oracle: %s; stages %s" %
          (label, n, dt * 1e3, n / dt, bad, paths.tolist(), eq, {k: round(v, 2) for k, v in b.stage_times().items()}), flush=True)
    b.close(); ctx.close() if hasattr(ctx, "close") else None
