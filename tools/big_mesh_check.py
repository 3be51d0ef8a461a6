"""One-off: parity + stage times on multi-million-triangle meshes (single wave per mesh: latency-bound)."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, time, oracle, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
import test_gpu_parity as T
ctx = dsa.Context(0); ctx.set_profiling(True)
streams = []
for kind, nx, ny in ((synth.GRID, 1000, 1000), (synth.TORUS, 700, 900), (synth.HOLES, 600, 500)):
    pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 9)
    streams.append(synth.encode_mesh(pos, faces, nrm, uv))
    print(kind, nx, ny, len(faces), len(streams[-1]))
    if "--general" in sys.argv:     # the same mesh as a stock encoder would write it: valence symbols, TexCoordsPortable, GeometricNormal
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(predictive_connectivity=2, uv_prediction=5, normal_prediction=6,
                                                                                  pos_prediction=4 if kind == synth.TORUS else 1, traversal_method=1 if kind == synth.HOLES else 0)))
b = dsa.Batch(ctx, streams)
for _ in range(2):
    t0 = time.time(); b.decode(); print("decode s", round(time.time() - t0, 3), {k: round(v, 1) for k, v in b.stage_times().items()})
for i, s in enumerate(streams):
    assert b.status(i) == 0, (i, b.status(i), b.mesh_info(i).detail)
    if "--check" in sys.argv: T.assert_same(b.result(i), oracle.decode(s), b, i)
d = np.array([b.debug_array(i, 4, np.uint32, 12) for i in range(len(streams))]); print(d)
