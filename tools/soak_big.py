"""Randomised differential run on larger meshes (6k-70k triangles, fast-path options): long symbol streams with
many reservoir refills and window reloads at arbitrary alignments; the chains of TexCoordsPortable and ConstrainedMultiParallelogram
over rings longer than their LDS windows.  usage: python tools/soak_big.py [seed]"""
import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, 'tools')
import numpy as np, oracle, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
cases = []
for k in range(160):
    kind = int(rng.choice([synth.GRID, synth.TORUS, synth.HOLES, synth.TWO_PARTS, synth.SPHERE]))
    nx, ny = int(rng.integers(60, 220)), int(rng.integers(50, 160))
    opt = dict(pos_bits=int(rng.integers(8, 17)), uv_bits=int(rng.integers(8, 15)), normal_bits=int(rng.integers(6, 13)),
               single_connectivity=int(rng.integers(0, 2)), force_scheme=int(rng.choice([-1, -1, 1, 0])), compression_level=int(rng.integers(3, 9)),
               pos_prediction=int(rng.choice([0, 1, 1, 4])), uv_prediction=int(rng.choice([0, 1, 1, 5])),
               normal_prediction=int(rng.choice([0, 0, 6])), predictive_connectivity=int(rng.choice([0, 0, 2])))
    pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, int(rng.integers(0, 1 << 30)))
    cases.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt)))
ctx = dsa.Context(0)
b = dsa.Batch(ctx, cases); b.decode()
bad = 0
for i, s in enumerate(cases):
    ref = oracle.decode(s)
    if b.status(i) != 0: bad += 1; print("status", i, b.status(i), b.mesh_info(i).detail); continue
    m = b.result(i).ConnectedData
    ok = np.array_equal(m.Faces, ref.faces)
    for a, r in zip(m.Attributes, ref.attributes):
        ok = ok and np.array_equal(a.PortableValues, r.portable) and np.array_equal(a.PointMap, r.point_map) and a.Values.tobytes() == r.values.tobytes()
    if not ok: bad += 1; print("differs", i)
print(len(cases), "cases,", bad, "bad; total faces", sum(oracle.decode(s).num_faces for s in cases[:5]), "...")
sys.exit(1 if bad else 0)
