"""Which kernels decode seamed / TexCoordsPortable streams: decode_path per case (0 = the wave-per-mesh kernels, 2 = second chance on
the general path) and equality with the oracle.  usage: python tools/seam_paths.py"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle
import draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
from meshutil import seamed_mesh
from test_gpu_parity import assert_same

ctx = dsa.Context(0)
cases = []
for kind, nx, ny in [(synth.GRID, 12, 9), (synth.TORUS, 10, 8), (synth.SPHERE, 8, 7), (synth.HOLES, 20, 16), (synth.TWO_PARTS, 8, 6), (synth.GRID, 40, 33)]:
    for charts in [(None, "stripes"), ("checker", "island"), ("random", "random"), ("single", "none")]:
        for opt in [dict(), dict(predictive_connectivity=2), dict(uv_prediction=5), dict(uv_prediction=5, predictive_connectivity=2, normal_prediction=6), dict(force_scheme=0)]:
            args = seamed_mesh(synth, kind, nx, ny, 11, *charts)
            cases.append(((kind, nx, ny, charts, opt), synth.encode_mesh_corners(*args, opt=synth.options(**opt))))
for kind, nx, ny in [(synth.GRID, 12, 9), (synth.TORUS, 24, 40), (synth.HOLES, 20, 16)]:
    for opt in [dict(uv_prediction=5), dict(uv_prediction=5, predictive_connectivity=2), dict(uv_prediction=5, single_connectivity=1), dict(uv_prediction=5, normal_prediction=6, predictive_connectivity=2)]:
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 3)
        cases.append(((kind, nx, ny, "per-vertex", opt), synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt))))
cases.append((("house_04.obj.drc",), open(os.path.join(ROOT, "tests", "golden", "house_04.obj.drc"), "rb").read()))
b = dsa.Batch(ctx, [s for _, s in cases])
b.decode()
bad = 0
for i, (what, s) in enumerate(cases):
    info = b.mesh_info(i)
    ok = True
    if info.status == 0:
        try:
            assert_same(b.result(i), oracle.decode(s))
        except AssertionError:
            ok = False
    print(what, "status", info.status, "detail", info.detail, "path", info.decode_path, "equal" if ok else "DIFFERENT")
    bad += (not ok) or info.status != 0
print("bad", bad)
