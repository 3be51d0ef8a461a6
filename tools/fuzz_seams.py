"""Corrupted seamed / TexCoordsPortable streams through the fast kernels against the oracle: where the oracle decodes, the device
must decode the same or refuse; where it refuses, the device must not succeed with different data.
usage: python tools/fuzz_seams.py [corruptions per family] [seed]"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle
import draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
from meshutil import seamed_mesh
from test_gpu_parity import _corruptions, assert_same

count = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = dsa.Context(0)
families = []
for k, (kind, nx, ny, charts, opt) in enumerate([
        (synth.TORUS, 12, 10, ("checker", "stripes"), dict(force_scheme=1, uv_prediction=5)),
        (synth.HOLES, 20, 16, (None, "random"), dict(force_scheme=1, predictive_connectivity=2)),
        (synth.GRID, 24, 20, ("stripes", "island"), dict(force_scheme=1, uv_prediction=5, predictive_connectivity=2)),
        (synth.SPHERE, 10, 9, (None, "checker"), dict(force_scheme=1))]):
    families.append(synth.encode_mesh_corners(*seamed_mesh(synth, kind, nx, ny, 5 + k, *charts), opt=synth.options(**opt)))
pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 16, 12, 9)
families.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(uv_prediction=5, normal_prediction=6, predictive_connectivity=2, force_scheme=1)))
# ConstrainedMultiParallelogram positions (k_crease_bits, k_multipara_prepare, k_multipara), per vertex and beside seamed attributes
families.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=4, uv_prediction=5, normal_prediction=6, predictive_connectivity=2)))
families.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=4, force_scheme=0)))
families.append(synth.encode_mesh_corners(*seamed_mesh(synth, synth.HOLES, 20, 16, 11, None, "stripes"), opt=synth.options(pos_prediction=4, uv_prediction=5)))
families.append(open(os.path.join(ROOT, "tests", "golden", "house_04.obj.drc"), "rb").read())
streams = []
for k, f in enumerate(families):
    streams += _corruptions(f, count, seed * 100 + k) + [f]
b = dsa.Batch(ctx, streams)
b.decode()
ok = refused = stricter = wrong = 0
sites = {}
for i, sbytes in enumerate(streams):
    try:
        ref = oracle.decode(sbytes)
    except oracle.OracleError:
        ref = None
    st = b.status(i)
    if ref is not None and st == 0:
        try:
            assert_same(b.result(i), ref)
            ok += 1
        except AssertionError:
            wrong += 1
            print("DIFFERENT", i, b.mesh_info(i).decode_path)
    elif ref is None:
        if st == 0:
            wrong += 1
            print("ACCEPTED what the oracle refuses", i, b.mesh_info(i).decode_path)
        refused += 1
    else:
        stricter += 1
        info = b.mesh_info(i)
        sites[(info.status, info.detail)] = sites.get((info.status, info.detail), 0) + 1
print("streams %d: equal %d, both refuse %d, device stricter %d %s, wrong %d" % (len(streams), ok, refused, stricter, sites, wrong))
sys.exit(1 if wrong else 0)
