"""Step time of the bench batch re-encoded with attribute seams: 4096 x 65 536-triangle meshes whose texture coordinates (and
optionally normals) are given per corner (three UV charts: seams from boundary to boundary), with parallelogram or TexCoordsPortable
prediction, standard or valence connectivity.  The batch cycles through 32 distinct meshes (the writer is the Python-driven CPU
coder).  Also: positions by ConstrainedMultiParallelogram (what stock encoders write at level 9).
usage: python tools/seam_timing.py [meshes [variant numbers, comma separated]]; SEAM_TIMING_PAIR=1 in the environment adds the
rate with two batches of the same streams in flight."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle
import draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
from meshutil import seamed_mesh
from test_gpu_parity import assert_same

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None       # variant numbers
ctx = dsa.Context(0)
ctx.set_profiling(True)
variants = [("per-vertex, parallelogram (the bench batch)", None, dict()),
            ("UV seams, parallelogram", (None, "stripes"), dict()),
            ("UV seams, TexCoordsPortable", (None, "stripes"), dict(uv_prediction=5)),
            ("UV + normal seams, TexCoordsPortable", ("checker", "stripes"), dict(uv_prediction=5)),
            ("UV seams, TexCoordsPortable, valence", (None, "stripes"), dict(uv_prediction=5, predictive_connectivity=2)),
            ("per-vertex, TexCoordsPortable", None, dict(uv_prediction=5)),
            ("stock default: valence + GeometricNormal + TexCoordsPortable, per vertex", None, dict(uv_prediction=5, predictive_connectivity=2, normal_prediction=6)),
            ("stock default with UV seams", (None, "stripes"), dict(uv_prediction=5, predictive_connectivity=2, normal_prediction=6)),
            ("stock default with UV and normal seams (3 charts each)", ("stripes", "stripes"), dict(uv_prediction=5, predictive_connectivity=2, normal_prediction=6)),
            ("positions by ConstrainedMultiParallelogram, rest as the bench batch", None, dict(pos_prediction=4)),
            ("stock highest levels: valence + ConstrainedMultiParallelogram + GeometricNormal + TexCoordsPortable", None, dict(pos_prediction=4, uv_prediction=5, predictive_connectivity=2, normal_prediction=6)),
            ("stock highest levels with UV seams", (None, "stripes"), dict(pos_prediction=4, uv_prediction=5, predictive_connectivity=2, normal_prediction=6))]
for vi, (name, charts, opt) in enumerate(variants):
    if only is not None and vi not in only:
        continue
    distinct = []
    for k in range(32):
        if charts is None:
            pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 128, 256, 1000 + k)
            distinct.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt)))
        else:
            distinct.append(synth.encode_mesh_corners(*seamed_mesh(synth, synth.GRID, 128, 256, 1000 + k, *charts), opt=synth.options(**opt)))
    streams = [distinct[i % 32] for i in range(n)]
    b = dsa.Batch(ctx, streams)
    b.decode()
    b.decode()
    t0 = time.perf_counter()
    for _ in range(3):
        b.decode()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    paths = sorted(set(b.mesh_info(i).decode_path for i in range(0, n, 61)))
    bad = sum(1 for i in range(n) if b.status(i) != 0)
    for i in (0, 1, 31):
        assert_same(b.result(i), oracle.decode(streams[i]))
    kt = {k: round(v, 2) for k, v in b.kernel_times().items()}
    if os.environ.get("SEAM_TIMING_PAIR"):            # two batches of the same streams in flight (the context's two stream sets)
        b2 = dsa.Batch(ctx, streams)
        pair = [b, b2]
        for k in range(2): pair[k].decode(wait=False)
        for x in pair: x.wait()
        t0 = time.perf_counter()
        for k in range(6): pair[k % 2].decode(wait=False)
        for x in pair: x.wait()
        ms2 = (time.perf_counter() - t0) / 6 * 1e3
        bad2 = sum(1 for x in pair for i in range(0, n, 16) if x.status(i) != 0)
        assert_same(b2.result(1), oracle.decode(streams[1]))
        kt["pair_ms_per_step"] = round(ms2, 2); kt["pair_meshes_per_s"] = round(n / ms2 * 1e3); kt["pair_failed"] = bad2
        b2.close()
    print("%-75s %7.2f ms  %7.0f meshes/s  paths %s failed %d  bytes/mesh %d\n    %s" % (name, ms, n / ms * 1e3, paths, bad, len(distinct[0]), kt), flush=True)
    b.close()
    ctx.trim()
