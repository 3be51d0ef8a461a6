"""Every combination of the writer's dialect switches on small meshes, against the oracle: seam pattern per attribute (none / seamed)
x symbol scheme (raw, tagged, uncompressed integers) x prediction of positions (difference, parallelogram, constrained
multi-parallelogram), texture coordinates (difference, parallelogram, TexCoordsPortable, constrained multi-parallelogram) and normals
(difference, GeometricNormal) x connectivity symbols (standard, valence) x attribute subset -- the combinations a random draw
reaches rarely (a late-located attribute beside a scheme that reuses a region, say).  usage: python tools/dialect_matrix.py [seed]"""
import itertools, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
from meshutil import seamed_mesh
from test_gpu_parity import assert_same

def run(seed=1, ctx=None):
    own = ctx is None
    if own:
        ctx = dsa.Context(0)
    topologies = [(synth.GRID, 14, 11), (synth.TORUS, 12, 9), (synth.HOLES, 16, 13), (synth.TWO_PARTS, 9, 6)]
    streams, labels = [], []
    for ti, (kind, nx, ny) in enumerate(topologies):
        for n_chart, u_chart in itertools.product((None, "stripes", "random"), (None, "stripes", "checker")):
            args = seamed_mesh(synth, kind, nx, ny, seed + ti, n_chart, u_chart)
            for scheme in (dict(force_scheme=1), dict(force_scheme=0), dict(raw_integers=2, pos_bits=10, uv_bits=10)):
                for pp, up, npred, conn in itertools.product((0, 1, 4), (0, 1, 5, 4), (0, 6), (0, 2)):
                    if (ti + pp + up + npred + conn) % 2:            # half of the product per topology (each combination on two of the four)
                        continue
                    opt = dict(pos_prediction=pp, uv_prediction=up, normal_prediction=npred, predictive_connectivity=conn, **scheme)
                    try:
                        if n_chart is None and u_chart is None:
                            pos, faces, nrm, nid, uv, uid = args
                            # per-vertex: the same values through the per-vertex entry point (corner ids are the identity here)
                            s = synth.encode_mesh_corners(*args, opt=synth.options(**opt))
                        else:
                            s = synth.encode_mesh_corners(*args, opt=synth.options(**opt))
                    except RuntimeError:
                        continue                                        # (a combination the writer refuses)
                    streams.append(s); labels.append((kind, n_chart, u_chart, opt))
    print(len(streams), "streams", flush=True)
    bad = 0
    paths = {}
    for at in range(0, len(streams), 2048):
        part = streams[at:at + 2048]
        b = dsa.Batch(ctx, part)
        b.decode()
        for i, s in enumerate(part):
            ref = oracle.decode(s)
            info = b.mesh_info(i)
            paths[info.decode_path] = paths.get(info.decode_path, 0) + 1
            try:
                assert b.status(i) == 0, (b.status(i), info.detail)
                assert_same(b.result(i), ref)
            except AssertionError as e:
                bad += 1
                if bad <= 12:
                    print("BAD", labels[at + i], "path", info.decode_path, str(e)[:100], flush=True)
        b.close()
    print("%d streams, %d bad, decode paths %s" % (len(streams), bad, paths))
    if own:
        ctx.close()
    return len(streams), bad, paths


if __name__ == "__main__":
    n, bad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    sys.exit(1 if bad else 0)
