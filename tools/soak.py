"""Randomised differential run: N synthetic meshes with random encoder options (topology, size, bit depths, symbol
scheme, prediction schemes, attribute order, Edgebreaker symbol coding, single / per-attribute connectivity) decoded
in one batch and compared with the oracle.  usage: python tools/soak.py [count] [seed]"""
import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle
import draco_sharp_amd as dsa, draco_sharp_amd.synth as synth


def random_case(rng):
    kind = int(rng.choice([synth.GRID, synth.TORUS, synth.SPHERE, synth.HOLES, synth.TWO_PARTS]))
    nx, ny = int(rng.integers(4, 48)), int(rng.integers(4, 40))
    if kind == synth.HOLES:
        nx, ny = max(nx, 12), max(ny, 12)
    opt = dict(pos_bits=int(rng.integers(4, 21)), uv_bits=int(rng.integers(4, 17)), normal_bits=int(rng.integers(3, 15)),
               single_connectivity=int(rng.integers(0, 2)), force_scheme=int(rng.integers(-1, 2)),
               compression_level=int(rng.integers(0, 11)), pos_prediction=int(rng.choice([0, 1, 2, 4])),
               uv_prediction=int(rng.choice([0, 1, 2, 4, 5])), normal_prediction=int(rng.choice([0, 6])),
               traversal_method=int(rng.integers(0, 3)), predictive_connectivity=int(rng.integers(0, 3)))
    # decoder branches no stock setting reaches (one case in three): the non-canonicalised octahedral transform, uncompressed
    # integers (the width must hold the zig-zagged corrections), prediction method -2
    if rng.integers(0, 3) == 0:
        opt["normal_transform"] = int(rng.choice([2, 3]))
        opt["no_prediction"] = int(rng.integers(0, 8))
        raw = int(rng.choice([0, 1, 2, 4]))
        if raw == 1:
            opt.update(pos_bits=min(opt["pos_bits"], 6), uv_bits=min(opt["uv_bits"], 6), normal_bits=min(opt["normal_bits"], 6))
        elif raw == 2:
            opt.update(pos_bits=min(opt["pos_bits"], 14), uv_bits=min(opt["uv_bits"], 14))
        opt["raw_integers"] = raw
    mesh_seed = int(rng.integers(0, 1 << 30))
    pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, mesh_seed)
    with_n, with_uv = bool(rng.integers(0, 4)), bool(rng.integers(0, 4))
    # one case in three gives its attributes per corner (attribute seams, corner attributes); they need a connectivity of their own
    if rng.integers(0, 3) == 0 and (with_n or with_uv):
        from meshutil import seamed_mesh
        patterns = ["stripes", "island", "checker", "random", "single", "none", None]
        charts = (str(rng.choice(patterns[:6])) if with_n and rng.integers(0, 2) else None, str(rng.choice(patterns[:6])) if with_uv and rng.integers(0, 2) else None)
        opt["single_connectivity"] = 0
        # half of these stay on the wave-per-mesh kernels whatever else was drawn (depth-first order, no predictive symbols), and a
        # third of those put tagged symbols into the corner attributes: the walk of the stream then stops in front of them and what
        # follows is located behind the seam tables
        if rng.integers(0, 2):
            opt.update(traversal_method=0, predictive_connectivity=int(rng.choice([0, 2])), pos_prediction=int(rng.choice([0, 1, 4])), uv_prediction=int(rng.choice([0, 1, 5])))
            if rng.integers(0, 3) == 0 and not opt.get("raw_integers"):
                opt["force_scheme"] = 0
        pos, faces, nrm, nid, uv, uid = seamed_mesh(synth, kind, nx, ny, mesh_seed, *charts)
        return synth.encode_mesh_corners(pos, faces, nrm if with_n else None, nid if with_n else None, uv if with_uv else None, uid if with_uv else None,
                                         opt=synth.options(**opt)), (kind, nx, ny, opt, with_n, with_uv, charts)
    # one per-vertex case in four carries a generic attribute of 1 - 4 uint8 components (vertex colours)
    gen = None
    if rng.integers(0, 4) == 0:
        gc = int(rng.integers(1, 5))
        opt["generic_components"] = gc
        gen = rng.integers(0, 256, (len(pos), gc)).astype(np.uint8)
    return synth.encode_mesh(pos, faces, nrm if with_n else None, uv if with_uv else None, generic=gen, opt=synth.options(**opt)), (kind, nx, ny, opt, with_n, with_uv)


def run(count, seed, ctx=None):
    rng = np.random.default_rng(seed)
    cases = [random_case(rng) for _ in range(count)]
    own = ctx is None
    ctx = ctx or dsa.Context(0)
    b = dsa.Batch(ctx, [c[0] for c in cases])
    b.decode()
    bad = []
    for i, (data, what) in enumerate(cases):
        try:
            ref = oracle.decode(data)
        except oracle.OracleError as e:          # a stream the reference refuses (its own limits): the device must refuse it too
            if b.status(i) == 0:
                bad.append((i, what, "oracle refuses (%s), device decodes" % e))
            continue
        if b.status(i) != 0:
            bad.append((i, what, "status %d site %d" % (b.status(i), b.mesh_info(i).detail)))
            continue
        m = b.result(i).ConnectedData
        ok = np.array_equal(m.Faces, ref.faces) and len(m.Attributes) == len(ref.attributes)
        for a, r in zip(m.Attributes, ref.attributes):
            ok = ok and np.array_equal(a.PortableValues, r.portable) and np.array_equal(a.PointMap, r.point_map) and a.Values.tobytes() == r.values.tobytes()
        if not ok:
            bad.append((i, what, "differs from the oracle"))
    b.close()
    if own:
        ctx.close()
    return bad


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = run(count, seed)
    print("%d cases, %d bad" % (count, len(bad)))
    for x in bad[:10]:
        print(x)
    sys.exit(1 if bad else 0)
