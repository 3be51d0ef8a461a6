"""Every combination of the writer's dialect switches on small meshes, against the oracle: seam pattern per attribute (none / seamed)
x symbol scheme (raw, tagged, uncompressed integers) x prediction of positions (difference, parallelogram, constrained
multi-parallelogram), texture coordinates (difference, parallelogram, TexCoordsPortable, constrained multi-parallelogram) and normals
(difference, GeometricNormal) x connectivity symbols (standard, valence) x attribute subset -- the combinations a random draw
reaches rarely (a late-located attribute beside a scheme that reuses a region, say).  usage: python tools/dialect_matrix.py [seed [big]]"""
import itertools, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
from meshutil import seamed_mesh
from test_gpu_parity import assert_same

def run(seed=1, ctx=None, big=False):
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
    # second family, attributes per vertex: quantisation bits (the rANS precision tiers), one decoder for all attributes or one each,
    # a generic attribute of 0 / 1 / 4 components, the symbol scheme left to the writer
    for ti, (kind, nx, ny) in enumerate(topologies[:3]):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx + 6, ny + 5, seed + 10 + ti)
        for bits, single, gc, scheme in itertools.product((dict(), dict(pos_bits=14, uv_bits=12, normal_bits=10)), (0, 1), (0, 1, 4), (-1, 0, 1)):
            gen = None if gc == 0 else ((np.arange(len(pos) * gc, dtype=np.int64) * 7919 + ti) % 256).astype(np.uint8).reshape(-1, gc)
            for pp, up, npred, conn in itertools.product((0, 1, 4), (0, 1, 5), (0, 6), (0, 2)):
                if (ti + pp + up + npred + conn + single + gc) % 3:      # a third of the product per topology
                    continue
                opt = dict(pos_prediction=pp, uv_prediction=up, normal_prediction=npred, predictive_connectivity=conn, single_connectivity=single,
                           force_scheme=scheme, generic_components=max(gc, 1), **bits)
                try:
                    s = synth.encode_mesh(pos, faces, nrm, uv, generic=gen, opt=synth.options(**opt))
                except RuntimeError:
                    continue
                streams.append(s); labels.append((kind, None, None, opt))
    # fourth family: quantisation bits and compression levels on meshes large enough for wide alphabets -- every rANS precision from 12
    # to 18 bits (Entropy/RAnsSymbolCoding.cs:10-27), the hand-scheduled decoders of 12 and of 13 - 15 bits, the compiled one above
    for ti, (kind, nx, ny) in enumerate(((synth.GRID, 70, 60), (synth.HOLES, 64, 50))):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, seed + 30 + ti)
        for k in range(36):
            opt = dict(pos_bits=8 + k % 11, uv_bits=8 + (k * 3) % 9, normal_bits=6 + (k * 5) % 9, compression_level=(0, 5, 7, 10)[k % 4], force_scheme=(1, -1)[k % 2],
                       pos_prediction=(1, 0, 4)[k % 3], predictive_connectivity=(0, 2)[(k // 2) % 2])
            streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt))); labels.append((kind, None, None, opt))
    # third family (big=True): 65 536-triangle meshes, where rings are longer than the chains' LDS windows and ids need all their bits
    if big:
        for ti, kind in enumerate((synth.GRID, synth.TORUS)):
            for u_chart in (None, "stripes"):
                args = seamed_mesh(synth, kind, 128, 256, seed + 20 + ti, None, u_chart)
                for pp, up, npred, conn, scheme in itertools.product((1, 4), (1, 5), (0, 6), (0, 2), (-1, 0)):
                    opt = dict(pos_prediction=pp, uv_prediction=up, normal_prediction=npred, predictive_connectivity=conn, force_scheme=scheme)
                    streams.append(synth.encode_mesh_corners(*args, opt=synth.options(**opt))); labels.append((kind, None, u_chart, opt))
    print(len(streams), "streams", flush=True)
    bad = 0
    paths = {}
    precisions = set()
    # in parts of at most 2048 (connectivity and traversal as two kernels) and, second pass, as one crowded batch (k_chain, the
    # operands by the traversal waves, the octahedral streams kernel, the register gate)
    parts = [(list(range(at, min(at + 2048, len(streams))))) for at in range(0, len(streams), 2048)] + [list(range(len(streams)))]
    # third pass: the streams without corner attributes alone, five times over -- a crowded batch that takes k_chain
    plain = [i for i, (_, n_chart, u_chart, _) in enumerate(labels) if n_chart is None and u_chart is None]
    parts.append((plain * 5)[:max(2100, len(plain))] if len(plain) * 5 >= 2100 else plain * 5)
    for idx in parts:
        part = [streams[i] for i in idx]
        at = None
        b = dsa.Batch(ctx, part)
        b.decode()
        for i, s in enumerate(part):
            ref = oracle.decode(s)
            info = b.mesh_info(i)
            paths[info.decode_path] = paths.get(info.decode_path, 0) + 1
            if info.decode_path == 0:
                for src, _, prec, _ in b.debug_array(i, 5, np.uint32, 64).reshape(16, 4)[:len(ref.attributes)]:
                    if src == 1:
                        precisions.add(int(prec))
            try:
                assert b.status(i) == 0, (b.status(i), info.detail)
                assert_same(b.result(i), ref)
            except AssertionError as e:
                bad += 1
                if bad <= 12:
                    print("BAD", labels[idx[i]], "path", info.decode_path, "batch of", len(part), str(e)[:100], flush=True)
        b.close()
    print("%d streams (%d decodes in %d batches), %d bad, decode paths %s, rANS precisions of raw streams on the fast kernels %s" % (len(streams), sum(len(x) for x in parts), len(parts), bad, paths, sorted(precisions)))
    if own:
        ctx.close()
    return len(streams), bad, paths


if __name__ == "__main__":
    n, bad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, big=len(sys.argv) > 2 and sys.argv[2] == "big")
    sys.exit(1 if bad else 0)
