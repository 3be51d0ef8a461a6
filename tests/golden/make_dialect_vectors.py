"""Golden vectors of the dialects the fast kernels took on in round 4: small streams written by the repo's CPU coder
(draco-sharp_amd/synth) and what the oracle decodes from them, as SHA-256 digests -- one line per stream in dialect_vectors.json.
They pin writer and oracle against each other ACROSS rounds (a change to either that alters a byte shows here), and give the GPU
suite a fixed set of streams per dialect: attribute seams, TexCoordsPortable, GeometricNormal, valence symbols,
ConstrainedMultiParallelogram.  The streams themselves are committed too (dialect_vectors.bin: lengths + bytes), so that the
digests are checked on what was written THEN, not on what the writer produces now.
usage: python tests/golden/make_dialect_vectors.py     (rewrites both files)"""
import hashlib, json, os, struct, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import oracle
import draco_sharp_amd.synth as synth
from meshutil import seamed_mesh


def digest(m):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(m.faces, np.int32).tobytes())
    for a in m.attributes:
        h.update(struct.pack("<iiiII", a.att_type, a.data_type, a.num_components, a.num_entries, len(a.point_map)))
        h.update(np.ascontiguousarray(a.point_map, np.uint32).tobytes())
        if a.portable is not None:
            h.update(np.ascontiguousarray(a.portable, np.int32).tobytes())
        h.update(a.values.tobytes())
    return h.hexdigest()


def cases():
    out = []
    for kind, nx, ny, name in ((synth.GRID, 24, 17, "grid"), (synth.TORUS, 16, 12, "torus"), (synth.HOLES, 20, 16, "holes"), (synth.TWO_PARTS, 9, 6, "two_parts")):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 77)
        for label, opt in (("default", dict()), ("stock_level7", dict(uv_prediction=5, normal_prediction=6, predictive_connectivity=2)),
                           ("stock_level9", dict(pos_prediction=4, uv_prediction=5, normal_prediction=6, predictive_connectivity=2)),
                           ("tagged_14bit", dict(force_scheme=0, pos_bits=14))):
            out.append(("%s/%s" % (name, label), synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt))))
        for label, charts, opt in (("uv_seams", (None, "stripes"), dict()), ("uv_normal_seams_level7", ("checker", "island"), dict(uv_prediction=5, normal_prediction=6, predictive_connectivity=2)),
                                   ("uv_seams_level9", (None, "random"), dict(pos_prediction=4, uv_prediction=5))):
            out.append(("%s/%s" % (name, label), synth.encode_mesh_corners(*seamed_mesh(synth, kind, nx, ny, 78, *charts), opt=synth.options(**opt))))
    return out


if __name__ == "__main__":
    rows, blob = [], bytearray()
    for name, data in cases():
        m = oracle.decode(data)
        rows.append({"name": name, "bytes": len(data), "stream_sha256": hashlib.sha256(data).hexdigest(), "decoded_sha256": digest(m),
                     "faces": int(m.num_faces), "points": int(m.num_points), "pred_methods": [int(a.pred_method) for a in m.attributes]})
        blob += struct.pack("<I", len(data)) + data
    with open(os.path.join(HERE, "dialect_vectors.json"), "w") as f:
        json.dump(rows, f, indent=1)
    with open(os.path.join(HERE, "dialect_vectors.bin"), "wb") as f:
        f.write(bytes(blob))
    print(len(rows), "vectors,", len(blob), "bytes")
