"""tests/golden/dialect_vectors.{json,bin}: committed streams of the dialects stock encoders write (seams, TexCoordsPortable,
GeometricNormal, valence symbols, ConstrainedMultiParallelogram) with the digests of what the oracle decoded from them when they
were made.  CPU: the oracle still decodes the committed bytes to the same arrays, and the CPU coder still writes the same bytes.
GPU: the device path decodes the committed bytes to the same arrays, on the wave-per-mesh kernels."""
import hashlib
import json
import os
import struct
import sys

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_dialect_vectors as mk          # noqa: E402  (the generating script: its digest() is the definition of the digests)


def load():
    rows = json.load(open(os.path.join(HERE, "golden", "dialect_vectors.json")))
    raw = open(os.path.join(HERE, "golden", "dialect_vectors.bin"), "rb").read()
    streams, at = [], 0
    for r in rows:
        n, = struct.unpack_from("<I", raw, at)
        streams.append(raw[at + 4:at + 4 + n])
        at += 4 + n
    assert at == len(raw) and len(streams) == len(rows)
    return rows, streams


def test_the_oracle_decodes_the_committed_streams_to_the_committed_digests():
    rows, streams = load()
    assert len(rows) == 28
    for r, s in zip(rows, streams):
        assert len(s) == r["bytes"] and hashlib.sha256(s).hexdigest() == r["stream_sha256"], r["name"]
        m = oracle.decode(s)
        assert (m.num_faces, m.num_points) == (r["faces"], r["points"]), r["name"]
        assert [a.pred_method for a in m.attributes] == r["pred_methods"], r["name"]
        assert mk.digest(m) == r["decoded_sha256"], r["name"]
    # every scheme of the round is among them
    methods = {m for r in rows for m in r["pred_methods"]}
    assert {1, 4, 5, 6} <= methods


def test_the_cpu_coder_still_writes_the_committed_bytes():
    rows, streams = load()
    for (name, data), r, s in zip(mk.cases(), rows, streams):
        assert name == r["name"]
        assert data == s, name


@pytest.mark.gpu
def test_the_device_decodes_the_committed_streams_to_the_committed_digests():
    import draco_sharp_amd as dsa
    from test_gpu_parity import assert_same
    rows, streams = load()
    ctx = dsa.Context(0)
    b = dsa.Batch(ctx, streams)
    b.decode()

    class View:                       # what digest() reads, from the device's result
        pass
    for i, (r, s) in enumerate(zip(rows, streams)):
        assert b.status(i) == 0, (r["name"], b.mesh_info(i).detail)
        assert b.mesh_info(i).decode_path == 0, r["name"]
        ref = oracle.decode(s)
        got = b.result(i)
        assert_same(got, ref, b, i)
        m = got.ConnectedData
        v = View()
        v.faces = m.Faces
        v.attributes = []
        for a, ra in zip(m.Attributes, ref.attributes):
            x = View()
            x.att_type, x.data_type, x.num_components, x.num_entries = a.AttributeType, a.DataType, a.NumComponents, a.UniqueEntriesCount
            x.point_map = a.PointMap if len(ra.point_map) else np.zeros(0, np.uint32)       # (the oracle reports an identity map as empty)
            x.portable = a.PortableValues if ra.portable is not None else None
            x.values = a.Values
            v.attributes.append(x)
        assert mk.digest(v) == r["decoded_sha256"], r["name"]
    b.close()
    ctx.close()
