"""Streams that carry a metadata block (header flag 0x8000, Metadata/MetadataDecoder.cs:5-49): the decode path must
step over it exactly -- three skippers: the oracle's, the host sizing parse and k_locate's explicit-stack one -- and the
host mirror parses the block into DracoMetadata."""
import numpy as np
import pytest

import oracle
import draco_sharp_amd as dsa
import draco_sharp_amd.synth as synth
from meshutil import metadata_element, with_metadata


def _cases():
    pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 10, 8, 3)
    plain = synth.encode_mesh(pos, faces, nrm, uv)
    leaf = metadata_element([(b"name", b"torus"), (b"big", bytes(range(256)) * 2)])
    deep = metadata_element([(b"k", b"v")])
    for level in range(6):
        deep = metadata_element([(b"level", bytes([level]))], [(b"child", deep), (b"twin", metadata_element())])
    file_el = metadata_element([(b"generator", b"draco-sharp_amd tests"), (b"", b"")], [(b"materials", leaf), (b"tree", deep)])
    atts = [(0, metadata_element([(b"name", b"position")])), (2, metadata_element([(b"name", b"uv")], [(b"sampler", leaf)]))]
    return plain, [with_metadata(plain, [], metadata_element()), with_metadata(plain, atts, file_el),
                   with_metadata(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(predictive_connectivity=2, uv_prediction=5)), atts, file_el)]


def test_oracle_steps_over_metadata():
    plain, cases = _cases()
    ref = oracle.decode(plain)
    for data, block in cases:
        m = oracle.decode(data)
        assert m.end_pos == len(data) and m.flags == 0x8000
        assert np.array_equal(m.faces, ref.faces)
        for a, r in zip(m.attributes, ref.attributes):
            assert np.array_equal(a.portable, r.portable)


def test_metadata_block_parses_into_the_object_model():
    _, cases = _cases()
    md = dsa.parse_metadata(cases[0][1])
    assert md.Attributes == [] and md.File.Keys == [] and md.File.SubMetadata == []
    md = dsa.parse_metadata(cases[1][1])
    assert [e.Id for e in md.Attributes] == [0, 2] and md.Attributes[0].GetEntry("name") == b"position"
    assert md.File.GetEntry("generator") == b"draco-sharp_amd tests" and md.File.GetEntry(b"") == b""
    assert md.File.SubMetadataKeys == [b"materials", b"tree"]
    assert md.File.SubMetadata[0].GetEntry("big") == bytes(range(256)) * 2
    node, depth = md.File.SubMetadata[1], 0
    while node.SubMetadata:
        assert node.SubMetadataKeys == [b"child", b"twin"] and node.SubMetadata[1].Keys == []
        node, depth = node.SubMetadata[0], depth + 1
    assert depth == 6 and node.GetEntry("k") == b"v"
    with pytest.raises(dsa.InvalidDataException):
        dsa.parse_metadata(cases[1][1][:-3])


def test_metadata_nesting_limit_is_the_same_everywhere():
    plain, _ = _cases()
    e = metadata_element()
    for _ in range(15):
        e = metadata_element([], [(b"d", e)])
    ok, block = with_metadata(plain, [], e)              # 16 levels: accepted
    assert oracle.decode(ok).end_pos == len(ok)
    dsa.parse_metadata(block)
    too_deep, block = with_metadata(plain, [], metadata_element([], [(b"d", e)]))
    with pytest.raises(oracle.OracleError):
        oracle.decode(too_deep)
    with pytest.raises(dsa.InvalidDataException):
        dsa.parse_metadata(block)


@pytest.mark.gpu
def test_gpu_decode_with_metadata():
    plain, cases = _cases()
    e = metadata_element()
    for _ in range(15):
        e = metadata_element([], [(b"d", e)])
    deep_ok = with_metadata(plain, [], e)[0]
    too_deep = with_metadata(plain, [], metadata_element([], [(b"d", e)]))[0]
    truncated = cases[1][0][:40]
    streams = [plain] + [c[0] for c in cases] + [deep_ok, too_deep, truncated]
    ctx = dsa.Context(0)
    b = dsa.Batch(ctx, streams)
    b.decode()
    ref = oracle.decode(plain)
    for i in range(5):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        d = b.result(i)
        want = oracle.decode(streams[i])
        assert np.array_equal(d.ConnectedData.Faces, want.faces)
        for a, r in zip(d.ConnectedData.Attributes, want.attributes):
            assert np.array_equal(a.PortableValues, r.portable) and a.Values.tobytes() == r.values.tobytes()
        if i == 0:
            assert d.Metadata is None
        else:
            assert d.Header.Flags == 0x8000 and d.Metadata is not None
    md = b.result(2).Metadata
    assert [x.Id for x in md.Attributes] == [0, 2] and md.File.SubMetadataKeys == [b"materials", b"tree"]
    assert b.result(3).Metadata.File.GetEntry("generator") == b"draco-sharp_amd tests"      # general-path mesh
    assert np.array_equal(b.result(1).ConnectedData.Faces, ref.faces)
    assert b.status(5) == 1 and b.status(6) == 1                                              # DSA_ERR_INVALID_DATA
    b.close()
    ctx.close()
