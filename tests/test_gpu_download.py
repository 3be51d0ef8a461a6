"""The boundary as a draco-sharp caller sees it: host bytes in, host arrays out (DracoDecoder.cs:19-42 returns host objects).
dsa_batch_download moves every output array of a batch to the host in one transfer; two batches of one context can be in
flight (upload of the second beside the kernels of the first); every path must deliver the oracle's arrays."""
import os

import numpy as np
import pytest

import draco_sharp_amd as dsa
import draco_sharp_amd.synth as synth
import oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ctx():
    c = dsa.Context(0)
    yield c
    c.close()


def streams_mixed(seed):
    out = []
    for k, (kind, nx, ny) in enumerate(((synth.GRID, 40, 33), (synth.TORUS, 24, 20), (synth.HOLES, 20, 16), (synth.TWO_PARTS, 12, 9))):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, seed + k)
        out.append(synth.encode_mesh(pos, faces, nrm, uv))
    with open(os.path.join(HERE, "golden", "house_04.obj.drc"), "rb") as f:
        out.append(f.read())                                           # general path (valence, seams)
    pos, nrm, uv, faces = synth.make_mesh(synth.HOLES, 14, 12, seed)
    out.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(normal_prediction=6)))   # second chance: block 1
    # stock level 9 with vertex colours: ConstrainedMultiParallelogram positions and a four-component uint8 attribute (its rows are
    # 4 bytes in the value block, not 4 x 4)
    gen = ((np.arange(len(pos) * 4, dtype=np.int64) * 7919 + seed) % 256).astype(np.uint8).reshape(-1, 4)
    out.append(synth.encode_mesh(pos, faces, nrm, uv, generic=gen, opt=synth.options(pos_prediction=4, uv_prediction=5, normal_prediction=6, predictive_connectivity=2,
                                                                                     generic_components=4)))
    out.append(b"DRACO\x02\x02\x01\x01\x00\x00garbage")               # a bad stream fails alone
    out.append(synth.encode_point_cloud(np.random.default_rng(seed).random((500, 3), np.float32)))
    return out


def check_views(b, streams, compact=False):
    for i, s in enumerate(streams):
        try:
            ref = oracle.decode(s)
        except oracle.OracleError:
            assert b.status(i) != 0
            with pytest.raises(Exception):
                b.host_views(i)
            continue
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        v = b.host_views(i)
        assert np.array_equal(v["faces"], ref.faces)
        assert len(v["attributes"]) == len(ref.attributes)
        ident = np.arange(ref.num_points, dtype=np.uint32)
        for a, r in zip(v["attributes"], ref.attributes):
            assert a["values"].tobytes() == r.values.tobytes()
            assert np.array_equal(ident if a["point_map"] is None else a["point_map"], r.point_map if len(r.point_map) else ident)
        if compact and b.mesh_info(i).decode_path != 2:        # (meshes decoded a second time keep the full layout)
            # the compact host copy: uint16 faces where the points fit, one map array for attributes decoded in one order
            assert v["faces"].dtype == (np.uint16 if ref.num_points <= 65536 and len(ref.faces) else v["faces"].dtype)
            maps = [a["point_map"] for a in v["attributes"]]
            if ref.encoder_type == 0:
                assert all(m is None for m in maps)
            elif b.mesh_info(i).decode_path == 0:
                pos_like = [m for m, d in zip(maps, ref.decoders_of_attributes()) if d == 0]
                assert all(m.ctypes.data == pos_like[0].ctypes.data for m in pos_like)
        # and the per-array accessors, now served from the host copy, agree
        m = b.result(i).ConnectedData
        assert np.array_equal(getattr(m, "Faces", np.zeros((0, 3), np.int32)), ref.faces)
        for a, r in zip(m.Attributes, ref.attributes):
            assert a.Values.tobytes() == r.values.tobytes()


def test_download_delivers_every_array(ctx):
    streams = streams_mixed(5)
    b = dsa.Batch(ctx, streams)
    b.decode(wait=False)
    b.download(wait=False)          # queued behind the kernels; one wait for both
    b.wait()
    assert b.output_bytes > 0
    check_views(b, streams)
    # download after the results were collected, and again after a second decode of the same batch
    b2 = dsa.Batch(ctx, streams)
    b2.decode()
    b2.download()
    check_views(b2, streams)
    b2.decode(wait=False)
    b2.download()
    check_views(b2, streams)
    b.close(); b2.close()


def test_compact_download_delivers_every_array(ctx):
    """dsa_batch_download_compact: a third less on the link for the same arrays -- uint16 faces, shared maps, no identity maps --
    through the zero-copy views and, widened, through the per-array accessors; seamed meshes keep a map per corner decoder."""
    from meshutil import seamed_mesh
    streams = streams_mixed(9)
    streams.append(synth.encode_mesh_corners(*seamed_mesh(synth, synth.GRID, 40, 33, 3, "checker", "stripes"), opt=synth.options(force_scheme=1)))
    b = dsa.Batch(ctx, streams)
    b.decode(wait=False)
    b.download(wait=False, compact=True)
    b.wait()
    assert 0 < b.compact_bytes < b.output_bytes
    check_views(b, streams, compact=True)
    b.decode(wait=False)                      # and the full layout again on the same batch
    b.download(compact=False)
    check_views(b, streams)
    b.decode(wait=False)
    b.download(compact=True)
    check_views(b, streams, compact=True)
    b.close()


def test_two_batches_in_flight(ctx):
    """upload(k+1) beside decode(k) beside download(k-1): nothing of one batch may leak into another."""
    sets = [streams_mixed(11 + 7 * k) for k in range(4)]
    live = []
    for k, streams in enumerate(sets):
        b = dsa.Batch(ctx, streams)        # staged + upload queued
        b.decode(wait=False)
        b.download(wait=False)
        live.append((b, streams))
        if len(live) == 3:
            b0, s0 = live.pop(0)
            b0.wait()
            check_views(b0, s0)
            b0.close()                     # its arena and mirror go back to the context for the next batch
    for b0, s0 in live:
        b0.wait()
        check_views(b0, s0)
        b0.close()


def test_download_into_caller_memory(ctx):
    import ctypes as C
    from draco_sharp_amd import native
    L = native.lib()
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 64, 48, 2)
    s = synth.encode_mesh(pos, faces, nrm, uv)
    b = dsa.Batch(ctx, [s, s])
    b.decode(wait=False)
    nbytes = b.output_bytes
    p = L.dsa_host_alloc(nbytes)
    assert p
    try:
        assert L.dsa_batch_download(b._h, p, nbytes - 1) != 0          # too small a destination is refused
        assert L.dsa_batch_download(b._h, p, nbytes) == 0
        b.wait()
        assert L.dsa_batch_host_output(b._h, 0) == p
        ref = oracle.decode(s)
        for i in range(2):
            v = b.host_views(i)
            assert np.array_equal(v["faces"], ref.faces)
            assert v["attributes"][0]["values"].tobytes() == ref.attributes[0].values.tobytes()
    finally:
        b.close()
        L.dsa_host_free(p)
