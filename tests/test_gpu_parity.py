"""Parity of the HIP decode path (through the C-ABI) with the CPU oracle on the same bytes:
connectivity, traversal order, integer attributes and point maps bit-exact; dequantised
floats bit-equal (the north star allows 1 ulp)."""
import numpy as np
import pytest

import oracle
import draco_sharp_amd as dsa
import draco_sharp_amd.synth as synth

pytestmark = pytest.mark.gpu

KINDS = [(synth.GRID, 9, 7), (synth.TORUS, 8, 6), (synth.SPHERE, 8, 7), (synth.HOLES, 20, 16), (synth.TWO_PARTS, 9, 6),
         (synth.GRID, 40, 33), (synth.TORUS, 24, 40)]


@pytest.fixture(scope="module")
def ctx():
    c = dsa.Context(0)
    yield c
    c.close()


def assert_same(got, ref, batch=None, i=None):
    """got: dsa.Draco from the GPU, ref: oracle.OracleMesh."""
    m = got.ConnectedData
    assert (got.Header.MajorVersion, got.Header.MinorVersion, got.Header.EncoderType, got.Header.EncoderMethod,
            got.Header.Flags) == (ref.major, ref.minor, ref.encoder_type, ref.encoder_method, ref.flags)
    assert m.FacesCount == ref.num_faces and m.PointsCount == ref.num_points
    assert np.array_equal(m.Faces, ref.faces)
    assert_same_attributes(m, ref)
    if batch is not None:
        nf = ref.num_faces
        assert np.array_equal(batch.debug_array(i, 0, np.uint32, 3 * nf), ref.opposite)
        assert np.array_equal(batch.debug_array(i, 1, np.uint32, 3 * nf), ref.corner_to_vertex)
        assert np.array_equal(batch.debug_array(i, 2, np.uint32, ref.num_vertices), ref.decoders[0]["data_to_corner"])


def assert_same_attributes(m, ref):
    assert len(m.Attributes) == len(ref.attributes)
    for a, r in zip(m.Attributes, ref.attributes):
        assert (a.AttributeType, a.DataType, a.NumComponents, a.UniqueId, a.DecoderType) == \
               (r.att_type, r.data_type, r.num_components, r.unique_id, r.seq_type)
        assert a.UniqueEntriesCount == r.num_entries
        # the oracle reports an identity mapping (PointAttribute.IsMappingIdentity) as an empty map
        assert np.array_equal(a.PointMap, r.point_map if len(r.point_map) else np.arange(m.PointsCount, dtype=np.uint32))
        if r.portable is not None:
            assert np.array_equal(a.PortableValues, r.portable)
        if a.Values.dtype == np.float32:
            assert np.array_equal(a.Values.view(np.uint32), r.values.view(np.uint32))   # bit-equal (<= 1 ulp required)
        else:
            assert np.array_equal(a.Values, r.values)


def run_batch(ctx, streams):
    b = dsa.Batch(ctx, streams)
    b.decode()
    return b


@pytest.mark.parametrize("single", [0, 1])
def test_topologies_match_oracle(ctx, single):
    streams = []
    for kind, nx, ny in KINDS:
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 3)
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(single_connectivity=single)))
    b = run_batch(ctx, streams)
    for i, s in enumerate(streams):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        assert_same(b.result(i), oracle.decode(s), b, i)
    b.close()


@pytest.mark.parametrize("scheme", [0, 1])
@pytest.mark.parametrize("pred", [0, 1])
def test_symbol_schemes_and_predictions(ctx, scheme, pred):
    streams = []
    for seed in range(4):
        pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 14 + seed, 12, 10 + seed)
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=scheme, pos_prediction=pred,
                                                                                uv_prediction=pred)))
    b = run_batch(ctx, streams)
    for i, s in enumerate(streams):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        assert_same(b.result(i), oracle.decode(s), b, i)
    b.close()


@pytest.mark.parametrize("bits", [(8, 8, 4), (14, 12, 10), (16, 16, 12), (20, 18, 14)])
def test_bit_depths(ctx, bits):
    # deeper quantisation -> wider alphabets: exercises the multi-block table search and the large-alphabet path
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 48, 40, 5)
    s = synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_bits=bits[0], uv_bits=bits[1], normal_bits=bits[2]))
    b = run_batch(ctx, [s])
    assert b.status(0) == 0, b.mesh_info(0).detail
    assert_same(b.result(0), oracle.decode(s), b, 0)
    b.close()


def test_generic_and_positions_only(ctx):
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 9, 9, 8)
    gen = (np.arange(len(pos)) % 251).astype(np.uint8)
    streams = [synth.encode_mesh(pos, faces, None, None, generic=gen), synth.encode_mesh(pos, faces)]
    b = run_batch(ctx, streams)
    for i, s in enumerate(streams):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        assert_same(b.result(i), oracle.decode(s), b, i)
    b.close()


def test_64k_triangle_meshes(ctx):
    # BASELINE.json config 2: one 64k-triangle Edgebreaker mesh, bit-exact (grid and torus topologies)
    streams = []
    for kind in (synth.GRID, synth.TORUS):
        pos, nrm, uv, faces = synth.make_mesh(kind, 128, 256, 2)
        streams.append(synth.encode_mesh(pos, faces, nrm, uv))
    b = run_batch(ctx, streams)
    for i, s in enumerate(streams):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        ref = oracle.decode(s)
        assert ref.num_faces == 65536
        assert_same(b.result(i), ref, b, i)
    b.close()


def test_two_batches_in_flight(ctx):
    """dsa_batch_decode is asynchronous: two batches enqueued back to back on one context, waited afterwards."""
    sets = []
    for seed in (1, 2):
        streams = []
        for kind, nx, ny in KINDS[:4]:
            pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, seed)
            streams.append(synth.encode_mesh(pos, faces, nrm, uv))
        sets.append(streams)
    b1, b2 = dsa.Batch(ctx, sets[0]), dsa.Batch(ctx, sets[1])
    b1.decode(wait=False)
    b2.decode(wait=False)
    b2.wait()
    b1.wait()
    for b, streams in ((b1, sets[0]), (b2, sets[1])):
        for i, sbytes in enumerate(streams):
            assert_same(b.result(i), oracle.decode(sbytes), b, i)
        b.close()


def test_many_holes_and_components(ctx):
    """Thousands of boundary loops (one topology-split event each) and of connected components: the split-corner
    dictionary and the traversal restarts must stay linear (a linear search there once cost seconds per mesh)."""
    import time
    pos, nrm, uv, faces = synth.make_mesh(synth.HOLES, 300, 250, 5)
    big = synth.encode_mesh(pos, faces, nrm, uv)
    b = run_batch(ctx, [big])
    t0 = time.perf_counter()
    b.decode()
    dt = time.perf_counter() - t0
    assert b.status(0) == 0, b.mesh_info(0).detail
    ref = oracle.decode(big)
    assert ref.num_faces > 100000
    assert_same(b.result(0), ref, b, 0)
    assert dt < 2.0, dt
    b.close()


def test_bad_streams_do_not_poison_the_batch(ctx):
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 9, 7, 3)
    good = synth.encode_mesh(pos, faces, nrm, uv)
    streams = [good, b"", b"DRACX" + good[5:], good[:40], good[: len(good) // 2], good[:-1], good]
    v1 = bytearray(good); v1[5] = 1
    streams.append(bytes(v1))
    kd = bytearray(synth.encode_point_cloud(pos)); kd[8] = 1
    streams.append(bytes(kd))             # kd-tree point cloud: not on the device path
    b = run_batch(ctx, streams)
    ref = oracle.decode(good)
    for i in (0, 6):
        assert b.status(i) == 0
        assert_same(b.result(i), ref, b, i)
    for i in (1, 2, 3, 4, 5, 7):
        assert b.status(i) == 1, (i, b.status(i))
        with pytest.raises(dsa.InvalidDataException):
            b.result(i)
    assert b.status(8) == 2
    with pytest.raises(NotImplementedError):
        b.result(8)
    b.close()


def test_point_cloud_sequential(ctx):
    """BASELINE.json configs[0]: 1k-point cloud, quantised positions, sequential decoder -- plus a larger one and
    both symbol schemes.  Entry i is point i (linear sequencer); no faces."""
    rng = np.random.default_rng(1)
    streams = []
    for n, bits, scheme in ((1000, 11, -1), (1, 11, -1), (100000, 14, 1), (5000, 9, 0)):
        pos = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        streams.append(synth.encode_point_cloud(pos, synth.options(pos_bits=bits, force_scheme=scheme)))
    b = run_batch(ctx, streams)
    for i, sbytes in enumerate(streams):
        ref = oracle.decode(sbytes)
        got = b.result(i)
        assert ref.encoder_type == 0 and type(got.ConnectedData) is dsa.PointCloud
        assert got.ConnectedData.PointsCount == ref.num_points
        assert_same_attributes(got.ConnectedData, ref)
        assert np.array_equal(got.ConnectedData.Attributes[0].PointMap, np.arange(ref.num_points, dtype=np.uint32))
    b.close()


def test_sequential_meshes(ctx):
    """MeshSequentialDecoder streams (faces as point indices, linear attribute order): raw index widths u8 / u16 /
    varint and the compressed form; both symbol schemes."""
    streams = []
    for kind, nx, ny in ((synth.GRID, 9, 7), (synth.TORUS, 20, 16), (synth.GRID, 300, 250)):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 3)
        for compressed in (True, False):
            streams.append(synth.encode_mesh_sequential(pos, faces, nrm, uv, compressed=compressed))
    pos, nrm, uv, faces = synth.make_mesh(synth.SPHERE, 8, 7, 3)
    streams.append(synth.encode_mesh_sequential(pos, faces, None, uv, compressed=True, opt=synth.options(force_scheme=0)))
    b = run_batch(ctx, streams)
    for i, sbytes in enumerate(streams):
        ref = oracle.decode(sbytes)
        assert b.status(i) == 0, (i, b.status(i), b.mesh_info(i).detail)
        got = b.result(i)
        assert got.Header.EncoderMethod == 0 and np.array_equal(got.ConnectedData.Faces, ref.faces)
        assert got.ConnectedData.PointsCount == ref.num_points
        assert_same_attributes(got.ConnectedData, ref)
    b.close()


def _corruptions(data, count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        m = bytearray(data)
        kind = int(rng.integers(0, 4))
        if kind == 0:
            for _ in range(int(rng.integers(1, 5))):
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            for _ in range(int(rng.integers(1, 9))):
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        elif kind == 2:
            m = m[: int(rng.integers(11, len(m)))]
        else:
            at = int(rng.integers(11, len(m)))
            for q in range(int(rng.integers(1, 17))):
                if at + q < len(m):
                    m[at + q] = int(rng.integers(0, 256))
        out.append(bytes(m))
    return out


def test_corrupt_streams_agree_with_the_oracle(ctx, house04_bytes):
    """Random corruptions of a fast-path mesh and of house_04 in one batch: where the oracle decodes, the GPU
    must decode the same; where the oracle rejects, the GPU must not report success with different data.
    (Both sides may legitimately accept a damaged payload: then they have to agree on the result.)"""
    pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 10, 8, 3)
    base = synth.encode_mesh(pos, faces, nrm, uv)
    # the stock-encoder mix (valence symbols, TexCoordsPortable, GeometricNormal, prediction-degree order) and the
    # rarer one (predictive symbols, both multi-parallelogram schemes): all general-path code
    stock = synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(predictive_connectivity=2, uv_prediction=5, normal_prediction=6, traversal_method=1))
    rare = synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(predictive_connectivity=1, pos_prediction=4, uv_prediction=2, single_connectivity=1))
    # GeometricNormal on the fast kernels (the flip-bit block, the corner fans) with valence-coded connectivity
    geo = synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(predictive_connectivity=2, normal_prediction=6))
    # tagged symbol streams of a mesh large enough for k_tags (the tag stream on a wave of its own, the walk resumed behind it)
    bpos, bnrm, buv, bfaces = synth.make_mesh(synth.GRID, 128, 128, 4)
    big_tagged = synth.encode_mesh(bpos, bfaces, bnrm, buv, opt=synth.options(force_scheme=0))
    # attribute seams on a torus: two seamed attributes with different seams (seam bits, attribute corner tables, corner attributes)
    from meshutil import seamed_mesh
    seamed = synth.encode_mesh_corners(*seamed_mesh(synth, synth.TORUS, 10, 8, 3, "checker", "stripes"), opt=synth.options(uv_prediction=5))
    streams = (_corruptions(base, 96, 5) + _corruptions(house04_bytes, 64, 6) + _corruptions(stock, 64, 7) + _corruptions(rare, 64, 8) +
               _corruptions(geo, 96, 9) + _corruptions(big_tagged, 64, 10) + _corruptions(seamed, 96, 11) +
               [base, house04_bytes, stock, rare, geo, big_tagged, seamed])
    b = run_batch(ctx, streams)
    agree_ok = agree_bad = gpu_stricter = 0
    stricter_sites = {}
    for i, sbytes in enumerate(streams):
        try:
            ref = oracle.decode(sbytes)
        except oracle.OracleError:
            ref = None
        st = b.status(i)
        if ref is not None and st == 0:
            assert_same(b.result(i), ref)
            agree_ok += 1
        elif ref is None:
            assert st != 0, (i, "GPU accepted a stream the oracle rejects")
            agree_bad += 1
        else:
            gpu_stricter += 1          # a check of the device path that the oracle does not make
            info = b.mesh_info(i)
            stricter_sites[(info.status, info.detail)] = stricter_sites.get((info.status, info.detail), 0) + 1
    assert agree_ok >= 4 and agree_bad > 0
    # Where the device path refuses what the oracle lets through, it is one of its own validations, by site: 123 (a traversal
    # method byte above 1, which the oracle reads as depth-first), 263 (the census of linked corners) and 681 (the general path's
    # bound on attribute seam data).  Four seeds of this mix gave 0 - 1 such streams of 292 (sites 123, 681); anything else, or
    # more than a handful, is a difference to look at.  (454 streams with the GeometricNormal and the large tagged family.)
    assert set(stricter_sites) <= {(1, 123), (1, 263), (1, 681), (1, 668)}, stricter_sites      # 668: more orientation bits than entries
    assert gpu_stricter <= 4, (agree_ok, agree_bad, stricter_sites)
    b.close()


def test_reference_sample_house04(ctx, house04_bytes):
    """The reference's own sample (valence traversal, 59 topology splits, UV seams, TexCoordsPortable) through the
    C-ABI: equal to the oracle, which tests/test_oracle_golden.py pins on house_04.obj."""
    good = synth.encode_mesh(*_small_mesh())
    b = run_batch(ctx, [house04_bytes, good, house04_bytes])
    ref = oracle.decode(house04_bytes)
    for i in (0, 2):
        assert b.status(i) == 0, (b.status(i), b.mesh_info(i).detail)
        assert_same(b.result(i), ref, b, i)
    assert_same(b.result(1), oracle.decode(good), b, 1)      # a fast-path mesh in the same batch
    # and directly against the fixture's ground truth (house_04.obj), without the oracle in between:
    # every decoded face corner must carry the quantised (position, uv) of the matching .obj corner
    import os
    from meshutil import face_multiset, quantize
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "house_04_expected.npz"))
    m = b.result(0).ConnectedData
    pos, uv = m.Attributes[0], m.Attributes[1]
    assert (m.FacesCount, pos.UniqueEntriesCount, pos.QuantizationBits, uv.QuantizationBits) == (2588, 1775, 11, 10)
    qv = quantize(g["v"], pos.MinValues[:3], pos.Range, pos.QuantizationBits)
    qvt = quantize(g["vt"], uv.MinValues[:2], uv.Range, uv.QuantizationBits)
    exp_keys = np.concatenate([qv[g["fv"].ravel()], qvt[g["fvt"].ravel()]], axis=1)
    exp = face_multiset(np.arange(exp_keys.shape[0]).reshape(-1, 3), exp_keys)
    got_keys = np.concatenate([pos.PortableValues[pos.PointMap], uv.PortableValues[uv.PointMap]], axis=1)
    assert face_multiset(m.Faces, got_keys) == exp
    b.close()


def _small_mesh():
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 9, 7, 3)
    return pos, faces, nrm, uv


def test_general_path_on_every_synthetic_case(ctx, monkeypatch):
    """DSA_FORCE_GENERAL routes every Edgebreaker mesh through k_general (dsa_general.h): the serial restatement
    must agree with the oracle on all topologies, both symbol schemes, all predictions and bit depths."""
    monkeypatch.setenv("DSA_FORCE_GENERAL", "1")
    streams = []
    for kind, nx, ny in KINDS:
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 11)
        for single in (0, 1):
            streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(single_connectivity=single)))
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 12, 9, 5)
    for scheme in (0, 1):
        for pred in (0, 1):
            streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=scheme, pos_prediction=pred, uv_prediction=pred)))
    for bits in (4, 16, 20):
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_bits=bits, uv_bits=min(bits, 16), normal_bits=min(bits, 12))))
    b = run_batch(ctx, streams)
    for i, sbytes in enumerate(streams):
        assert b.status(i) == 0, (i, b.status(i), b.mesh_info(i).detail)
        assert_same(b.result(i), oracle.decode(sbytes), b, i)
        assert b.debug_array(i, 4, np.uint32, 12)[6] == 0       # no k_traverse clock: the fast kernels skipped the mesh
    b.close()
    monkeypatch.delenv("DSA_FORCE_GENERAL")
    b = run_batch(ctx, streams[:2])
    assert b.debug_array(0, 4, np.uint32, 12)[6] != 0            # and without the switch they take it
    b.close()


RARE = [dict(normal_transform=2), dict(raw_integers=4), dict(raw_integers=2, pos_bits=12, uv_bits=10, normal_bits=8),
        dict(raw_integers=1, pos_bits=6, uv_bits=6, normal_bits=5), dict(no_prediction=1), dict(no_prediction=2), dict(no_prediction=4),
        dict(no_prediction=7, raw_integers=2, pos_bits=10, uv_bits=10, normal_bits=7), dict(normal_transform=2, raw_integers=4, no_prediction=3),
        dict(normal_transform=2, force_scheme=0), dict(no_prediction=7, force_scheme=0, single_connectivity=1)]


@pytest.mark.parametrize("general", [False, True])
def test_rare_decoder_branches(ctx, monkeypatch, general):
    """The non-canonicalised octahedral transform (PredictionSchemeNormalOctahedronDecodingTransform.cs:47-76), integers
    stored uncompressed at 1 / 2 / 4 bytes (SequentialIntegerAttributeDecoder.cs:68-84) and prediction method -2, on the
    fast kernels and on the general path."""
    if general:
        monkeypatch.setenv("DSA_FORCE_GENERAL", "1")
    streams = []
    for kind, nx, ny in ((synth.GRID, 14, 11), (synth.TORUS, 10, 8), (synth.HOLES, 14, 12), (synth.GRID, 70, 50)):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 21)
        for opt in RARE:
            streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt)))
    b = run_batch(ctx, streams)
    for i, sbytes in enumerate(streams):
        assert b.status(i) == 0, (i, b.status(i), b.mesh_info(i).detail)
        assert_same(b.result(i), oracle.decode(sbytes), b, i)
        assert (b.debug_array(i, 4, np.uint32, 12)[6] == 0) == general
    b.close()


def test_many_large_alphabets_in_one_batch(ctx):
    """2048 small meshes whose streams carry alphabets beyond the LDS search (> 4032 symbols): the cumulative tables
    live in the attributes' own regions, or in the batch pool, or -- when that is spent -- the mesh is decoded again by
    the general path; the verdict of a stream never depends on the rest of the batch."""
    rng = np.random.default_rng(3)
    base = []
    for k in range(8):
        pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 40, 40, 50 + k)
        pos = pos + rng.normal(0, 0.05, pos.shape).astype(np.float32)
        base.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_bits=16, uv_bits=16, normal_bits=8, force_scheme=1)))
    refs = [oracle.decode(s) for s in base]
    streams = [base[i % 8] for i in range(2048)]
    b = run_batch(ctx, streams)
    assert all(int(r.attributes[0].symbols.max()) + 1 > 4032 for r in refs)      # the position alphabets are large ones
    bad = [(i, b.status(i), b.mesh_info(i).detail) for i in range(len(streams)) if b.status(i) != 0]
    assert not bad, bad[:5]
    for i in list(range(0, 2048, 97)) + [2047]:
        assert_same(b.result(i), refs[i % 8])
    b.close()


def test_schemes_behind_the_symbol_streams_take_the_second_chance(ctx):
    """Which prediction scheme an attribute uses stands behind its symbol stream, where the host parse does not go.
    GeometricNormal (method 6) and TexCoordsPortable (method 5) run on the fast kernels (k_flip_bits + k_predict_geometric,
    k_orient_bits + k_texcoords_prepare + k_texcoords); the multi-parallelogram schemes need the general path's tables: k_locate
    hands such meshes back (DSA_SITE_RETRY_GENERAL) and dsa_batch_wait decodes them again through k_general, next to meshes that
    stay on the fast kernels."""
    streams, geo = [], []
    for k, (kind, nx, ny) in enumerate(KINDS):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 21)
        g = k % 3 != 1
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(normal_prediction=6 if g else 0, single_connectivity=k & 1,
                                                                                  pos_bits=11 + k, normal_bits=8 + (k % 5))))
        geo.append(False)
    # TexCoordsPortable stays on the fast kernels too (round 4)
    pos, nrm, uv, faces = synth.make_mesh(synth.HOLES, 20, 16, 22)
    for npred in (0, 6):
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(uv_prediction=5, normal_prediction=npred, force_scheme=1)))
        geo.append(False)
    # the plain multi-parallelogram scheme goes to the general path, and the constrained one where the first attribute does not show it
    for ppred, upred in ((2, 1), (1, 4), (1, 2)):
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=ppred, uv_prediction=upred)))
        geo.append(True)
    b = run_batch(ctx, streams)
    for i, sbytes in enumerate(streams):
        assert b.status(i) == 0, (i, b.status(i), b.mesh_info(i).detail)
        ref = oracle.decode(sbytes)
        assert any(x.pred_method in (2, 4) for x in ref.attributes) == geo[i]
        assert_same(b.result(i), ref, b, i)
        assert (b.debug_array(i, 4, np.uint32, 12)[6] == 0) == geo[i]      # decoded by k_general / by the fast kernels
    # decoding the same batch again rebuilds the second-chance batch
    b.decode()
    for i in (0, 1, len(streams) - 1):
        assert_same(b.result(i), oracle.decode(streams[i]))
    b.close()
    # a geometric-normal stream and the difference-coded stream of the same mesh decode to the same normals
    pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 24, 40, 2)
    b = run_batch(ctx, [synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(normal_prediction=p)) for p in (0, 6)])
    n0, n6 = (b.result(i).ConnectedData.Attributes[1] for i in (0, 1))
    assert n6.PredictionMethod == 6 if hasattr(n6, "PredictionMethod") else True
    assert np.array_equal(n0.PortableValues, n6.PortableValues) and np.array_equal(n0.Values.view(np.uint32), n6.Values.view(np.uint32))
    b.close()


def test_prediction_degree_traversal(ctx):
    """MeshTraversalMethod 1 (MaxPredictionDegreeTraverser) is visible to the host parse in the decoder triples: such
    meshes get the general scratch up front and are decoded by k_general*, beside fast-path meshes."""
    streams = []
    for k, (kind, nx, ny) in enumerate(KINDS):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 31)
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(traversal_method=k % 3, single_connectivity=(k >> 1) & 1,
                                                                                  pos_prediction=(1, 4, 2)[k % 3], uv_prediction=(1, 5)[k & 1],
                                                                                  predictive_connectivity=(1, 0, 2, 2, 1, 1, 2)[k])))
    b = run_batch(ctx, streams)
    for i, sbytes in enumerate(streams):
        assert b.status(i) == 0, (i, b.status(i), b.mesh_info(i).detail)
        ref = oracle.decode(sbytes)
        assert ref.decoders[0]["traversal_method"] == (1 if i % 3 else 0)
        assert ref.traversal_type == (1, 0, 2, 2, 1, 1, 2)[i]    # predictive Edgebreaker symbols: general path as well (valence ones: with the prediction-degree order)
        assert_same(b.result(i), ref, b, i)
    b.close()


def test_single_decode_api(ctx):
    pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 12, 9, 4)
    s = synth.encode_mesh(pos, faces, nrm, uv)
    got = dsa.DracoDecoder(ctx).Decode(s)
    assert_same(got, oracle.decode(s))
    mesh = got.ConnectedData
    p = mesh.GetNamedAttribute(0)
    assert p.MappedIndex(mesh.GetFace(0)[0]) == int(p.PointMap[mesh.Faces[0, 0]])
    with pytest.raises(dsa.InvalidDataException):
        dsa.DracoDecoder(ctx).Decode(s[:100])


def test_batch_properties_full_size(ctx):
    # size-independent properties on a batch of 64k-triangle meshes: every face references valid points,
    # every point map entry is in range, unit normals, positions inside the quantisation box.
    blob, offsets = synth.make_batch(synth.GRID, 128, 256, 1000, 16)
    b = dsa.Batch(ctx, blob=blob, offsets=offsets)
    b.decode()
    for i in range(b.n):
        assert b.status(i) == 0
    for i in (0, 7, 15):
        d = b.result(i)
        m = d.ConnectedData
        assert m.FacesCount == 65536 and m.PointsCount == 33153
        assert m.Faces.min() >= 0 and m.Faces.max() == m.PointsCount - 1
        pos, nrm, uv = m.Attributes
        for a in m.Attributes:
            assert a.PointMap.max() < a.UniqueEntriesCount
            assert np.array_equal(np.sort(np.unique(a.PointMap)), np.arange(a.UniqueEntriesCount))
        assert np.allclose(np.linalg.norm(nrm.Values, axis=1), 1.0, atol=1e-5)
        lo = np.array(pos.MinValues, np.float32)
        assert np.all(pos.Values >= lo - 1e-6) and np.all(pos.Values <= lo + pos.Range * 1.0001)
        s = bytes(blob[int(offsets[i]):int(offsets[i + 1])])
        assert_same(d, oracle.decode(s), b, i)
    assert b.algorithmic_bytes > 16 * (65536 * 12 + 33153 * 32)
    b.close()


def test_randomised_options_agree_with_the_oracle(ctx):
    """tools/soak.py: random topology / size / bit depths / symbol scheme / prediction schemes / attribute order /
    Edgebreaker symbol coding / connectivity mode per mesh, one batch, every result equal to the oracle's."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import soak
    assert soak.run(120, 7, ctx) == []


def test_many_attributes(ctx):
    """Up to 16 attributes per stream decode on the device path; more is DSA_ERR_NOT_IMPLEMENTED, not invalid data."""
    from meshutil import raw_point_cloud_stream
    kinds = [(0, 9, 3), (1, 9, 3), (2, 2, 4), (3, 9, 2), (3, 9, 2), (4, 4, 4), (4, 9, 4), (4, 6, 1), (4, 1, 2), (4, 3, 3), (2, 2, 3), (4, 5, 1)]
    ok12, vals12 = raw_point_cloud_stream(333, kinds, 1)
    ok16, vals16 = raw_point_cloud_stream(100, (kinds * 2)[:16], 2)
    too_many, _ = raw_point_cloud_stream(50, (kinds * 2)[:17], 3)
    b = run_batch(ctx, [ok12, too_many, ok16])
    assert b.status(1) == 2                                    # DSA_ERR_NOT_IMPLEMENTED
    with pytest.raises(NotImplementedError):
        b.result(1)
    for i, (sbytes, vals) in ((0, (ok12, vals12)), (2, (ok16, vals16))):
        assert b.status(i) == 0, (i, b.status(i), b.mesh_info(i).detail)
        ref = oracle.decode(sbytes)
        d = b.result(i)
        assert isinstance(d.ConnectedData, dsa.PointCloud) and len(d.ConnectedData.Attributes) == len(vals)
        assert_same_attributes(d.ConnectedData, ref)
        for a, v in zip(d.ConnectedData.Attributes, vals):
            assert a.Values.dtype == v.dtype and np.array_equal(a.Values, v)
    b.close()


def test_oversized_claims_are_set_aside_not_the_batch(ctx):
    """Headers may claim far more elements than their bytes can carry (the sizing parse allows 1024 per byte, rANS can
    go that low).  When the arena of a batch does not fit the device, the largest claimants get a per-mesh
    DSA_ERR_OUT_OF_MEMORY and the rest of the batch decodes."""
    from meshutil import _varint
    nf, nv = 600_000_000, 300_000_000
    head = (b"DRACO" + bytes([2, 2, 1, 1, 0, 0]) + bytes([0]) + _varint(nv) + _varint(nf) + bytes([0]) + _varint(nf) + _varint(0) + _varint(0) +
            _varint(0) + bytes([128]) + _varint(0) +                      # no symbol bytes, empty start-face block
            bytes([1, 0xFF, 0, 0]) + _varint(1) + bytes([0, 9, 3, 0]) + _varint(0) + bytes([2]))
    hostile = head + bytes(1 << 20)                                        # 1 MiB of padding makes the claim admissible
    pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 10, 8, 3)
    good = synth.encode_mesh(pos, faces, nrm, uv)
    streams = [good] + [hostile] * 8 + [good]
    b = run_batch(ctx, streams)
    st = [b.status(i) for i in range(len(streams))]
    assert st[0] == 0 and st[-1] == 0
    assert_same(b.result(0), oracle.decode(good))
    assert set(st[1:-1]) <= {1, 5} and 5 in st[1:-1], st                   # set aside (5) or rejected by the device parse (1)
    with pytest.raises(MemoryError):
        b.result(1 + st[1:-1].index(5))
    assert b.arena_bytes < 288 * (1 << 30)
    b.close()


def test_device_views_are_the_results_without_a_copy(ctx):
    """dsa_batch_device_faces / _attribute_values / _point_map: pointers into the batch arena for consumers that stay on
    the GPU (a renderer, a torch pipeline).  Wrapped as torch tensors they must hold what the copy calls return."""
    import torch
    streams = []
    for kind, nx, ny, opt in ((synth.TORUS, 12, 9, {}), (synth.HOLES, 20, 16, {"predictive_connectivity": 2, "normal_prediction": 6})):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 17)
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt)))
    b = run_batch(ctx, streams)
    for i in range(len(streams)):
        m = b.result(i).ConnectedData
        dv = b.device_views(i)
        assert dv["faces"].is_cuda and dv["faces"].dtype == torch.int32
        assert np.array_equal(dv["faces"].cpu().numpy(), m.Faces)
        for a, d in zip(m.Attributes, dv["attributes"]):
            assert np.array_equal(d["values"].cpu().numpy().view(a.Values.dtype), a.Values)
            assert np.array_equal(d["point_map"].cpu().numpy().view(np.uint32), a.PointMap)
        # positions per point, gathered on the device
        p = dv["attributes"][0]
        per_point = p["values"][p["point_map"].long()]
        assert np.array_equal(per_point.cpu().numpy(), m.Attributes[0].Values[m.Attributes[0].PointMap])
    b.close()


def test_high_precision_symbol_streams(ctx):
    """rANS precisions above 12 bits: alphabets of up to 2048 symbols (k_symbols_wide, table in registers), sparse large
    alphabets searched through their non-zero symbols (14-bit positions), and what is left to the LDS tiers -- every
    compression level's precision rule, one batch, against the oracle."""
    cases = []
    for k, opts in enumerate(({"pos_bits": 12, "uv_bits": 12, "normal_bits": 10}, {"pos_bits": 13, "compression_level": 0}, {"pos_bits": 14, "compression_level": 10},
                              {"pos_bits": 14, "uv_bits": 13, "normal_bits": 12}, {"pos_bits": 16, "uv_bits": 15}, {"pos_bits": 12, "compression_level": 7},
                              {"pos_bits": 20, "uv_bits": 16, "normal_bits": 14})):
        pos, nrm, uv, faces = synth.make_mesh(synth.GRID if k % 2 == 0 else synth.TORUS, 96, 80, 40 + k)
        cases.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=1, **opts)))
    # long streams at 16-bit precision: a remainder of 65535 is as large as the low half of the register table's padding word
    for k, opts in enumerate(({"pos_bits": 16, "uv_bits": 14, "compression_level": 8}, {"pos_bits": 15, "uv_bits": 13, "normal_bits": 12, "compression_level": 6},
                              {"pos_bits": 16, "uv_bits": 12, "compression_level": 7})):
        pos, nrm, uv, faces = synth.make_mesh(synth.GRID if k != 1 else synth.HOLES, 200, 150, 70 + k)
        cases.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=1, **opts)))
    b = dsa.Batch(ctx, cases)
    b.decode()
    seen = set()
    for i, data in enumerate(cases):
        assert b.status(i) == 0, b.mesh_info(i).detail
        assert_same(b.result(i), oracle.decode(data), b, i)
        info = b.debug_array(i, 5, np.uint32, 64).reshape(16, 4)[:3]
        seen.update(int(p) for src, _, p, _ in info if src == 1)
    assert 16 in seen and len(seen) >= 3               # several precisions, the table's widest among them, were really met
    b.close()


def _pad_raw_tables(data, mbl, extra):
    """Every raw symbol section of `data` whose max_bit_length byte is `mbl`: `extra` (<= 64) more symbols of zero frequency appended
    to its probability table (one zero-run token, RAnsSymbolDecoder.cs:28-37) -- a table a stock encoder never writes (it trims
    trailing zeros) and the reference's slot table decodes like the trimmed one.  Sections are found by a validating scan: scheme byte 1,
    the length byte, a table whose frequencies sum to the precision."""
    precision = 1 << max(12, min(20, 3 * mbl // 2))
    out, at, done = bytearray(), 0, 0
    o = 0
    while o + 3 < len(data):
        if data[o] == 1 and data[o + 1] == mbl:
            q, n, sh = o + 2, 0, 0
            while q < len(data):
                n |= (data[q] & 0x7F) << sh; sh += 7; q += 1
                if not data[q - 1] & 0x80: break
            total, i, ok, tq = 0, 0, 64 < n <= 2048 - extra, q
            while ok and i < n and tq < len(data):
                tok = data[tq] & 3
                if tok == 3: i += (data[tq] >> 2) + 1; tq += 1
                else:
                    pr = data[tq] >> 2
                    for k in range(tok): pr |= data[tq + 1 + k] << (8 * (k + 1) - 2)
                    total += pr; i += 1; tq += 1 + tok
            if ok and i == n and total == precision:
                m, var = n + extra, bytearray()
                while True:
                    var.append((m & 0x7F) | (0x80 if m > 0x7F else 0)); m >>= 7
                    if not m: break
                out += data[at:o + 2] + var + data[q:tq] + bytes([((extra - 1) << 2) | 3])
                at, o, done = tq, tq, done + 1
                continue
        o += 1
    return bytes(out + data[at:]), done


def test_trailing_zero_frequencies_at_16_bit_precision(ctx):
    """a non-compact table whose last symbols have zero frequency has cumulative frequency 2^16 there, which the
    16 + 16-bit register table of k_symbols_wide must treat as padding."""
    cases = []
    for k, extra in enumerate((1, 17, 64)):
        # 10-bit octahedral normals at compression level 8: 1023 symbols, max_bit_length 11 -> 16-bit precision, table not compacted
        pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 96, 80, 90)
        data = synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=1, pos_bits=11, uv_bits=10, normal_bits=10, compression_level=8))
        padded, done = _pad_raw_tables(data, 11, extra)
        assert done >= 1, "no 16-bit-precision raw section found to pad"
        want, got = oracle.decode(data), oracle.decode(padded)
        assert all(np.array_equal(x.portable, y.portable) for x, y in zip(want.attributes, got.attributes))   # same symbols, longer table
        cases.append(padded)
    b = dsa.Batch(ctx, cases)
    b.decode()
    for i, data in enumerate(cases):
        assert b.status(i) == 0, b.mesh_info(i).detail
        assert_same(b.result(i), oracle.decode(data), b, i)
    b.close()


def test_valence_streams_take_the_fast_kernels(ctx):
    """Valence-coded Edgebreaker symbols (MeshEdgeBreakerTraversalValenceDecoder.cs:22-154, what stock encoders write for meshes
    of a thousand faces and more) on the wave-per-mesh kernels: the six context lists decoded by the connectivity wave, strip
    runs checked against them.  Every topology -- handles (topology splits), holes, several components --, both connectivity
    layouts, and a 64k-triangle mesh; connectivity, order and values against the oracle."""
    cases = []
    for k, (kind, nx, ny) in enumerate(KINDS):
        for single in (0, 1):
            pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 30 + k)
            cases.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(predictive_connectivity=2, single_connectivity=single)))
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 128, 256, 77)
    cases.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(predictive_connectivity=2)))
    pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 128, 256, 78)
    cases.append(synth.encode_mesh(pos, faces, None, None, opt=synth.options(predictive_connectivity=2, pos_bits=14)))
    b = run_batch(ctx, cases)
    fast = 0
    for i, data in enumerate(cases):
        ref = oracle.decode(data)
        assert ref.traversal_type == 2
        info = b.mesh_info(i)
        assert info.status == 0, (i, info.detail)
        # (a list the encoder wrote with the tagged scheme -- it does for the few symbols of a tiny mesh -- sends the mesh to the
        # general path at the second attempt; a mesh of the size stock encoders use valence symbols for must not go there)
        assert info.decode_path == 0 or (ref.num_faces < 1000 and info.decode_path == 2), "valence stream %d (%d faces) was sent to the general path" % (i, ref.num_faces)
        fast += info.decode_path == 0
        assert_same(b.result(i), ref, b, i)
    assert fast >= 6
    b.close()


def test_octahedral_delta_one_lane_per_stream(ctx, monkeypatch):
    """k_predict_oct_streams (what crowded batches use for the octahedral delta of normals): 64 streams of different lengths per
    wave -- odd and even entry counts, every topology (spheres and tori cover both halves of the octahedron and every
    quadrant), 2 - 14 bits -- equal to the oracle; then a crowded batch (3 648 meshes: the chain kernel and the batch-size
    rule itself) without the switch."""
    monkeypatch.setenv("DSA_OCT_STREAMS", "1")
    streams = []
    for j, (kind, nx, ny) in enumerate(KINDS + [(synth.SPHERE, 21, 17), (synth.TORUS, 13, 11), (synth.GRID, 3, 2), (synth.GRID, 2, 2)]):
        for bits in (2, 5, 8, 11, 14):
            pos, nrm, uv, faces = synth.make_mesh(kind, nx + (bits % 3), ny, 40 + j)
            streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(normal_bits=bits)))
    b = run_batch(ctx, streams)
    entries = set()
    for i, s in enumerate(streams):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        ref = oracle.decode(s)
        entries.add(ref.attributes[1].num_entries % 2)
        assert_same(b.result(i), ref, b, i)
    assert entries == {0, 1}
    b.close()
    monkeypatch.delenv("DSA_OCT_STREAMS")
    crowd = [streams[i % len(streams)] for i in range(3648)]
    b = run_batch(ctx, crowd)
    refs = [oracle.decode(s) for s in streams]
    for i in range(len(crowd)):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
    for i in list(range(0, 3648, 37)) + [3647]:
        assert_same(b.result(i), refs[i % len(streams)], b, i)
    b.close()


def test_geometric_normals_on_the_fast_kernels(ctx):
    """GeometricNormal prediction (what stock encoders pick for normals at their default level) without the general path:
    every topology (fans around boundary vertices, handles, several components), valence-coded connectivity as stock encoders
    write it, octahedra of 2 - 14 bits, positions by parallelogram or by difference (the early strand) -- equal to the oracle,
    decode_path 0; and a 64k-triangle mesh."""
    streams = []
    for k, (kind, nx, ny) in enumerate(KINDS + [(synth.SPHERE, 21, 17), (synth.HOLES, 33, 29), (synth.GRID, 128, 256)]):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 60 + k)
        for j, (bits, conn, ppred) in enumerate(((8, 0, 1), (10, 2, 1), (3, 0, 0), (14, 2, 0))):
            if kind == synth.GRID and nx == 128 and j > 1:
                continue
            with_uv = (k + j) % 2 == 0
            streams.append(synth.encode_mesh(pos, faces, nrm, uv if with_uv else None,
                                             opt=synth.options(normal_prediction=6, normal_bits=bits, predictive_connectivity=conn, pos_prediction=ppred,
                                                               single_connectivity=1)))
    b = run_batch(ctx, streams)
    fast = 0
    for i, s in enumerate(streams):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        ref = oracle.decode(s)
        assert any(x.pred_method == 6 for x in ref.attributes)
        assert_same(b.result(i), ref, b, i)
        info = b.mesh_info(i)
        # small valence-coded meshes may carry tagged context lists, which the general path decodes (decode_path 2)
        assert info.decode_path == 0 or (info.decode_path == 2 and ref.num_faces < 1000 and ref.traversal_type == 2), (i, info.decode_path)
        fast += info.decode_path == 0
    assert fast >= len(streams) * 2 // 3
    b.close()


def test_tag_streams_on_a_wave_of_their_own(ctx):
    """Tagged symbol streams of meshes large enough for k_tags (the register-table decoder on the tag stream, the walk taken up
    behind it by k_locate_resume, one round per attribute): every attribute tagged, some tagged and some raw in one mesh, a batch
    mixing both with small meshes whose tags the walk decodes itself, 9 - 16 bit attributes, valence symbols and GeometricNormal
    normals on top -- equal to the oracle."""
    streams = []
    for j, (kind, nx, ny) in enumerate(((synth.GRID, 128, 256), (synth.TORUS, 96, 128), (synth.HOLES, 70, 90), (synth.GRID, 9, 7), (synth.SPHERE, 40, 60))):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 70 + j)
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=0)))
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=0, pos_bits=14, uv_bits=16, normal_bits=9,
                                                                                predictive_connectivity=2 if j % 2 else 0, normal_prediction=6 if j % 3 == 0 else 0)))
        # the encoder's own choice at 16 bits: high-entropy attributes come out tagged, the others raw
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_bits=16, uv_bits=9)))
    b = run_batch(ctx, streams)
    tagged = 0
    for i, s in enumerate(streams):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        ref = oracle.decode(s)
        assert_same(b.result(i), ref, b, i)
        info = b.debug_array(i, 5, np.uint32, 64).reshape(16, 4)[:3]
        tagged += int((info[:, 0] == 0).sum())
    assert tagged >= 20          # source 0 = tagged
    b.close()


def test_decodes_alternate_between_the_two_stream_sets(ctx):
    """A context owns two sets of streams and gives every decode the next one: two batches decoded in turn without waiting in
    between (the sustained leg of bench.py), and the same batch decoded twice in a row (the second waits for the first), all end
    equal to the oracle."""
    sets = []
    for seed in (5, 6):
        streams = []
        for kind, nx, ny in KINDS:
            pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, seed)
            streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(predictive_connectivity=2 * (seed & 1))))
        sets.append(streams)
    pair = [dsa.Batch(ctx, sets[0]), dsa.Batch(ctx, sets[1])]
    for k in range(7):
        pair[k % 2].decode(wait=False)
    pair[0].decode(wait=False)           # the same batch again, on the other set: ordered behind its previous decode
    for b in pair:
        b.wait()
    for b, streams in zip(pair, sets):
        for i, s in enumerate(streams):
            assert b.status(i) == 0, (i, b.mesh_info(i).detail)
            assert_same(b.result(i), oracle.decode(s), b, i)
        b.close()


def test_the_schedule_checks_its_own_assumptions(ctx):
    """k_register_gate times the late symbol launch of a crowded batch by occupancy arithmetic on three kernels' register counts;
    the context checks that arithmetic against this build's counts when it is created and says what it found."""
    note = ctx.schedule_note()
    assert note.startswith("k_register_gate in use") or note.startswith("k_register_gate left out"), note
    assert "k_chain" in note and "k_symbols_reg" in note
    print(note)


def test_the_bench_batch_at_full_size(ctx):
    """BASELINE.json configs[2] as bench.py times it: 4096 x 65 536-triangle meshes in ONE batch.  Kernel selection depends on the
    batch size (k_chain above 2048 meshes, k_predict_oct_streams + the gated late symbols from 3584), so the kernels of the
    headline number are only exercised at this size: 130 meshes spread over the batch, with its first and its last, are compared
    with the oracle bit for bit, every mesh must have succeeded on the fast kernels, and the crowded-batch kernels must be the
    ones that ran."""
    n = 4096
    blob, offsets = synth.make_batch(synth.GRID, 128, 256, 1000, n, normals=True, uvs=True)
    ctx.set_profiling(True)
    try:
        b = dsa.Batch(ctx, blob=blob, offsets=offsets)
        b.decode()
        kernels = b.kernel_times()
    finally:
        ctx.set_profiling(False)
    assert kernels.get("k_chain", 0) > 0 and kernels.get("k_predict_oct_streams", 0) > 0 and "k_traverse" not in kernels, kernels
    assert all(b.status(i) == 0 for i in range(n))
    assert all(b.mesh_info(i).decode_path == 0 for i in range(0, n, 37))
    picks = sorted(set([0, n - 1] + [int(x) for x in np.linspace(0, n - 1, 128)]))
    for i in picks:
        s = bytes(blob[int(offsets[i]):int(offsets[i + 1])])
        ref = oracle.decode(s)
        assert ref.num_faces == 65536
        assert_same(b.result(i), ref)
    b.close()
    ctx.trim()          # 26 GB of arena go back before the next test


def k_scheme(i):
    return (-1, 0, 1)[i % 3]


@pytest.mark.gpu
def test_constrained_multi_parallelogram_positions_on_the_fast_kernels(ctx):
    """What stock encoders write at their two highest compression levels: positions by ConstrainedMultiParallelogram (method 4), with
    TexCoordsPortable / GeometricNormal / parallelogram beside it.  The host parse sees the scheme byte of the first attribute (the
    values of the first decoder start with it) and sets the records aside; k_crease_bits, k_multipara_prepare and k_multipara decode
    it: decode_path 0 on every topology, with standard and valence connectivity, every symbol scheme, one 64k-triangle mesh and a
    crowded batch; texture coordinates and a generic attribute with the scheme beside such positions as well -- and a mesh whose first
    attribute does not show the scheme still takes the general path for it."""
    streams = []
    for kind, nx, ny in KINDS + [(synth.HOLES, 40, 33), (synth.SPHERE, 30, 21)]:
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 31)
        for opt in (dict(), dict(uv_prediction=5, normal_prediction=6), dict(predictive_connectivity=2, uv_prediction=5), dict(force_scheme=0),
                    dict(pos_bits=14, uv_prediction=5, normal_prediction=6, predictive_connectivity=2), dict(raw_integers=4)):
            streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=4, **opt)))
        streams.append(synth.encode_mesh(pos, faces, None, None, opt=synth.options(pos_prediction=4)))
        # the scheme on the texture coordinates as well, and on a generic attribute (vertex colours of a scan): every attribute of the
        # position connectivity gets its records where the first one shows the scheme
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=4, uv_prediction=4)))
        streams.append(synth.encode_mesh(pos, faces, nrm, uv, generic=(np.arange(len(pos), dtype=np.int32) * 7919) % 251, opt=synth.options(pos_prediction=4, uv_prediction=4, force_scheme=k_scheme(len(streams)))))
        # four components (the RGBA colours of a scan) and two
        for gc in (4, 2):
            g = ((np.arange(len(pos) * gc, dtype=np.int64) * 7919) % 251).astype(np.uint8).reshape(-1, gc)
            streams.append(synth.encode_mesh(pos, faces, nrm, uv, generic=g, opt=synth.options(pos_prediction=4, uv_prediction=5, normal_prediction=6, generic_components=gc)))
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 128, 256, 5)
    streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=4, uv_prediction=5, normal_prediction=6, predictive_connectivity=2)))
    pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 128, 256, 6)
    streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=4)))
    b = run_batch(ctx, streams)
    for i, sbytes in enumerate(streams):
        assert b.status(i) == 0, (i, b.status(i), b.mesh_info(i).detail)
        ref = oracle.decode(sbytes)
        assert ref.attributes[0].pred_method == 4
        assert_same(b.result(i), ref, b, i)
        assert b.mesh_info(i).decode_path == 0, (i, b.mesh_info(i).decode_path)
    b.close()
    # crowded: the chain's four meshes to a wave, ragged sizes
    crowd = [streams[i % len(streams)] for i in range(300)]
    b = run_batch(ctx, crowd)
    for i in range(300):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
    for i in list(range(0, 300, 7)) + [299]:
        assert_same(b.result(i), oracle.decode(crowd[i]))
        assert b.mesh_info(i).decode_path == 0
    b.close()
    # the scheme on a later attribute of a mesh whose positions do not use it: the general path, as before
    pos, nrm, uv, faces = synth.make_mesh(synth.HOLES, 20, 16, 22)
    later = synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=1, uv_prediction=4))
    b = run_batch(ctx, [later, streams[0]])
    assert b.status(0) == 0 and b.mesh_info(0).decode_path != 0 and b.mesh_info(1).decode_path == 0
    assert_same(b.result(0), oracle.decode(later))
    assert_same(b.result(1), oracle.decode(streams[0]))
    b.close()


@pytest.mark.gpu
def test_constrained_multi_parallelogram_with_seams_and_corrupt_streams(ctx):
    """Positions by ConstrainedMultiParallelogram in meshes whose other attributes have seams (corner-attribute decoders), and damaged
    streams of the dialect: the device path and the oracle agree on which streams decode and on every value of those."""
    import random
    from meshutil import seamed_mesh
    cases = []
    for kind, nx, ny in ((synth.GRID, 24, 17), (synth.TORUS, 16, 12), (synth.HOLES, 20, 16), (synth.SPHERE, 12, 9)):
        for charts, opt in (((None, "stripes"), dict(uv_prediction=5)), (("checker", "island"), dict(uv_prediction=5, normal_prediction=6, predictive_connectivity=2)),
                            (("random", "random"), dict())):
            args = seamed_mesh(synth, kind, nx, ny, 19, *charts)
            cases.append(synth.encode_mesh_corners(*args, opt=synth.options(pos_prediction=4, **opt)))
    b = run_batch(ctx, cases)
    for i, sbytes in enumerate(cases):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        assert_same(b.result(i), oracle.decode(sbytes), b, i)
        assert b.mesh_info(i).decode_path == 0
    b.close()
    rng = random.Random(77)
    pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 10, 8, 3)
    base = [synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=4, uv_prediction=5, normal_prediction=6)), cases[0], cases[4]]
    damaged = []
    for k in range(240):
        d = bytearray(base[k % 3])
        mode = k % 4
        if mode == 0:
            for _ in range(rng.randint(1, 3)):
                d[rng.randrange(20, len(d))] ^= 1 << rng.randrange(8)
        elif mode == 1:
            at = rng.randrange(20, len(d)); d[at:at + rng.randint(1, 8)] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 8)))
        elif mode == 2:
            d = d[:rng.randrange(30, len(d))]
        else:
            at = rng.randrange(len(d) // 2, len(d)); d[at] = rng.randrange(256)
        damaged.append(bytes(d))
    b = run_batch(ctx, damaged)
    stricter = []
    for i, sbytes in enumerate(damaged):
        try:
            ref = oracle.decode(sbytes)
        except oracle.OracleError:
            ref = None
        if ref is None:
            assert b.status(i) != 0, i
        elif b.status(i) != 0:
            stricter.append((b.status(i), b.mesh_info(i).detail))
        else:
            assert_same(b.result(i), ref)
    # (the device path's own validations, as in the other corrupt-stream tests: counts that exceed what the mesh can hold)
    assert set(stricter) <= {(1, 263), (1, 305), (1, 668), (1, 657), (1, 681), (1, 673), (1, 676)}, stricter
    b.close()


@pytest.mark.gpu
def test_generic_attributes_of_one_to_four_components(ctx):
    """Per-vertex uint8 attributes of 1 - 4 components (vertex colours) through the symbol, prediction and output kernels, with the
    parallelogram scheme and with difference coding, tagged and raw symbols: equal to the oracle, on the wave-per-mesh kernels."""
    streams = []
    for kind, nx, ny in KINDS:
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 47)
        for gc in (1, 2, 3, 4):
            g = ((np.arange(len(pos) * gc, dtype=np.int64) * 104729 + 17 * gc) % 256).astype(np.uint8).reshape(-1, gc)
            for opt in (dict(), dict(force_scheme=0), dict(pos_prediction=0), dict(predictive_connectivity=2, uv_prediction=5, normal_prediction=6)):
                streams.append(synth.encode_mesh(pos, faces, nrm, uv, generic=g, opt=synth.options(generic_components=gc, **opt)))
    b = run_batch(ctx, streams)
    for i, sbytes in enumerate(streams):
        assert b.status(i) == 0, (i, b.mesh_info(i).detail)
        ref = oracle.decode(sbytes)
        assert ref.attributes[-1].att_type == 4
        assert_same(b.result(i), ref, b, i)
        assert b.mesh_info(i).decode_path == 0
    b.close()
