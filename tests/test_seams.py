"""Attribute seams: seam bits, attribute corner tables and corner attributes
(MeshEdgeBreakerDecoder.cs:502-638, MeshAttributeCornerTable.cs:19-155; writer side MeshEdgeBreakerEncoder.cs:403-440,
:545-566).  The reference's house_04 sample was the only seamed stream any test decoded; the synthetic writer now takes
attributes per corner (synth.encode_mesh_corners), so every topology x seam pattern x coder option is covered three ways:
  * CPU: the oracle decodes the stream to its last byte and the decoded corners equal the INPUT mesh quantised by the numpy
    rules of tests/meshutil.py (no decoder or writer code on the expected side);
  * CPU: the product's general-path source, compiled for the host under ASan / UBSan, equals the oracle;
  * GPU: the HIP path through the C-ABI equals the oracle bit for bit and the input per corner."""
import os

import numpy as np
import pytest

import oracle
import draco_sharp_amd.synth as synth
from meshutil import seamed_mesh, source_corner_faces_seamed
from test_independent_pin import decoded_faces

TOPOLOGIES = [(synth.GRID, 12, 9), (synth.TORUS, 10, 8), (synth.SPHERE, 8, 7), (synth.HOLES, 20, 16), (synth.TWO_PARTS, 8, 6)]
# (normal charts, uv charts): one seamed attribute, two with different seams, seams on every edge, a lone cut-out face,
# per-corner ids without any seam
PATTERNS = [(None, "stripes"), ("stripes", None), ("checker", "island"), ("random", "random"), ("single", "none"), ("island", "checker")]
OPTIONS = [dict(), dict(predictive_connectivity=2), dict(predictive_connectivity=1), dict(uv_prediction=5),
           dict(uv_prediction=5, predictive_connectivity=2, normal_prediction=6), dict(uv_prediction=2), dict(uv_prediction=4, pos_prediction=4),
           dict(force_scheme=0), dict(uv_prediction=0, pos_prediction=0), dict(traversal_method=2), dict(traversal_method=1, normal_prediction=6),
           dict(raw_integers=2), dict(no_prediction=7)]


def expected_and_decoded(args, m):
    pos, faces, nrm, nid, uv, uid = args
    expected, params = source_corner_faces_seamed(pos, faces, nrm, nid, uv, uid)
    ident = np.arange(m.num_points, dtype=np.uint32)
    got = decoded_faces(m.faces, [(a.portable, a.point_map if len(a.point_map) else ident) for a in m.attributes])
    return expected, got


@pytest.mark.parametrize("kind,nx,ny", TOPOLOGIES)
@pytest.mark.parametrize("charts", PATTERNS)
def test_seamed_streams_decode_to_the_quantised_input(kind, nx, ny, charts):
    args = seamed_mesh(synth, kind, nx, ny, 5, *charts)
    for opt in OPTIONS:
        data = synth.encode_mesh_corners(*args, opt=synth.options(**opt))
        m = oracle.decode(data)
        assert m.end_pos == len(data), opt
        # element type 1 (corner attribute) exactly for the attributes that have a seam
        seams = [c not in (None, "none") for c in charts]
        assert [d["element_type"] for d in m.decoders] == [0] + [int(s) for s in seams], opt
        assert m.attributes[0].num_entries == len(args[0])
        for a, s in zip(m.attributes[1:], seams):
            assert (a.num_entries > len(args[0])) == s
        expected, got = expected_and_decoded(args, m)
        assert got.shape == expected.shape and np.array_equal(got, expected), opt


def test_seam_writer_refuses_what_it_cannot_express():
    pos, faces, nrm, nid, uv, uid = seamed_mesh(synth, synth.GRID, 6, 5, 1, None, "stripes")
    with pytest.raises(RuntimeError, match="connectivity of their own"):
        synth.encode_mesh_corners(pos, faces, nrm, nid, uv, uid, opt=synth.options(single_connectivity=1))
    with pytest.raises(ValueError):
        synth.encode_mesh_corners(pos, faces, nrm, nid, uv, uid[:-1])
    with pytest.raises(RuntimeError, match="out of range"):
        synth.encode_mesh_corners(pos, faces, nrm, nid, uv[:-1], uid)


def test_seams_at_64k_triangles():
    args = seamed_mesh(synth, synth.GRID, 128, 256, 9, "stripes", "checker")
    data = synth.encode_mesh_corners(*args, opt=synth.options(predictive_connectivity=2, uv_prediction=5))
    m = oracle.decode(data)
    assert m.num_faces == 65536 and m.end_pos == len(data)
    expected, got = expected_and_decoded(args, m)
    assert np.array_equal(got, expected)


# ------------------------------------------------------------------ the general path's source on the host (ASan / UBSan)
@pytest.mark.parametrize("kind,nx,ny", TOPOLOGIES)
def test_seamed_streams_through_the_general_path_source(kind, nx, ny, tmp_path):
    import test_hostcheck as th
    exe = _hostcheck_exe(th)
    for charts in PATTERNS:
        args = seamed_mesh(synth, kind, nx, ny, 7, *charts)
        for opt in (dict(), dict(predictive_connectivity=2, uv_prediction=5, normal_prediction=6), dict(uv_prediction=4, pos_prediction=2, force_scheme=0)):
            data = synth.encode_mesh_corners(*args, opt=synth.options(**opt))
            status, detail, got = th.host_decode(exe, data, tmp_path, force=True)
            assert status == 0, (charts, opt, detail)
            th.assert_equals_oracle(got, oracle.decode(data))


def _hostcheck_exe(th):
    import os
    import subprocess
    deps = [th.SRC] + [os.path.join(th.CSRC, f) for f in ("dsa_general.h", "dsa_common.h", "dsa_host_parse.h", "dsa_types.h")]
    if not os.path.exists(th.EXE) or any(os.path.getmtime(d) > os.path.getmtime(th.EXE) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=signed-integer-overflow",
                        "-fno-sanitize-recover=undefined", "-o", th.EXE, th.SRC], check=True)
    return th.EXE


def test_corrupt_seamed_streams_never_leave_their_regions(tmp_path):
    import subprocess
    import test_hostcheck as th
    exe = _hostcheck_exe(th)
    cases = [seamed_mesh(synth, synth.TORUS, 10, 8, 3, "checker", "stripes"), seamed_mesh(synth, synth.HOLES, 14, 12, 4, None, "random")]
    for k, args in enumerate(cases):
        data = synth.encode_mesh_corners(*args, opt=synth.options(predictive_connectivity=2 * k, uv_prediction=5 if k else 1))
        src = tmp_path / ("fuzz%d.drc" % k)
        src.write_bytes(data)
        r = subprocess.run([exe, "fuzz", str(src), "1500", str(41 + k), "force"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        counts = dict(zip(r.stdout.split()[::2], map(int, r.stdout.split()[1::2])))
        assert counts["ok"] + counts["invalid"] + counts["notimpl"] + counts["notgeneral"] == 1500 and counts["invalid"] > 0


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ctx():
    import draco_sharp_amd as dsa
    c = dsa.Context(0)
    yield c
    c.close()


def _gpu_check(ctx, cases):
    """cases: [(args of encode_mesh_corners, stream)] decoded in one batch: equal to the oracle in every array and to the
    input per corner."""
    import draco_sharp_amd as dsa
    from test_gpu_parity import assert_same
    b = dsa.Batch(ctx, [s for _, s in cases])
    b.decode()
    paths = []
    for i, (args, s) in enumerate(cases):
        assert b.status(i) == 0, (i, b.status(i), b.mesh_info(i).detail)
        got = b.result(i)
        assert_same(got, oracle.decode(s))
        m = got.ConnectedData
        expected, _ = source_corner_faces_seamed(*args)
        mine = decoded_faces(m.Faces, [(a.PortableValues, a.PointMap) for a in m.Attributes])
        assert mine.shape == expected.shape and np.array_equal(mine, expected)
        paths.append(b.mesh_info(i).decode_path)
    b.close()
    return paths


@pytest.mark.gpu
@pytest.mark.parametrize("traversal", [0, 2])
def test_gpu_seamed_meshes_on_every_topology(ctx, traversal):
    cases = []
    for kind, nx, ny in TOPOLOGIES + [(synth.GRID, 40, 33), (synth.TORUS, 24, 40)]:
        for charts in PATTERNS:
            args = seamed_mesh(synth, kind, nx, ny, 11, *charts)
            cases.append((args, synth.encode_mesh_corners(*args, opt=synth.options(predictive_connectivity=traversal))))
    _gpu_check(ctx, cases)


@pytest.mark.gpu
def test_gpu_seamed_meshes_with_every_scheme(ctx):
    cases = []
    for k, opt in enumerate(OPTIONS):
        kind, nx, ny = TOPOLOGIES[k % len(TOPOLOGIES)]
        for charts in (PATTERNS[k % len(PATTERNS)], PATTERNS[(k + 3) % len(PATTERNS)]):
            args = seamed_mesh(synth, kind, nx + k, ny + 2, 20 + k, *charts)
            cases.append((args, synth.encode_mesh_corners(*args, opt=synth.options(**opt))))
    _gpu_check(ctx, cases)


@pytest.mark.gpu
def test_gpu_seams_take_the_fast_kernels(ctx, house04_bytes):
    """Corner-attribute decoders run on the wave-per-mesh kernels (k_seam_tables, k_traverse_att, k_seam_maps; TexCoordsPortable by
    k_texcoords): decode_path 0 for raw-coded seamed streams of every topology, standard and valence connectivity, one and two
    seamed attributes -- and for the reference's own sample, house_04 (valence symbols in tagged context lists, 59 topology splits,
    UV seams, TexCoordsPortable) -- GeometricNormal on normals with seams included (the fan of an entry ends at the seams).  What
    still takes the second chance: the multi-parallelogram schemes on a corner attribute, prediction-degree order."""
    import draco_sharp_amd as dsa
    from test_gpu_parity import assert_same
    cases = []
    for kind, nx, ny in TOPOLOGIES + [(synth.GRID, 40, 33)]:
        for charts in ((None, "stripes"), ("checker", "island"), ("random", "random")):
            for opt in (dict(), dict(uv_prediction=5), dict(predictive_connectivity=2, uv_prediction=5), dict(normal_prediction=6, uv_prediction=5)):
                args = seamed_mesh(synth, kind, nx, ny, 13, *charts)
                cases.append((args, synth.encode_mesh_corners(*args, opt=synth.options(force_scheme=1, **opt))))
    paths = _gpu_check(ctx, cases)
    assert set(paths) == {0}, paths
    # tagged symbols / uncompressed integers in a corner attribute: their extent is the entry count, so the walk of the stream stops
    # in front of them and is taken up behind k_seam_tables -- with whatever follows them in the stream -- still on the fast kernels
    late = []
    for kind, nx, ny in TOPOLOGIES:
        for charts in ((None, "stripes"), ("stripes", "checker"), ("random", None)):
            for opt in (dict(force_scheme=0), dict(raw_integers=2), dict(force_scheme=0, uv_prediction=5, normal_prediction=6), dict(force_scheme=0, predictive_connectivity=2)):
                args = seamed_mesh(synth, kind, nx, ny, 17, *charts)
                late.append((args, synth.encode_mesh_corners(*args, opt=synth.options(**opt))))
    assert set(_gpu_check(ctx, late)) == {0}
    b = dsa.Batch(ctx, [house04_bytes])
    b.decode()
    assert b.status(0) == 0 and b.mesh_info(0).decode_path == 0
    assert_same(b.result(0), oracle.decode(house04_bytes))
    b.close()


@pytest.mark.gpu
def test_gpu_seams_at_64k_triangles(ctx):
    cases = []
    for kind, charts, opt in ((synth.GRID, ("stripes", "checker"), dict(predictive_connectivity=2, uv_prediction=5)),
                              (synth.TORUS, (None, "island"), dict()), (synth.GRID, ("random", "random"), dict(force_scheme=0))):
        args = seamed_mesh(synth, kind, 128, 256, 9, *charts)
        cases.append((args, synth.encode_mesh_corners(*args, opt=synth.options(**opt))))
    assert _gpu_check(ctx, cases) == [0, 0, 0]         # the fast kernels throughout (the last stream has tagged symbols in its corner attributes)


@pytest.mark.gpu
def test_gpu_corrupt_seamed_streams_agree_with_the_oracle(ctx, house04_bytes):
    """A slice of tools/fuzz_seams.py in the suite: corrupted seamed / TexCoordsPortable / GeometricNormal streams through the fast
    seam kernels -- where the oracle decodes, the device decodes the same or refuses by one of its own checks; where the oracle
    refuses, the device does not succeed."""
    import draco_sharp_amd as dsa
    from test_gpu_parity import _corruptions, assert_same
    families = []
    for k, (kind, nx, ny, charts, opt) in enumerate([
            (synth.TORUS, 12, 10, ("checker", "stripes"), dict(force_scheme=1, uv_prediction=5, normal_prediction=6)),
            (synth.HOLES, 20, 16, (None, "random"), dict(predictive_connectivity=2)),
            (synth.GRID, 24, 20, ("stripes", "island"), dict(force_scheme=0, uv_prediction=5)),
            (synth.SPHERE, 10, 9, (None, "checker"), dict(raw_integers=2))]):
        families.append(synth.encode_mesh_corners(*seamed_mesh(synth, kind, nx, ny, 5 + k, *charts), opt=synth.options(**opt)))
    families.append(house04_bytes)
    streams = []
    for k, f in enumerate(families):
        streams += _corruptions(f, 80, 300 + k) + [f]
    b = dsa.Batch(ctx, streams)
    b.decode()
    equal = refused = stricter = 0
    sites = set()
    for i, sbytes in enumerate(streams):
        try:
            ref = oracle.decode(sbytes)
        except oracle.OracleError:
            ref = None
        st = b.status(i)
        if ref is not None and st == 0:
            assert_same(b.result(i), ref)
            equal += 1
        elif ref is None:
            assert st != 0, (i, "the device accepted a stream the oracle rejects")
            refused += 1
        else:
            stricter += 1
            info = b.mesh_info(i)
            sites.add((info.status, info.detail))
    assert equal >= len(families) and refused > 0
    # the device's own checks that the oracle does not make: the census of linked corners (263), an attribute traversal that does not
    # reach every attribute vertex (305), more orientation / flip bits than entries (668, 657), the bound on seam data (681)
    assert sites <= {(1, 263), (1, 305), (1, 668), (1, 657), (1, 681), (1, 123)}, sites
    assert stricter <= 8, (equal, refused, stricter, sites)
    b.close()


@pytest.mark.gpu
def test_gpu_late_located_vertex_attribute_beside_geometric_normals(ctx):
    """A parallelogram attribute on the POSITION connectivity that the walk locates only behind the seam tables (it stands behind a
    corner attribute with tagged symbols or uncompressed integers) reads its operands from the operand region of the position
    connectivity -- the region k_vertex_positions later fills with the positions by vertex for the GeometricNormal predictor.
    (A build of round 4 ran k_vertex_positions in front of that attribute's prediction: decode_path 0, status 0, wrong values.)"""
    cases = []
    for kind, nx, ny in TOPOLOGIES:
        for charts in (("random", None), ("stripes", None), ("checker", None)):
            for opt in (dict(force_scheme=0, normal_prediction=6), dict(raw_integers=2, normal_prediction=6), dict(force_scheme=0, normal_prediction=6, predictive_connectivity=2),
                        dict(force_scheme=0, normal_prediction=6, pos_prediction=0)):
                args = seamed_mesh(synth, kind, nx, ny, 23, *charts)
                cases.append((args, synth.encode_mesh_corners(*args, opt=synth.options(**opt))))
    assert set(_gpu_check(ctx, cases)) == {0}


@pytest.mark.gpu
def test_gpu_matrix_of_dialect_switches(ctx):
    """tools/dialect_matrix.py: the full product of seam pattern per attribute x symbol scheme x prediction scheme per attribute x
    connectivity symbols on four small topologies, plus a per-vertex family over quantisation bits, decoder layout and a generic
    attribute (3 888 streams; small batches, one crowded batch, a crowded batch without seams) -- the combinations a random draw
    reaches rarely -- equal to the oracle, most of them on the wave-per-mesh kernels."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("dialect_matrix", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "dialect_matrix.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, bad, paths = mod.run(1, ctx)
    assert n > 2000 and bad == 0
    assert paths.get(0, 0) > 0.7 * sum(paths.values()) and set(paths) <= {0, 2}
