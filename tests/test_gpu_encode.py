"""Encode direction (BASELINE.json configs[4]): quantisation, prediction and rANS coding as HIP kernels must
produce, byte for byte, the stream of the CPU coder (draco-sharp_amd/csrc/dsa_encode_host.h through
draco_sharp_amd.synth) -- and therefore round-trip bit-exactly through the decoder."""
import numpy as np
import pytest

import oracle
import draco_sharp_amd as dsa
import draco_sharp_amd.synth as synth
from meshutil import face_multiset_fast, source_corner_faces

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = dsa.Context(0)
    yield c
    c.close()


def cpu_stream(pos, faces, nrm, uv, cfg):
    opt = synth.options(pos_bits=cfg.position_bits, uv_bits=cfg.texcoord_bits, normal_bits=cfg.normal_bits,
                        single_connectivity=1 if cfg.single_connectivity else 0, force_scheme=cfg.symbol_scheme,
                        compression_level=10 - cfg.speed, pos_prediction=cfg.position_prediction, uv_prediction=cfg.texcoord_prediction)
    return synth.encode_mesh(pos, faces, nrm, uv, opt=opt)


CASES = [
    (synth.GRID, 9, 7, dsa.Config()),
    (synth.TORUS, 10, 8, dsa.Config(single_connectivity=True)),
    (synth.SPHERE, 8, 7, dsa.Config(symbol_scheme=0)),
    (synth.HOLES, 20, 16, dsa.Config(symbol_scheme=1, position_prediction=0, texcoord_prediction=0)),
    (synth.TWO_PARTS, 9, 6, dsa.Config(position_bits=16, texcoord_bits=14, normal_bits=10)),
    (synth.GRID, 40, 33, dsa.Config(position_bits=4, texcoord_bits=4, normal_bits=4, speed=1)),
    (synth.GRID, 128, 256, dsa.Config()),
    (synth.TORUS, 128, 256, dsa.Config(speed=9)),
]


def test_gpu_encoder_matches_cpu_coder_byte_for_byte(ctx):
    meshes, expected = [], []
    by_cfg = {}
    for kind, nx, ny, cfg in CASES:
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 21)
        by_cfg.setdefault(id(cfg), (cfg, []))[1].append((pos, faces, nrm, uv))
    enc = dsa.DracoEncoder(ctx)
    for cfg, group in by_cfg.values():
        got = enc.EncodeBatch([dsa.MeshData(p, f, n, u) for p, f, n, u in group], cfg)
        for (p, f, n, u), g in zip(group, got):
            exp = cpu_stream(p, f, n, u, cfg)
            assert len(g) == len(exp)
            assert g == exp
            meshes.append((p, f, n, u, cfg)); expected.append(g)
    # round trip through the GPU decoder: every face corner carries the quantised input -- position, octahedral normal
    # and texture coordinate, derived from the INPUT with the numpy rules of meshutil (not with the coder under test)
    b = dsa.Batch(ctx, expected)
    b.decode()
    for i, (p, f, n, u, cfg) in enumerate(meshes):
        assert b.status(i) == 0
        m = b.result(i).ConnectedData
        want, _ = source_corner_faces(p, n, u, f, cfg.position_bits, cfg.normal_bits, cfg.texcoord_bits)
        keys = np.concatenate([np.asarray(a.PortableValues, np.int64)[np.asarray(a.PointMap, np.int64)] for a in m.Attributes], axis=1)
        got = face_multiset_fast(m.Faces, keys)
        assert got.shape == want.shape and np.array_equal(got, want)
    b.close()


def test_positions_only_and_partial_attributes(ctx):
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 12, 9, 4)
    enc = dsa.DracoEncoder(ctx)
    for n_, u_ in ((None, None), (nrm, None), (None, uv)):
        g = enc.Encode(dsa.MeshData(pos, faces, n_, u_))
        assert g == synth.encode_mesh(pos, faces, n_, u_)
        ref = oracle.decode(g)
        assert ref.num_faces == len(faces) and len(ref.attributes) == 1 + (n_ is not None) + (u_ is not None)


def test_bad_mesh_fails_alone(ctx):
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 6, 5, 4)
    bad_faces = faces.copy(); bad_faces[0, 0] = len(pos) + 5                  # index out of range
    nonmanifold = np.concatenate([faces, faces[:1]])                          # duplicated face: duplicate half-edges
    enc = dsa.DracoEncoder(ctx)
    assert enc.Encode(dsa.MeshData(pos, faces, nrm, uv)) == synth.encode_mesh(pos, faces, nrm, uv)
    for f in (bad_faces, nonmanifold):
        with pytest.raises(dsa.InvalidDataException):
            enc.EncodeBatch([dsa.MeshData(pos, faces), dsa.MeshData(pos, f)])


def test_edge_case_inputs_match_cpu_coder(ctx):
    """One triangle, degenerate ranges, zero normals, huge coordinates, extreme bit depths."""
    enc = dsa.DracoEncoder(ctx)
    tri = np.array([[0, 1, 2]], np.uint32)
    p3 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    cases = [
        (p3, tri, None, None, dsa.Config()),
        (np.zeros((3, 3), np.float32) + 5.0, tri, np.zeros((3, 3), np.float32), np.zeros((3, 2), np.float32), dsa.Config()),     # range 0, zero normals
        (p3 * 1e6 - 3e5, tri, np.array([[0, 0, -1], [1e-9, 0, 0], [-1, -1, -1]], np.float32), p3[:, :2], dsa.Config(position_bits=20, texcoord_bits=20, normal_bits=20)),
        (p3, tri, np.array([[0, 0, 1], [0, 1, 0], [1, 0, 0]], np.float32), p3[:, :2], dsa.Config(position_bits=1, texcoord_bits=1, normal_bits=2)),
    ]
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 30, 20, 8)
    cases.append((pos * np.float32(1e-6), faces, -nrm, uv * 1000, dsa.Config(speed=0)))
    cases.append((pos, faces, nrm, uv, dsa.Config(speed=10, symbol_scheme=0)))
    for p, f, n_, u_, cfg in cases:
        got = enc.Encode(dsa.MeshData(p, f, n_, u_), cfg)
        assert got == cpu_stream(p, f, n_, u_, cfg)
        ref = oracle.decode(got)
        assert ref.num_faces == len(f)


def test_device_connectivity_equals_host_connectivity(ctx, monkeypatch):
    """k_enc_connectivity (corner table, Edgebreaker symbols, attribute order, operand entries on the device) against the
    host coder's connectivity behind the same attribute kernels (DSA_ENC_HOST_CONN=1): identical streams on every
    topology -- open and closed surfaces, holes (boundary loops met mid-traversal), handles (topology splits), several
    components -- and identical verdicts on meshes neither of them codes."""
    enc = dsa.DracoEncoder(ctx)
    group = []
    for kind, nx, ny in ((synth.GRID, 9, 7), (synth.TORUS, 10, 8), (synth.SPHERE, 8, 7), (synth.HOLES, 20, 16), (synth.HOLES, 33, 29),
                         (synth.TWO_PARTS, 9, 6), (synth.TORUS, 40, 24), (synth.GRID, 128, 256)):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 5)
        group.append(dsa.MeshData(pos, faces, nrm, uv))
    # two meshes the coders refuse: a face listed twice (non-manifold edge), two cones sharing only their apex (non-manifold vertex)
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 6, 5, 5)
    bad_edge = dsa.MeshData(pos, np.concatenate([faces, faces[:1]]), nrm, uv)
    apex = np.array([[0, 0, 0], [1, 0, 1], [0, 1, 1], [-1, 0, 1], [1, 0, -1], [0, 1, -1], [-1, 0, -1]], np.float32)
    bad_vertex = dsa.MeshData(apex, np.array([[0, 1, 2], [0, 2, 3], [0, 5, 4], [0, 6, 5]], np.uint32), None, None)

    def run():
        ok = enc.EncodeBatch(group)
        verdicts = []
        for m in (bad_edge, bad_vertex):
            try:
                enc.EncodeBatch([m]); verdicts.append("coded")
            except Exception as e:            # noqa: BLE001 - the binding raises the reference's exception type
                verdicts.append(type(e).__name__)
        return ok, verdicts

    monkeypatch.setenv("DSA_ENC_HOST_CONN", "0")            # unset, a batch this small would take the host path
    monkeypatch.setenv("DSA_ENC_HOST_PLAN", "0")
    dev, dev_verdicts = run()
    monkeypatch.setenv("DSA_ENC_HOST_CONN", "1")
    host, host_verdicts = run()
    assert len(dev) == len(host) == len(group)
    for d, h in zip(dev, host):
        assert d == h
    assert dev_verdicts == host_verdicts and "coded" not in dev_verdicts


def test_device_symbol_plan_equals_host_symbol_plan(ctx, monkeypatch):
    """k_enc_plan (scheme choice + table normalisation on the device, dsa_symbol_plan.h one lane per stream) against the
    same code run by the host between the two device phases (DSA_ENC_HOST_PLAN=1), and both against the CPU coder:
    forced and chosen schemes, every compression level's precision rule, coarse and fine quantisation."""
    enc = dsa.DracoEncoder(ctx)
    cfgs = [dsa.Config(), dsa.Config(symbol_scheme=0), dsa.Config(symbol_scheme=1), dsa.Config(speed=0), dsa.Config(speed=2), dsa.Config(speed=7),
            dsa.Config(speed=10), dsa.Config(position_bits=3, texcoord_bits=2, normal_bits=3), dsa.Config(position_bits=18, texcoord_bits=16, normal_bits=12)]
    group = []
    for k, (kind, nx, ny) in enumerate(((synth.GRID, 24, 20), (synth.TORUS, 16, 12), (synth.HOLES, 20, 16), (synth.GRID, 128, 256))):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 9 + k)
        group.append((pos, faces, nrm, uv))
    for cfg in cfgs:
        monkeypatch.setenv("DSA_ENC_HOST_PLAN", "0")
        monkeypatch.setenv("DSA_ENC_HOST_CONN", "0")
        dev = enc.EncodeBatch([dsa.MeshData(*m) for m in group], cfg)
        monkeypatch.setenv("DSA_ENC_HOST_PLAN", "1")
        host = enc.EncodeBatch([dsa.MeshData(*m) for m in group], cfg)
        for (p, f, n, u), d, h in zip(group, dev, host):
            assert d == h
            assert d == cpu_stream(p, f, n, u, cfg)


def test_vertex_of_huge_valence(ctx, monkeypatch):
    """A cone: 20 000 faces around one apex.  The device corner table searches the shorter of an edge's two vertex lists, so the
    apex costs its neighbours nothing; the stream equals the CPU coder's."""
    n = 20000
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    pos = np.concatenate([[[0, 0, 1]], np.stack([np.cos(ang), np.sin(ang), np.zeros(n)], 1)]).astype(np.float32)
    faces = np.stack([np.zeros(n, np.uint32), 1 + np.arange(n, dtype=np.uint32), 1 + (np.arange(n, dtype=np.uint32) + 1) % n], 1)
    enc = dsa.DracoEncoder(ctx)
    monkeypatch.setenv("DSA_ENC_HOST_CONN", "0")
    monkeypatch.setenv("DSA_ENC_HOST_PLAN", "0")
    got = enc.EncodeBatch([dsa.MeshData(pos, faces)])[0]
    assert got == synth.encode_mesh(pos, faces, None, None)
    ref = oracle.decode(got)
    assert ref.num_faces == n


def _tool(name, *args):
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    return subprocess.run([sys.executable, os.path.join("tools", name)] + [str(a) for a in args], cwd=root, capture_output=True, text=True, timeout=900)


def test_damaged_meshes_get_the_same_verdict_from_device_and_host_connectivity():
    """A slice of tools/fuzz_encode.py (a process of its own: it sets the encoder's switches for the whole run): flipped, rewired,
    duplicated faces and random face soups -- the device connectivity codes the same meshes to the same bytes and refuses the same
    ones as the host coder."""
    r = _tool("fuzz_encode.py", 200)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert "0 verdicts / streams differ" in r.stdout


def test_randomised_encoder_options_slice():
    """A slice of tools/soak_encode.py: random topology / size / bit depths / compression level per mesh, device paths against the
    CPU coder byte for byte, every stream decoded again."""
    r = _tool("soak_encode.py", 60, 5)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]


def test_mixed_shapes_over_many_chunks_and_lanes(ctx, monkeypatch):
    """The chunk pipeline of dsa_encode_batch (uploads in turns by chunk, faces first, the walks on their own stream) and the
    lane-per-mesh walks with meshes of different shape in one wave: closed surfaces (an interior start face), holes, two
    components, sizes from 24 to 5000 faces -- every stream byte for byte the CPU coder's, whatever the chunk size and the
    number of meshes to a wave."""
    kinds = (synth.GRID, synth.TORUS, synth.SPHERE, synth.HOLES, synth.TWO_PARTS)
    meshes, expected = [], []
    for k in range(75):
        kind = kinds[k % len(kinds)]
        nx, ny = 4 + (7 * k) % 47, 3 + (5 * k) % 53
        if kind == synth.HOLES: nx, ny = max(nx, 12), max(ny, 12)
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 100 + k)
        meshes.append(dsa.MeshData(pos, faces, nrm, uv)); expected.append(synth.encode_mesh(pos, faces, nrm, uv))
    enc = dsa.DracoEncoder(ctx)
    monkeypatch.setenv("DSA_ENC_HOST_CONN", "0")
    monkeypatch.setenv("DSA_ENC_HOST_PLAN", "0")
    for chunk, walk_lanes in ((8, 8), (16, 3), (7, 64), (75, 16), (20, 1)):
        monkeypatch.setenv("DSA_ENC_CHUNK", str(chunk))
        monkeypatch.setenv("DSA_ENC_WALK_LANES", str(walk_lanes))
        got = enc.EncodeBatch(meshes)
        assert got.sizes == [len(e) for e in expected]
        for i, e in enumerate(expected):
            assert got[i] == e, (chunk, walk_lanes, i)


def test_a_chunk_of_nothing_but_bad_meshes_holds_nobody_up(ctx, monkeypatch):
    """A chunk whose meshes all fail the host's checks uploads nothing; its turn on the link must pass on (the other chunks'
    uploads wait for their turn in chunk order), and the batch reports the failure instead of standing still."""
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 6, 5, 4)
    bad = faces.copy(); bad[0, 0] = len(pos) + 5
    good, broken = dsa.MeshData(pos, faces, nrm, uv), dsa.MeshData(pos, bad, nrm, uv)
    enc = dsa.DracoEncoder(ctx)
    monkeypatch.setenv("DSA_ENC_HOST_CONN", "0")
    monkeypatch.setenv("DSA_ENC_CHUNK", "4")
    for layout in ([good] * 4 + [broken] * 4 + [good] * 12, [broken] * 4 + [good] * 20, [good] * 20 + [broken] * 4):
        with pytest.raises(dsa.InvalidDataException):
            enc.EncodeBatch(layout)
    assert enc.EncodeBatch([good] * 24)[23] == synth.encode_mesh(pos, faces, nrm, uv)


def test_one_triangle_and_other_tiny_meshes_on_the_device_path(ctx, monkeypatch):
    """A mesh of one face fills its traversal stack with its only entry (the device walks once took a full stack for an
    overflow); tiny closed and fan-shaped meshes beside it, connectivity and plans forced onto the device."""
    monkeypatch.setenv("DSA_ENC_HOST_CONN", "0")
    monkeypatch.setenv("DSA_ENC_HOST_PLAN", "0")
    p3 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    p4 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    fan = np.array([[0, i, i + 1] for i in range(1, 40)] + [[0, 40, 1]], np.uint32)
    pf = np.concatenate([[[0, 0, 1]], [[np.cos(a), np.sin(a), 0] for a in np.linspace(0, 2 * np.pi, 40, endpoint=False)]]).astype(np.float32)
    cases = [(p3, np.array([[0, 1, 2]], np.uint32)), (p4, np.array([[0, 1, 2], [0, 2, 3]], np.uint32)),
             (p4, np.array([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]], np.uint32)), (pf, fan)]
    enc = dsa.DracoEncoder(ctx)
    got = enc.EncodeBatch([dsa.MeshData(p, f) for p, f in cases])
    for (p, f), g in zip(cases, got):
        assert g == synth.encode_mesh(p, f, None, None)
        assert oracle.decode(g).num_faces == len(f)


def test_more_than_65536_vertices_keep_32_bit_faces(ctx, monkeypatch):
    """Faces travel as 16-bit indices when a mesh has at most 65 536 vertices; a larger one keeps its 32-bit indices, both in
    one chunk."""
    monkeypatch.setenv("DSA_ENC_HOST_CONN", "0")
    monkeypatch.setenv("DSA_ENC_HOST_PLAN", "0")
    big = synth.make_mesh(synth.GRID, 300, 250, 5)            # 75 551 vertices
    edge = synth.make_mesh(synth.GRID, 255, 255, 6)           # 65 536 vertices exactly
    small = synth.make_mesh(synth.TORUS, 20, 12, 7)
    cases = [big, small, edge, small]
    assert len(big[0]) > 65536 and len(edge[0]) == 65536
    enc = dsa.DracoEncoder(ctx)
    got = enc.EncodeBatch([dsa.MeshData(p, f, n, u) for p, n, u, f in cases])
    for (p, n, u, f), g in zip(cases, got):
        assert g == synth.encode_mesh(p, f, n, u)


def test_generic_uint8_attributes_through_the_device_encoder(ctx, monkeypatch):
    """dsa_mesh_input.generic (ABI 4): one uint8 attribute of 1 - 4 components per vertex (vertex colours) -- coded as an integer
    attribute by the HIP kernels, byte for byte what the CPU coder writes, with the connectivity on the device and on the host, and
    back through the decoder to the input values."""
    enc = dsa.DracoEncoder(ctx)
    meshes, expect, inputs = [], [], []
    k = 0
    for kind, nx, ny in ((synth.GRID, 24, 17), (synth.TORUS, 16, 12), (synth.HOLES, 20, 16), (synth.SPHERE, 12, 9), (synth.TWO_PARTS, 9, 6), (synth.GRID, 128, 256)):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 61 + k)
        for gc in (1, 3, 4, 2):
            g = ((np.arange(len(pos) * gc, dtype=np.int64) * 104729 + 31 * k) % 256).astype(np.uint8).reshape(-1, gc)
            cfg = [dsa.Config(), dsa.Config(symbol_scheme=0), dsa.Config(single_connectivity=True), dsa.Config(position_prediction=0, texcoord_prediction=0)][k % 4]
            with_n, with_uv = k % 3 != 1, k % 5 != 2
            meshes.append((dsa.MeshData(pos, faces, nrm if with_n else None, uv if with_uv else None, generic=g), cfg))
            opt = synth.options(pos_bits=cfg.position_bits, uv_bits=cfg.texcoord_bits, normal_bits=cfg.normal_bits,
                                single_connectivity=1 if cfg.single_connectivity else 0, force_scheme=cfg.symbol_scheme,
                                compression_level=10 - cfg.speed, pos_prediction=cfg.position_prediction, uv_prediction=cfg.texcoord_prediction,
                                generic_components=gc)
            expect.append(synth.encode_mesh(pos, faces, nrm if with_n else None, uv if with_uv else None, generic=g, opt=opt))
            inputs.append(g)
            k += 1
    for host_conn in ("1", "0"):
        monkeypatch.setenv("DSA_ENC_HOST_CONN", host_conn)
        monkeypatch.setenv("DSA_ENC_HOST_PLAN", host_conn)
        for cfg_id in range(4):                      # one batch per configuration (the options are the batch's)
            idx = [i for i in range(len(meshes)) if i % 4 == cfg_id]
            out = enc.EncodeBatch([meshes[i][0] for i in idx], meshes[idx[0]][1])
            for j, i in enumerate(idx):
                assert out[j] == expect[i], (host_conn, i)
    # and back: the decoder returns the generic values per point
    b = dsa.Batch(ctx, expect)
    b.decode()
    for i, g in enumerate(inputs):
        assert b.status(i) == 0
        m = b.result(i).ConnectedData
        a = m.Attributes[-1]
        assert a.AttributeType == 4 and a.NumComponents == g.shape[1]
        ref = oracle.decode(expect[i])
        assert a.Values.tobytes() == ref.attributes[-1].values.tobytes()
        per_point = np.asarray(a.Values).reshape(-1, g.shape[1])[a.PointMap]
        assert sorted(map(tuple, per_point.tolist())) == sorted(map(tuple, g.tolist()))
    b.close()
