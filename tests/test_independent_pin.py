"""An independent pin for what the reference holds no vector for (standard Edgebreaker traversal, octahedral normals,
tagged symbols, delta / parallelogram prediction): the expected geometry is derived from the INPUT mesh with numpy
restatements of the quantisation rules (tests/meshutil.py: source_quantization, oct_quantize -- SURVEY.md App. D),
independent of the stream writer's C++, and the decoded face-corner multiset must equal it.  A reader bug and a writer
bug that agree no longer pass.  CPU: the oracle; GPU: the HIP path through the C-ABI at 64k triangles."""
import numpy as np
import pytest

import oracle
import draco_sharp_amd.synth as synth
from meshutil import face_multiset_fast, source_corner_faces

CASES = [(synth.GRID, 128, 256), (synth.TORUS, 128, 256), (synth.HOLES, 160, 200)]
OPTIONS = [dict(), dict(force_scheme=0), dict(pos_prediction=0, uv_prediction=0), dict(single_connectivity=1, force_scheme=1)]


def decoded_faces(faces, attributes):
    """attributes: [(portable[entries, nc], point_map[points])] in stream order position, normal, texture coordinate."""
    keys = np.concatenate([np.asarray(p, np.int64)[np.asarray(m, np.int64)] for p, m in attributes], axis=1)
    return face_multiset_fast(faces, keys)


def check_params(a_pos, a_uv, params):
    pmin, prange, umin, urange = params
    assert np.array_equal(np.asarray(a_pos[0][:3], np.float32), pmin) and np.float32(a_pos[1]) == prange
    assert np.array_equal(np.asarray(a_uv[0][:2], np.float32), umin) and np.float32(a_uv[1]) == urange


@pytest.mark.parametrize("kind,nx,ny", [(synth.GRID, 40, 33), (synth.TORUS, 24, 40), (synth.HOLES, 20, 16), (synth.SPHERE, 12, 11), (synth.TWO_PARTS, 12, 9)] + CASES[:2])
@pytest.mark.parametrize("opt", OPTIONS)
def test_oracle_reproduces_the_quantised_input(kind, nx, ny, opt):
    pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 31)
    expected, params = source_corner_faces(pos, nrm, uv, faces)
    m = oracle.decode(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt)))
    ap, an, au = m.attributes
    assert (ap.q_bits, an.oct_bits, au.q_bits) == (11, 8, 10)
    check_params((ap.q_min, ap.q_range), (au.q_min, au.q_range), params)
    ident = np.arange(m.num_points, dtype=np.uint32)
    got = decoded_faces(m.faces, [(a.portable, a.point_map if len(a.point_map) else ident) for a in (ap, an, au)])
    assert got.shape == expected.shape and np.array_equal(got, expected)


@pytest.mark.gpu
def test_gpu_reproduces_the_quantised_input_at_64k_triangles():
    import draco_sharp_amd as dsa
    ctx = dsa.Context(0)
    cases, streams = [], []
    for kind, nx, ny in CASES:
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 31)
        exp = source_corner_faces(pos, nrm, uv, faces)
        for opt in OPTIONS:
            cases.append(exp)
            streams.append(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt)))
    b = dsa.Batch(ctx, streams)
    b.decode()
    for i, (expected, params) in enumerate(cases):
        assert b.status(i) == 0, (i, b.status(i), b.mesh_info(i).detail)
        m = b.result(i).ConnectedData
        ap, an, au = m.Attributes
        check_params((ap.MinValues + [0], ap.Range), (au.MinValues + [0], au.Range), params)
        got = decoded_faces(m.Faces, [(a.PortableValues, a.PointMap) for a in (ap, an, au)])
        assert got.shape == expected.shape and np.array_equal(got, expected)
    b.close()
    ctx.close()
