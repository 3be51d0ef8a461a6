"""Pins the oracle on the reference's only end-to-end vector:
src/Draco.Examples/Samples/house_04.obj.drc (+ house_04.obj), SURVEY.md Appendix C."""
import os

import numpy as np

import oracle
from meshutil import face_multiset, quantize

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_house04_landmarks(house04_bytes):
    m = oracle.decode(house04_bytes)
    assert (m.major, m.minor, m.encoder_type, m.encoder_method, m.flags) == (2, 2, 1, 1, 0)
    assert m.traversal_type == 2                      # valence Edgebreaker
    assert m.end_pos == len(house04_bytes) == 8196    # parses to the last byte
    assert m.num_faces == 2588 and m.num_attribute_data == 2
    assert [d["att_data_id"] for d in m.decoders] == [-1, 0, 1]
    assert [d["element_type"] for d in m.decoders] == [0, 1, 0]
    pos, uv, gen = m.attributes
    assert (pos.att_type, pos.data_type, pos.num_components, pos.seq_type) == (0, 9, 3, 2)
    assert (pos.pred_method, pos.pred_transform, pos.q_bits, pos.num_entries) == (1, 1, 11, 1775)
    assert np.allclose(pos.q_min[:3], [-538.2006, 0.0, -1003.7018], atol=1e-3) and abs(pos.q_range - 2009.9021) < 1e-3
    assert (uv.att_type, uv.num_components, uv.seq_type, uv.pred_method, uv.q_bits) == (3, 2, 2, 5, 10)
    assert (gen.att_type, gen.data_type, gen.num_components, gen.seq_type) == (4, 2, 1, 1)


def test_house04_matches_obj(house04_bytes):
    m = oracle.decode(house04_bytes)
    g = np.load(os.path.join(GOLD, "house_04_expected.npz"))
    pos, uv = m.attributes[0], m.attributes[1]
    qv = quantize(g["v"], pos.q_min[:3], pos.q_range, pos.q_bits)
    qvt = quantize(g["vt"], uv.q_min[:2], uv.q_range, uv.q_bits)
    # expected: every obj face corner carries (quantised position, quantised uv)
    exp_keys = np.concatenate([qv[g["fv"].ravel()], qvt[g["fvt"].ravel()]], axis=1)
    exp = face_multiset(np.arange(exp_keys.shape[0]).reshape(-1, 3), exp_keys)
    got_keys = np.concatenate([pos.portable[pos.point_map], uv.portable[uv.point_map]], axis=1)
    got = face_multiset(m.faces, got_keys)
    assert got == exp
    # dequantised floats land within half a quantisation step of the source geometry
    step = pos.q_range / ((1 << pos.q_bits) - 1)
    src = {tuple(k): g["v"][i] for i, k in enumerate(qv)}
    for e in range(0, pos.num_entries, 37):
        ref = src[tuple(int(x) for x in pos.portable[e])]
        assert np.all(np.abs(pos.values[e] - ref) <= 0.5 * step * 1.001 + 1e-3)
