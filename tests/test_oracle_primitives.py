"""Known answers the reference's own unit tests hold for the L0 utilities
(tests/Draco.UnitTests/IO/{EncoderBufferTests,ConstantsTests}.cs, IO/Core/MathUtilitiesTests.cs),
checked against the CPU oracle, plus entropy-coder round trips."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle
import draco_sharp_amd.synth as synth


def _buf(b):
    return (C.c_uint8 * len(b)).from_buffer_copy(b)


def test_varint_known_answers():
    # EncoderBufferTests.cs:29-46 round-trips 98 and 1739 (LEB128)
    L = oracle.lib()
    for value, enc in ((98, bytes([98])), (1739, bytes([0xCB, 0x0D]))):
        n = C.c_size_t()
        assert L.orc_varint(_buf(enc), len(enc), C.byref(n)) == value
        assert n.value == len(enc)


def test_bits_known_answer():
    # EncoderBufferTests.cs:8-27: 9 bits 0b001100010 written LSB-first read back identically
    value = 0b001100010
    enc = bytes([value & 0xFF, value >> 8])
    counts = np.array([9], np.int32)
    out = np.zeros(1, np.uint32)
    pos = oracle.lib().orc_bits(_buf(enc), 2, counts.ctypes.data, 1, out.ctypes.data)
    assert out[0] == value and pos == 2
    # a field with bit 8 set (the reference's (byte) cast, defect D-2, would drop it)
    value = 0b101100010
    enc = bytes([value & 0xFF, value >> 8])
    oracle.lib().orc_bits(_buf(enc), 2, counts.ctypes.data, 1, out.ctypes.data)
    assert out[0] == value


def test_int_sqrt_known_answers():
    # MathUtilitiesTests.cs:7-20
    L = oracle.lib()
    assert L.orc_int_sqrt(0) == 0
    assert L.orc_int_sqrt(4) == 2
    assert L.orc_int_sqrt(48722615824) == 220732


def test_reinterpret_and_zigzag():
    # ConstantsTests.cs:7-21: int -3 reinterpreted as uint is 4294967293
    assert np.array([-3], np.int32).view(np.uint32)[0] == 4294967293
    assert np.isnan(np.array([-3], np.int32).view(np.float32)[0])
    L = oracle.lib()
    assert [L.orc_zigzag(s) for s in (0, 1, 2, 3, 4)] == [0, -1, 1, -2, 2]


def _decode_symbols(block, n, nc):
    out = np.zeros(n, np.uint32)
    used = oracle.lib().orc_decode_symbols(_buf(block), len(block), n, nc, out.ctypes.data)
    return out, used


@pytest.mark.parametrize("scheme", [0, 1, -1])
@pytest.mark.parametrize("nc", [1, 2, 3])
def test_symbol_roundtrip_geometric(scheme, nc):
    rng = np.random.default_rng(5 + nc)
    vals = (rng.geometric(0.05, 3000 * nc) - 1).astype(np.uint32)
    block = synth.encode_symbols(vals, nc, scheme)
    out, used = _decode_symbols(block, vals.size, nc)
    assert used == len(block)
    assert np.array_equal(out, vals)


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(0, 5000), min_size=1, max_size=400), st.sampled_from([0, 1, -1]), st.sampled_from([3, 5, 7, 10]))
def test_symbol_roundtrip_random(vals, scheme, level):
    v = np.array(vals, np.uint32)
    block = synth.encode_symbols(v, 1, scheme, level)
    out, used = _decode_symbols(block, v.size, 1)
    assert used == len(block)
    assert np.array_equal(out, v)


def test_symbol_roundtrip_wide_alphabet():
    # > 2^12 distinct symbols -> rANS precision above 12 bits; sparse table with zero runs
    rng = np.random.default_rng(11)
    vals = (rng.integers(0, 60000, 20000) * (rng.random(20000) < 0.7)).astype(np.uint32)
    block = synth.encode_symbols(vals, 1, 1)
    out, used = _decode_symbols(block, vals.size, 1)
    assert used == len(block) and np.array_equal(out, vals)


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(0, 1), min_size=1, max_size=2000), st.floats(0.0, 1.0))
def test_rabs_roundtrip(bits, bias):
    b = np.array(bits, np.uint8)
    if bias < 0.3:
        b[:] = 0
    block = synth.encode_rabs(b)
    out = np.zeros(b.size, np.uint8)
    used = oracle.lib().orc_decode_rabs(_buf(block), len(block), b.size, out.ctypes.data)
    assert used == len(block)
    assert np.array_equal(out, b)
