"""The per-lane bodies of the product's lane-per-chain kernels (draco-sharp_amd/csrc/dsa_lanes.h: rANS symbol decode
one lane per stream, prediction inverse one lane per attribute) and the stream walk of k_locate (dsa_locate.h),
compiled for the host under AddressSanitizer + UBSan (tests/hostcheck/lanes_host.cpp) and compared with the oracle.
The connectivity the parallelogram operands need is taken from the oracle; the operands themselves are computed by
the product's para_operands_of.  A check of the product source on CPU, not a CPU decode path of the product."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle
import draco_sharp_amd.synth as synth

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostcheck", "lanes_host.cpp")
CSRC = os.path.join(HERE, "..", "draco-sharp_amd", "csrc")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    # always built from the sources of this checkout (a binary left in the tree could be older than they are)
    out = str(tmp_path_factory.mktemp("hostcheck") / "lanes_host")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=signed-integer-overflow",
                    "-fno-sanitize-recover=undefined", "-o", out, SRC], check=True)
    return out


def lanes_decode(exe, data, tmp_path, ref=None):
    src, out, conn = tmp_path / "in.drc", tmp_path / "out.bin", tmp_path / "conn.bin"
    src.write_bytes(data)
    args = [exe, "decode", str(src), str(out)]
    if ref is not None and ref.num_faces:
        d2c = ref.decoders[0]["data_to_corner"]
        conn.write_bytes(struct.pack("<III", ref.num_faces, ref.num_vertices, len(d2c)) + ref.opposite.tobytes() +
                         ref.corner_to_vertex.tobytes() + d2c.astype(np.uint32).tobytes())
        args.append(str(conn))
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr[-4000:]
    raw = out.read_bytes()
    status, detail, natt = struct.unpack_from("<iiI", raw, 0)
    off, atts = 12, []
    for _ in range(natt if status == 0 else 0):
        decoded, entries, ncp = struct.unpack_from("<III", raw, off); off += 12
        vals = None
        if decoded:
            vals = np.frombuffer(raw, np.int32, entries * ncp, off).reshape(entries, ncp); off += 4 * entries * ncp
        atts.append(vals)
    return status, detail, atts


CASES = [
    (synth.GRID, 40, 33, {}), (synth.TORUS, 24, 20, {}), (synth.SPHERE, 12, 11, {}), (synth.HOLES, 20, 16, {}), (synth.TWO_PARTS, 12, 9, {}),
    (synth.GRID, 40, 33, {"pos_bits": 14, "uv_bits": 12, "normal_bits": 10}),          # 13..15-bit rANS precision
    (synth.TORUS, 24, 20, {"pos_prediction": 0, "uv_prediction": 0}),                     # difference + wrap
    (synth.GRID, 30, 21, {"single_connectivity": 1}), (synth.GRID, 64, 64, {"pos_bits": 8, "uv_bits": 6, "normal_bits": 4}),
    (synth.GRID, 128, 64, {}),
    # non-canonicalised octahedral transform; prediction method -2 (the symbols are the values)
    (synth.HOLES, 20, 16, {"normal_transform": 2}), (synth.GRID, 40, 33, {"no_prediction": 7}), (synth.TORUS, 24, 20, {"no_prediction": 2, "normal_transform": 2}),
]


@pytest.mark.parametrize("kind,nx,ny,opts", CASES)
def test_lane_kernels_reproduce_the_oracle(exe, tmp_path, kind, nx, ny, opts):
    pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 11)
    data = synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=1, **opts))
    ref = oracle.decode(data)
    status, detail, atts = lanes_decode(exe, data, tmp_path, ref)
    assert status == 0, detail
    assert len(atts) == len(ref.attributes)
    seen = 0
    for got, r in zip(atts, ref.attributes):
        if got is None:
            continue                                  # alphabet beyond the lane tiers: the wave-per-stream kernels take it
        seen += 1
        assert np.array_equal(got, r.portable)
    assert seen >= 1


def test_point_cloud_delta_chain(exe, tmp_path):
    rng = np.random.default_rng(5)
    pos = np.cumsum(rng.normal(0, 0.01, (3000, 3)), axis=0).astype(np.float32)
    data = synth.encode_point_cloud(pos, opt=synth.options(force_scheme=1))
    ref = oracle.decode(data)
    status, detail, atts = lanes_decode(exe, data, tmp_path)
    assert status == 0, detail
    assert atts[0] is not None and np.array_equal(atts[0], ref.attributes[0].portable)


def test_tagged_streams_walk_under_asan(exe, tmp_path):
    """The tag streams of the tagged scheme are decoded by the stream walk itself (slot table + 16-byte chunks):
    valid streams must come out of it with status 0, damaged ones with a verdict and no out-of-bounds access."""
    rng = np.random.default_rng(4)
    for kind, nx, ny in ((synth.GRID, 40, 33), (synth.TORUS, 24, 20), (synth.HOLES, 20, 16)):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 13)
        good = synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=0))
        status, detail, atts = lanes_decode(exe, good, tmp_path)
        assert status == 0, detail
        for it in range(40):
            b = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(len(b) // 3, len(b)))] = int(rng.integers(0, 256))
            lanes_decode(exe, bytes(b), tmp_path)            # any verdict; ASan aborts on a bad access


def test_corrupt_tables_and_payloads_stay_in_bounds(exe, tmp_path):
    """Random damage to a valid stream: whatever the verdict, no access may leave the regions (ASan aborts the run);
    streams that still decode must agree with the oracle."""
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 24, 20, 3)
    good = synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=1))
    rng = np.random.default_rng(9)
    agreed = rejected = 0
    for it in range(150):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(len(b) // 3, len(b)))      # the attribute sections are behind the connectivity
            b[at] = int(rng.integers(0, 256))
        if it % 10 == 0:
            b = b[:int(rng.integers(len(b) // 2, len(b)))]
        try:
            ref = oracle.decode(bytes(b))
        except oracle.OracleError:
            ref = None
        status, detail, atts = lanes_decode(exe, bytes(b), tmp_path, ref)
        if status != 0:
            rejected += 1
            continue
        if ref is None:
            continue      # the walk + symbol + prediction stages found nothing wrong; a later stage of the full path does
        for got, r in zip(atts, ref.attributes):
            if got is not None:
                assert np.array_equal(got, r.portable)
        agreed += 1
    assert agreed > 0 and rejected > 0


def test_canonical_octahedral_recursion_equals_the_reference_step(exe):
    """ln_predict_oct advances the canonicalised octahedral transform in the canonical frame (dsa_lanes.h); here against
    ComputeOriginalValue entry by entry on random walks across every edge case of the transform, under ASan / UBSan."""
    for seed in (1, 2, 3):
        r = subprocess.run([exe, "octcheck", str(seed), "4000"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr[-2000:]


def test_packed_octahedral_step_is_exact_on_every_value(exe):
    """oct_pk_step (dsa_common.h: both components of the canonicalised octahedral delta in the 16-bit halves of one register,
    what k_predict_oct_streams runs) against ComputeOriginalValue on every value of the square x every correction in
    [0, max_q], octahedra of 2 - 5 bits (a million pairs); the random walks of the test above cover it up to 14 bits."""
    r = subprocess.run([exe, "octexhaust", "5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert "5 bits, 984064 pairs equal" in r.stdout


def test_division_through_a_double_estimate_is_exact(exe):
    """div_trunc_pos (dsa_common.h): the 64-bit divisions of the GeometricNormal arithmetic, on the device a double-precision estimate
    corrected by remainders instead of the compiler's division routine -- equal to the operator on two million random pairs of every
    magnitude, a fifth of them next to exact multiples."""
    r = subprocess.run([exe, "divcheck", "5", "2000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
