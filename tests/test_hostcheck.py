"""The product's serial general path (draco-sharp_amd/csrc/dsa_general.h) compiled for the host under
AddressSanitizer + UBSan (tests/hostcheck/general_host.cpp): the same source the GPU runs must reproduce the
oracle on the reference's house_04 sample and on synthetic meshes, and must survive corrupt streams without a
single out-of-bounds access.  This is a check of the product source on CPU, not a CPU decode path of the product."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle
import draco_sharp_amd.synth as synth

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostcheck", "general_host.cpp")
EXE = os.path.join(HERE, "hostcheck", "general_host")
CSRC = os.path.join(HERE, "..", "draco-sharp_amd", "csrc")


@pytest.fixture(scope="module")
def exe():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("dsa_general.h", "dsa_common.h", "dsa_host_parse.h", "dsa_types.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=signed-integer-overflow",
                        "-fno-sanitize-recover=undefined", "-o", EXE, SRC], check=True)
    return EXE


def host_decode(exe, data, tmp_path, force, half_lut=False):
    src, out = tmp_path / "in.drc", tmp_path / "out.bin"
    src.write_bytes(data)
    env = dict(os.environ, DSA_HALF_LUT="1") if half_lut else None
    r = subprocess.run([exe, "decode", str(src), str(out)] + (["force"] if force else []), capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = out.read_bytes()
    status, detail = struct.unpack_from("<ii", raw, 0)
    if status != 0:
        return status, detail, None
    nf, npnt, natt = struct.unpack_from("<III", raw, 8)
    off = 20
    faces = np.frombuffer(raw, np.int32, 3 * nf, off).reshape(nf, 3); off += 12 * nf
    atts = []
    for _ in range(natt):
        entries, ncp = struct.unpack_from("<II", raw, off); off += 8
        pmap = np.frombuffer(raw, np.uint32, npnt, off); off += 4 * npnt
        portable = np.frombuffer(raw, np.int32, entries * ncp, off).reshape(entries, ncp) if ncp else None; off += 4 * entries * ncp
        atts.append((entries, pmap, portable))
    return 0, 0, (faces, npnt, atts)


def assert_equals_oracle(got, ref):
    faces, npnt, atts = got
    assert np.array_equal(faces, ref.faces) and npnt == ref.num_points and len(atts) == len(ref.attributes)
    for (entries, pmap, portable), r in zip(atts, ref.attributes):
        assert entries == r.num_entries and np.array_equal(pmap, r.point_map)
        if r.portable is not None:
            assert np.array_equal(portable, r.portable)


def test_house04_through_the_general_path_source(exe, house04_bytes, tmp_path):
    status, detail, got = host_decode(exe, house04_bytes, tmp_path, force=True)     # (seams take the fast kernels unless forced)
    assert status == 0, detail
    assert_equals_oracle(got, oracle.decode(house04_bytes))


@pytest.mark.parametrize("kind,nx,ny,opts", [
    (synth.TORUS, 10, 8, {}), (synth.HOLES, 14, 12, {"single_connectivity": 1}), (synth.SPHERE, 8, 7, {"force_scheme": 0}),
    (synth.TWO_PARTS, 9, 6, {"pos_prediction": 0, "uv_prediction": 0}), (synth.GRID, 40, 33, {"force_scheme": 1}),
    # GeometricNormal prediction (method 6): area-weighted face normals from the quantised positions + flip bits
    (synth.HOLES, 14, 12, {"normal_prediction": 6}), (synth.TWO_PARTS, 9, 6, {"normal_prediction": 6, "single_connectivity": 1}),
    (synth.GRID, 30, 21, {"normal_prediction": 6, "pos_bits": 20, "normal_bits": 14}),
    (synth.TORUS, 12, 9, {"uv_prediction": 5}), (synth.HOLES, 14, 12, {"pos_prediction": 4, "uv_prediction": 2}),
    (synth.TWO_PARTS, 9, 6, {"pos_prediction": 2, "uv_prediction": 4, "single_connectivity": 1}),
    # prediction-degree traversal: positions only / every decoder / all attributes in one decoder
    (synth.HOLES, 14, 12, {"traversal_method": 1, "pos_prediction": 4}), (synth.TORUS, 12, 9, {"traversal_method": 2, "uv_prediction": 5}),
    (synth.TWO_PARTS, 9, 6, {"traversal_method": 1, "single_connectivity": 1, "normal_prediction": 6}),
    # predictive Edgebreaker traversal
    (synth.HOLES, 20, 16, {"predictive_connectivity": 1}), (synth.TORUS, 12, 9, {"predictive_connectivity": 1, "single_connectivity": 1}),
    # valence Edgebreaker traversal on holes, handles and two components, with the predictors stock encoders pair it with
    (synth.HOLES, 20, 16, {"predictive_connectivity": 2, "uv_prediction": 5, "normal_prediction": 6}),
    (synth.TORUS, 12, 9, {"predictive_connectivity": 2, "single_connectivity": 1, "force_scheme": 0}),
    (synth.TWO_PARTS, 9, 6, {"predictive_connectivity": 2, "pos_prediction": 4, "traversal_method": 1}), (synth.HOLES, 14, 12, {"uv_prediction": 5, "normal_prediction": 6, "single_connectivity": 1}),
    # decoder branches no stock setting reaches: non-canonicalised octahedral transform, uncompressed integers, prediction method -2
    (synth.HOLES, 14, 12, {"normal_transform": 2}), (synth.TORUS, 10, 8, {"raw_integers": 4, "predictive_connectivity": 2}),
    (synth.GRID, 14, 11, {"raw_integers": 1, "pos_bits": 6, "uv_bits": 6, "normal_bits": 5}), (synth.TWO_PARTS, 9, 6, {"raw_integers": 2, "pos_bits": 10, "normal_transform": 2, "force_scheme": 0}),
    (synth.HOLES, 14, 12, {"no_prediction": 7}), (synth.TORUS, 10, 8, {"no_prediction": 5, "uv_prediction": 5, "single_connectivity": 1})])
def test_synthetic_meshes_through_the_general_path_source(exe, tmp_path, kind, nx, ny, opts):
    pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 7)
    data = synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opts))
    ref = oracle.decode(data)
    for half_lut in (False, True):                    # the two rANS look-up table resolutions of the device launches
        status, detail, got = host_decode(exe, data, tmp_path, force=True, half_lut=half_lut)
        assert status == 0, detail
        assert_equals_oracle(got, ref)


@pytest.mark.parametrize("compressed", [True, False])
def test_sequential_mesh_through_the_general_path_source(exe, tmp_path, compressed):
    pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 20, 16, 7)
    data = synth.encode_mesh_sequential(pos, faces, nrm, uv, compressed=compressed)
    status, detail, got = host_decode(exe, data, tmp_path, force=False)
    assert status == 0, detail
    ref = oracle.decode(data)
    faces_got, npnt, atts = got
    assert np.array_equal(faces_got, ref.faces) and npnt == ref.num_points
    for (entries, pmap, portable), r in zip(atts, ref.attributes):
        assert entries == r.num_entries and np.array_equal(pmap, np.arange(npnt, dtype=np.uint32))
        assert np.array_equal(portable, r.portable)


def test_corrupt_streams_never_leave_their_regions(exe, house04_bytes, tmp_path):
    """Bit flips, random bytes, truncations and bursts: every gap between arena regions is poisoned, so any
    out-of-bounds access of the general path aborts the run."""
    pos, nrm, uv, faces = synth.make_mesh(synth.TORUS, 10, 8, 3)
    cases = [(house04_bytes, 3000, True), (synth.encode_mesh(pos, faces, nrm, uv), 1500, True),
             (synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(force_scheme=0, single_connectivity=1)), 1500, True),
             (synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(normal_prediction=6)), 1500, True),
             (synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=4, uv_prediction=5)), 1500, True),
             (synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(traversal_method=2)), 1500, False),
             (synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(predictive_connectivity=1)), 1500, False),
             (synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(predictive_connectivity=2)), 1500, False),
             (synth.encode_mesh_sequential(pos, faces, nrm, uv, compressed=True), 1000, False),
             (synth.encode_mesh_sequential(pos, faces, nrm, uv, compressed=False), 1000, False)]
    for k, (data, iters, force) in enumerate(cases):
        src = tmp_path / ("fuzz%d.drc" % k)
        src.write_bytes(data)
        r = subprocess.run([exe, "fuzz", str(src), str(iters), str(17 + k)] + (["force"] if force else []), capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        counts = dict(zip(r.stdout.split()[::2], map(int, r.stdout.split()[1::2])))
        assert counts["ok"] + counts["invalid"] + counts["notimpl"] + counts["notgeneral"] == iters and counts["invalid"] > 0


def test_the_sizing_parse_sets_the_multi_parallelogram_records_aside(exe, tmp_path):
    """ConstrainedMultiParallelogram on the fast kernels needs 48 bytes per entry + the crease flags; the one scheme byte the host
    parse can reach is the first attribute's, and only a stream that shows method 4 there gets the region (layout_mesh: mp_att)."""
    pos, nrm, uv, faces = synth.make_mesh(synth.HOLES, 20, 16, 3)
    def layout(data):
        src = tmp_path / "l.drc"
        src.write_bytes(data)
        r = subprocess.run([exe, "layout", str(src), "-"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        w = r.stdout.split()
        return dict(zip(w[::2], (int(x) for x in w[1::2])))
    with_mp = layout(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=4, uv_prediction=5)))
    # positions (attribute 0) and texture coordinates (attribute 2: integer / quantised, on the position connectivity); not the
    # octahedral normals (attribute 1)
    assert with_mp["status"] == 0 and with_mp["first_method"] == 4 and with_mp["mp_att"] == 0b101 and with_mp["tc0"] != 0
    assert with_mp["tc0_bytes"] >= 48 * with_mp["cap_vertices"] + with_mp["cap_vertices"] // 2
    plain = layout(synth.encode_mesh(pos, faces, nrm, uv))
    assert plain["first_method"] == 1 and plain["mp_att"] == 0 and plain["tc0"] == 0 and plain["end"] < with_mp["end"]
    later = layout(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(uv_prediction=4)))       # the scheme on a later attribute: not seen
    assert later["mp_att"] == 0
    general = layout(synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(pos_prediction=4, traversal_method=1)))
    assert general["general"] == 1 and general["mp_att"] == 0                                          # prediction-degree order: the general path
