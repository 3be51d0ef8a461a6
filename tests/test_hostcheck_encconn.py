"""The encoder's connectivity kernels (draco-sharp_amd/csrc/dsa_encode_conn.h: corner table, the two walks one lane per mesh,
operand entries) compiled for the host under AddressSanitizer + UBSan (tests/hostcheck/encconn_host.cpp) and held against the
host coder on the same faces: sound meshes of every shape the generator makes, and damaged ones (flipped, rewired, duplicated
faces; random face soups) -- the same results or the same refusal, and no access outside a mesh's arrays.  A check of the product
source on CPU, not a CPU encode path of the product."""
import os
import struct
import subprocess

import numpy as np
import pytest

import draco_sharp_amd.synth as synth

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostcheck", "encconn_host.cpp")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("encconn") / "encconn_host")      # always rebuilt: the sources under test change
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=signed-integer-overflow",
                    "-fno-sanitize-recover=undefined", "-o", out, SRC], check=True)
    return out


def run(exe, tmp_path, meshes):
    path = tmp_path / "meshes.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(meshes)))
        for nv, faces in meshes:
            faces = np.ascontiguousarray(faces, np.uint32).reshape(-1, 3)
            f.write(struct.pack("<II", nv, len(faces)))
            f.write(faces.tobytes())
    r = subprocess.run([exe, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_sound_meshes_of_every_shape(exe, tmp_path):
    meshes = []
    for k, kind in enumerate((synth.GRID, synth.TORUS, synth.SPHERE, synth.HOLES, synth.TWO_PARTS) * 5):
        nx, ny = 4 + (7 * k) % 29, 4 + (5 * k) % 31
        if kind == synth.HOLES: nx, ny = max(nx, 12), max(ny, 12)
        pos, _, _, faces = synth.make_mesh(kind, nx, ny, 40 + k)
        meshes.append((len(pos), faces))
    meshes.append((3, np.array([[0, 1, 2]])))                                   # one triangle
    meshes.append((4, np.array([[0, 1, 2], [0, 2, 3]])))
    meshes.append((4, np.array([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]])))     # a tetrahedron: closed, interior start face
    fan = np.array([[0, i, i + 1] for i in range(1, 200)] + [[0, 200, 1]])      # a vertex of valence 200
    meshes.append((201, fan))
    out = run(exe, tmp_path, meshes)
    assert "%d meshes, %d coded alike, 0 refused alike" % (len(meshes), len(meshes)) in out


def test_damaged_meshes_get_the_host_coders_verdict(exe, tmp_path):
    rng = np.random.default_rng(11)
    meshes = []
    for it in range(600):
        mode = it % 6
        if mode >= 4:                 # faces taken away: new holes, new components, now and then a vertex that is no longer manifold
            p, _, _, f = synth.make_mesh(int(rng.choice([synth.GRID, synth.TORUS, synth.SPHERE, synth.HOLES, synth.TWO_PARTS])), int(rng.integers(4, 20)), int(rng.integers(4, 20)), int(rng.integers(0, 1 << 30)))
            keep = np.ones(len(f), bool); keep[rng.integers(0, len(f), int(rng.integers(1, 12)))] = False
            faces = f[keep]
            used = np.unique(faces)          # vertices left without a face would be isolated: renumber
            remap = np.full(len(p), 0, np.int64); remap[used] = np.arange(len(used))
            meshes.append((len(used), remap[faces]))
            continue
        if mode == 0:
            nv = int(rng.integers(4, 40)); faces = rng.integers(0, nv, (int(rng.integers(1, 80)), 3)).astype(np.uint32)
        else:
            p, _, _, f = synth.make_mesh(int(rng.choice([synth.GRID, synth.TORUS, synth.SPHERE, synth.HOLES])), int(rng.integers(4, 14)), int(rng.integers(4, 14)), int(rng.integers(0, 1 << 30)))
            faces = f.copy(); nv = len(p)
            for _ in range(int(rng.integers(1, 4))):
                k = int(rng.integers(0, len(faces)))
                if mode == 1: faces[k] = faces[k][::-1]
                elif mode == 2: faces[k, int(rng.integers(0, 3))] = int(rng.integers(0, nv + 2))
                else: faces = np.concatenate([faces, faces[k:k + 1][:, [1, 2, 0]] if rng.integers(0, 2) else faces[int(rng.integers(0, len(faces)))][None, ::-1]])
        meshes.append((nv, faces))
    out = run(exe, tmp_path, meshes)
    assert "600 meshes" in out
    coded, refused = [int(x) for x in out.replace(",", "").split() if x.isdigit()][1:3]
    assert coded + refused == 600 and coded > 60 and refused > 200, out
