"""glTF containers in front of the batch path (draco-sharp_amd/gltf.py): GLB chunks, embedded / external / data-URI
buffers, primitives with and without KHR_draco_mesh_compression.  The container tests run anywhere; the decode test
needs the GPU."""
import base64
import json
import struct

import numpy as np
import pytest

import oracle
import draco_sharp_amd as dsa
import draco_sharp_amd.synth as synth
from draco_sharp_amd import gltf


def _streams():
    out = []
    for kind, nx, ny, opt in ((synth.TORUS, 10, 8, {}), (synth.HOLES, 14, 12, {"predictive_connectivity": 2, "uv_prediction": 5, "normal_prediction": 6})):
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, 5)
        out.append((synth.encode_mesh(pos, faces, nrm, uv, opt=synth.options(**opt)), len(pos), len(faces)))
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 6, 5, 5)
    out.append((synth.encode_mesh(pos, faces), len(pos), len(faces)))
    return out


def _document(streams, offsets, buffer_entry):
    accessors, prims = [], []
    for i, (s, nv, nf) in enumerate(streams):
        base = len(accessors)
        sem = {"POSITION": 0} if i == 2 else {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}
        accessors.append({"componentType": 5125, "count": 3 * nf, "type": "SCALAR"})
        for k, name in enumerate(sem):
            accessors.append({"componentType": 5126, "count": nv, "type": "VEC2" if name == "TEXCOORD_0" else "VEC3"})
        prims.append({"attributes": {name: base + 1 + k for k, name in enumerate(sem)}, "indices": base, "mode": 4,
                      "extensions": {gltf.EXTENSION: {"bufferView": i, "attributes": sem}}})
    # an uncompressed primitive must be left alone
    prims.append({"attributes": {"POSITION": 1}, "mode": 4})
    return {"asset": {"version": "2.0"}, "extensionsUsed": [gltf.EXTENSION], "extensionsRequired": [gltf.EXTENSION],
            "buffers": [buffer_entry], "accessors": accessors,
            "bufferViews": [{"buffer": 0, "byteOffset": o, "byteLength": len(s)} for o, (s, _, _) in zip(offsets, streams)],
            "meshes": [{"primitives": prims[:2]}, {"primitives": prims[2:]}]}


def _pack(streams):
    blob, offsets = bytearray(), []
    for s, _, _ in streams:
        while len(blob) % 4:
            blob.append(0)
        offsets.append(len(blob))
        blob += s
    return bytes(blob), offsets


def _glb(doc, blob):
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    binc = blob + b"\0" * (-len(blob) % 4)
    total = 12 + 8 + len(js) + 8 + len(binc)
    return struct.pack("<III", 0x46546C67, 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binc), 0x004E4942) + binc


def _assets(tmp_path):
    streams = _streams()
    blob, offsets = _pack(streams)
    glb = _glb(_document(streams, offsets, {"byteLength": len(blob)}), blob)
    embedded = json.dumps(_document(streams, offsets, {"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()})).encode()
    (tmp_path / "my scene.bin").write_bytes(blob)
    ext_path = tmp_path / "scene.gltf"
    ext_path.write_text(json.dumps(_document(streams, offsets, {"byteLength": len(blob), "uri": "my%20scene.bin"})))
    glb_path = tmp_path / "scene.glb"
    glb_path.write_bytes(glb)
    return streams, [glb, embedded, str(ext_path), glb_path]


def test_containers_yield_the_compressed_streams(tmp_path):
    streams, sources = _assets(tmp_path)
    for src in sources:
        asset = gltf.read_asset(src)
        prims = gltf.draco_primitives(asset)
        assert [(p.mesh, p.primitive) for p in prims] == [(0, 0), (0, 1), (1, 0)]
        assert [p.stream for p in prims] == [s for s, _, _ in streams]
        assert prims[0].attribute_ids == {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2} and prims[2].attribute_ids == {"POSITION": 0}


def test_broken_containers_are_invalid_data(tmp_path):
    streams, sources = _assets(tmp_path)
    glb = sources[0]
    for bad in (glb[:20], glb[:12] + struct.pack("<II", 1 << 30, 0x4E4F534A) + glb[20:], b"not a gltf", b"\xff\xfe\x00"):
        with pytest.raises(dsa.InvalidDataException):
            gltf.read_asset(bad)
    doc = json.loads(sources[1])
    doc["bufferViews"][1]["byteLength"] = 1 << 30
    with pytest.raises(dsa.InvalidDataException):
        gltf.draco_primitives(gltf.read_asset(json.dumps(doc).encode()))
    doc = json.loads(sources[1])
    doc["meshes"][0]["primitives"][0]["mode"] = 1
    with pytest.raises(dsa.InvalidDataException):
        gltf.draco_primitives(gltf.read_asset(json.dumps(doc).encode()))
    doc = json.loads(sources[1])
    doc["buffers"][0]["uri"] = "elsewhere.bin"
    with pytest.raises(dsa.InvalidDataException):
        gltf.read_asset(json.dumps(doc).encode())


def test_external_buffers_cannot_leave_the_asset_directory(tmp_path):
    """An asset is untrusted input: buffer URIs with a scheme, absolute paths, and paths that climb out of the asset's
    directory (plain, percent-encoded or through a symbolic link) are rejected, not opened."""
    import os
    streams, sources = _assets(tmp_path)
    secret = tmp_path.parent / "secret.bin"
    secret.write_bytes(b"x" * 4096)
    sub = tmp_path / "sub"
    sub.mkdir()
    (sub / "inner.bin").write_bytes((tmp_path / "my scene.bin").read_bytes())
    os.symlink(str(secret), str(tmp_path / "link.bin"))
    doc = json.loads((tmp_path / "scene.gltf").read_text())
    for uri in ("../secret.bin", "..%2Fsecret.bin", "%2e%2e/secret.bin", "sub/../../secret.bin", str(secret), "file://" + str(secret),
                "file:secret.bin", "http://example.invalid/a.bin", "link.bin", "/etc/hostname", "."):
        doc["buffers"][0]["uri"] = uri
        (tmp_path / "evil.gltf").write_text(json.dumps(doc))
        with pytest.raises(dsa.InvalidDataException):
            gltf.read_asset(str(tmp_path / "evil.gltf"))
    doc["buffers"][0]["uri"] = "sub/inner.bin"           # a path below the asset's directory is fine
    (tmp_path / "nested.gltf").write_text(json.dumps(doc))
    assert [p.stream for p in gltf.draco_primitives(gltf.read_asset(str(tmp_path / "nested.gltf")))] == [s for s, _, _ in streams]


@pytest.mark.gpu
def test_gpu_loader_decodes_all_primitives_in_one_batch(tmp_path):
    streams, sources = _assets(tmp_path)
    ctx = dsa.Context(0)
    loaded = gltf.GltfDracoLoader(ctx).load(sources)
    assert [len(x) for x in loaded] == [3] * len(sources)
    for per_asset in loaded:
        for prim, (s, nv, nf) in zip(per_asset, streams):
            ref = oracle.decode(s)
            assert prim.indices.dtype == np.uint32 and np.array_equal(prim.indices, ref.faces.reshape(-1).astype(np.uint32))
            by_uid = {a.unique_id: a for a in ref.attributes}
            for semantic, uid in prim.source.attribute_ids.items():
                want = by_uid[uid].values[by_uid[uid].point_map if len(by_uid[uid].point_map) else np.arange(ref.num_points)]
                assert prim.attributes[semantic].shape == (nv, want.shape[1])
                assert prim.attributes[semantic].tobytes() == want.tobytes()
    # accessor / stream disagreement is caught
    doc = json.loads(sources[1])
    doc["accessors"][1]["count"] += 1
    with pytest.raises(dsa.InvalidDataException):
        gltf.GltfDracoLoader(ctx).load([json.dumps(doc).encode()])
    doc = json.loads(sources[1])
    doc["meshes"][0]["primitives"][0]["extensions"][gltf.EXTENSION]["attributes"]["COLOR_0"] = 9
    with pytest.raises(dsa.InvalidDataException):
        gltf.GltfDracoLoader(ctx).load([json.dumps(doc).encode()])
    ctx.close()
