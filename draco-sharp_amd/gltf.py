"""glTF 2.0 containers in front of the batch decode path: every primitive that carries KHR_draco_mesh_compression in
any number of .gltf / .glb assets becomes one stream of one `Batch` (the caller on the input
side of the path).  Only the container is handled here -- JSON, GLB chunks, buffers and buffer views; what is inside
the buffer view goes to the GPU untouched.

The extension object is `{"bufferView": i, "attributes": {"POSITION": id, ...}}`: the ids are Draco unique ids
(PointCloud.GetAttributeByUniqueId), decoded values are per *point* and the faces are point indices, which is exactly
glTF's vertex / index model.
"""
import base64
import json
import os
import struct

import numpy as np

from .decoder import Batch, InvalidDataException, default_context

EXTENSION = "KHR_draco_mesh_compression"
_GLB_MAGIC, _CHUNK_JSON, _CHUNK_BIN = 0x46546C67, 0x4E4F534A, 0x004E4942
_TYPE_COMPONENTS = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


class GltfAsset:
    """Parsed container: the JSON document and the bytes of every buffer."""

    def __init__(self, doc, buffers, name=""):
        self.doc = doc
        self.buffers = buffers
        self.name = name

    def buffer_view(self, index):
        views = self.doc.get("bufferViews", [])
        if not 0 <= index < len(views):
            raise InvalidDataException("%s: bufferView %d does not exist" % (self.name, index))
        v = views[index]
        b = v.get("buffer", -1)
        if not 0 <= b < len(self.buffers):
            raise InvalidDataException("%s: bufferView %d names buffer %d" % (self.name, index, b))
        off, length = int(v.get("byteOffset", 0)), int(v["byteLength"])
        data = self.buffers[b]
        if off < 0 or length < 0 or off + length > len(data):
            raise InvalidDataException("%s: bufferView %d leaves its buffer" % (self.name, index))
        return bytes(data[off:off + length])


def read_asset(source, base_dir=None, name=None):
    """source: path of a .gltf / .glb file, or the bytes of either.  base_dir resolves relative buffer URIs."""
    if isinstance(source, (str, os.PathLike)):
        path = os.fspath(source)
        with open(path, "rb") as f:
            data = f.read()
        return read_asset(data, base_dir=os.path.dirname(os.path.abspath(path)), name=name or os.path.basename(path))
    data = bytes(source)
    name = name or "<bytes>"
    glb_bin = None
    if len(data) >= 12 and struct.unpack_from("<I", data, 0)[0] == _GLB_MAGIC:
        _, version, total = struct.unpack_from("<III", data, 0)
        if version != 2 or total > len(data) or total < 20:
            raise InvalidDataException("%s: not a GLB 2 container" % name)
        pos, doc = 12, None
        while pos + 8 <= total:
            clen, ctype = struct.unpack_from("<II", data, pos)
            pos += 8
            if clen > total - pos:
                raise InvalidDataException("%s: GLB chunk leaves the file" % name)
            if ctype == _CHUNK_JSON and doc is None:
                doc = json.loads(data[pos:pos + clen].decode("utf-8"))
            elif ctype == _CHUNK_BIN and glb_bin is None:
                glb_bin = data[pos:pos + clen]
            pos += (clen + 3) & ~3
        if doc is None:
            raise InvalidDataException("%s: GLB without a JSON chunk" % name)
    else:
        try:
            doc = json.loads(data.decode("utf-8"))
        except (UnicodeDecodeError, ValueError) as e:
            raise InvalidDataException("%s: neither GLB nor glTF JSON (%s)" % (name, e))
    buffers = []
    for i, b in enumerate(doc.get("buffers", [])):
        uri = b.get("uri")
        if uri is None:
            if i != 0 or glb_bin is None:
                raise InvalidDataException("%s: buffer %d has no uri and there is no GLB binary chunk" % (name, i))
            raw = glb_bin
        elif uri.startswith("data:"):
            head, _, payload = uri.partition(",")
            if not head.endswith(";base64"):
                raise InvalidDataException("%s: buffer %d: only base64 data URIs are supported" % (name, i))
            raw = base64.b64decode(payload)
        else:
            if base_dir is None:
                raise InvalidDataException("%s: buffer %d is external (%s) and no base directory was given" % (name, i, uri))
            raw = _read_external_buffer(name, i, base_dir, uri)
        if len(raw) < int(b.get("byteLength", 0)):
            raise InvalidDataException("%s: buffer %d is shorter than its byteLength" % (name, i))
        buffers.append(raw)
    return GltfAsset(doc, buffers, name)


def _read_external_buffer(name, i, base_dir, uri):
    """An external buffer is a file below the asset's own directory, nothing else: an asset is untrusted input, so
    URIs with a scheme (file:, http:, ...), absolute paths and paths that climb out of base_dir (also percent-encoded
    or through symbolic links) are rejected instead of opened."""
    import re
    from urllib.parse import unquote
    if re.match(r"^[A-Za-z][A-Za-z0-9+.-]*:", uri):
        raise InvalidDataException("%s: buffer %d: URI scheme not supported (%s)" % (name, i, uri.split(":", 1)[0]))
    rel = unquote(uri)
    if os.path.isabs(rel) or rel.startswith(("/", "\\")) or "\x00" in rel:
        raise InvalidDataException("%s: buffer %d: absolute buffer path" % (name, i))
    root = os.path.realpath(base_dir)
    path = os.path.realpath(os.path.join(root, rel))
    if os.path.commonpath([root, path]) != root or path == root:
        raise InvalidDataException("%s: buffer %d: buffer path leaves the asset directory" % (name, i))
    try:
        with open(path, "rb") as f:
            return f.read()
    except OSError as e:
        raise InvalidDataException("%s: buffer %d: cannot read external buffer (%s)" % (name, i, e.strerror))


class DracoPrimitive:
    """One compressed primitive: where it sits in the asset, its stream and the extension's attribute ids."""

    def __init__(self, asset, mesh, primitive, stream, attribute_ids, accessors, indices_accessor, mode):
        self.asset, self.mesh, self.primitive = asset, mesh, primitive
        self.stream = stream
        self.attribute_ids = attribute_ids          # semantic -> Draco unique id
        self.accessors = accessors                  # semantic -> accessor index of the primitive
        self.indices_accessor = indices_accessor
        self.mode = mode


def draco_primitives(asset):
    """Every primitive of the asset that carries the extension, in document order."""
    out = []
    for mi, mesh in enumerate(asset.doc.get("meshes", [])):
        for pi, prim in enumerate(mesh.get("primitives", [])):
            ext = prim.get("extensions", {}).get(EXTENSION)
            if ext is None:
                continue
            mode = prim.get("mode", 4)
            if mode not in (4, 5):                  # the extension allows TRIANGLES and TRIANGLE_STRIP only
                raise InvalidDataException("%s: mesh %d primitive %d: Draco compression with mode %d" % (asset.name, mi, pi, mode))
            stream = asset.buffer_view(int(ext["bufferView"]))
            ids = {k: int(v) for k, v in ext.get("attributes", {}).items()}
            out.append(DracoPrimitive(asset, mi, pi, stream, ids, dict(prim.get("attributes", {})), prim.get("indices"), mode))
    return out


class DecodedPrimitive:
    """indices: uint32[3 * faces] (point ids); attributes: semantic -> array[points, components]."""

    def __init__(self, source, draco, indices, attributes):
        self.source, self.draco, self.indices, self.attributes = source, draco, indices, attributes


class GltfDracoLoader:
    """Decodes the compressed primitives of many assets in one batch on one GPU."""

    def __init__(self, context=None):
        self.ctx = context or default_context()

    def load(self, sources):
        assets = [s if isinstance(s, GltfAsset) else read_asset(s) for s in sources]
        prims = [p for a in assets for p in draco_primitives(a)]
        results = [[] for _ in assets]
        if not prims:
            return results
        batch = Batch(self.ctx, [p.stream for p in prims])
        try:
            batch.decode()
            index_of = {id(a): i for i, a in enumerate(assets)}
            for i, p in enumerate(prims):
                results[index_of[id(p.asset)]].append(self._expand(p, batch.result(i)))
        finally:
            batch.close()
        return results

    @staticmethod
    def _expand(p, draco):
        where = "%s: mesh %d primitive %d" % (p.asset.name, p.mesh, p.primitive)
        geometry = draco.ConnectedData
        if not hasattr(geometry, "Faces"):
            raise InvalidDataException("%s: the stream is a point cloud" % where)
        accessors = p.asset.doc.get("accessors", [])

        def accessor(index, what):
            if index is None:
                return None
            if not 0 <= index < len(accessors):
                raise InvalidDataException("%s: %s accessor %d does not exist" % (where, what, index))
            return accessors[index]

        indices = np.ascontiguousarray(geometry.Faces, np.uint32).reshape(-1)
        acc = accessor(p.indices_accessor, "indices")
        if acc is not None and int(acc.get("count", -1)) != indices.size:
            raise InvalidDataException("%s: indices accessor counts %s, the stream has %d" % (where, acc.get("count"), indices.size))
        attributes = {}
        for semantic, uid in p.attribute_ids.items():
            att = geometry.GetAttributeByUniqueId(uid)
            if att is None:
                raise InvalidDataException("%s: no Draco attribute with unique id %d (%s)" % (where, uid, semantic))
            values = att.Values[att.PointMap]                       # one value per point = per glTF vertex
            acc = accessor(p.accessors.get(semantic), semantic)
            if acc is not None:
                if int(acc.get("count", -1)) != geometry.PointsCount:
                    raise InvalidDataException("%s: %s accessor counts %s, the stream has %d points" % (where, semantic, acc.get("count"), geometry.PointsCount))
                nc = _TYPE_COMPONENTS.get(acc.get("type"))
                if nc is not None and nc != values.shape[1]:
                    raise InvalidDataException("%s: %s accessor is %s, the stream has %d components" % (where, semantic, acc.get("type"), values.shape[1]))
            attributes[semantic] = values
        return DecodedPrimitive(p, draco, indices, attributes)
