"""Helpers shared by the parity tests: quantisation of source meshes exactly as the
stream writer does it, and order-independent comparison of decoded meshes."""
import numpy as np


def quantize(vals, qmin, qrange, bits):
    """float32 replica of AttributeQuantizationTransform / Quantizer."""
    vals = np.asarray(vals, np.float32)
    mn = np.asarray(qmin, np.float32)
    inv = np.float32(np.float32((1 << bits) - 1) / np.float32(qrange))
    return np.floor(((vals - mn).astype(np.float32) * inv).astype(np.float32) + np.float32(0.5)).astype(np.int64)


def canon_face(keys):
    r = [tuple(keys[i:] + keys[:i]) for i in range(3)]
    return min(r)


def face_multiset(faces, per_corner_keys):
    """faces: (F,3) indices into per_corner_keys rows -> sorted list of rotation-canonical faces."""
    out = []
    for f in faces:
        out.append(canon_face([tuple(int(x) for x in per_corner_keys[i]) for i in f]))
    out.sort()
    return out


def decoded_point_keys(mesh, attr_ids=None):
    """Per-point integer key = concatenated portable values of the chosen attributes."""
    atts = mesh.attributes if attr_ids is None else [mesh.attributes[i] for i in attr_ids]
    cols = []
    for a in atts:
        if a.portable is None:
            continue
        cols.append(a.portable[a.point_map] if len(a.point_map) else a.portable)
    return np.concatenate(cols, axis=1)


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def metadata_element(entries=(), subs=()):
    """One metadata element as the bitstream writes it (Metadata/MetadataEncoder.cs; value sizes are varints):
    entries = [(key, value)], subs = [(key, element_bytes)]."""
    out = bytearray(_varint(len(entries)))
    for k, v in entries:
        out += bytes([len(k)]) + k + _varint(len(v)) + v
    out += _varint(len(subs))
    for k, e in subs:
        out += bytes([len(k)]) + k + e
    return bytes(out)


def with_metadata(stream, attribute_elements, file_element):
    """Inserts a metadata block behind the 11-byte header of a .drc stream and sets header flag 0x8000.
    attribute_elements = [(attribute unique id, element_bytes)]."""
    s = bytearray(stream)
    assert s[:5] == b"DRACO" and not s[10] & 0x80
    s[10] |= 0x80
    block = bytearray(_varint(len(attribute_elements)))
    for att_id, e in attribute_elements:
        block += _varint(att_id) + e
    block += file_element
    return bytes(s[:11]) + bytes(block) + bytes(s[11:]), bytes(block)


def raw_point_cloud_stream(num_points, attributes, seed=0):
    """A sequential point cloud written by hand (DracoDecoder.cs:44-64 header, PointCloud sequential decoder,
    AttributesDecoder.cs:19-63 descriptors, SequentialAttributeDecoder.cs:75-86 raw values): one attributes decoder,
    every attribute a generic one (decoder type 0) whose values are stored as they are.
    attributes = [(attribute type, data type id, components)]; returns (stream, [value arrays])."""
    import numpy as np
    dtypes = {1: np.int8, 2: np.uint8, 3: np.int16, 4: np.uint16, 5: np.int32, 6: np.uint32, 9: np.float32}
    rng = np.random.default_rng(seed)
    out = bytearray(b"DRACO") + bytes([2, 2, 0, 0, 0, 0]) + int(num_points).to_bytes(4, "little") + bytes([1])
    out += _varint(len(attributes))
    for uid, (att_type, data_type, nc) in enumerate(attributes):
        out += bytes([att_type, data_type, nc, 0]) + _varint(uid)
    out += bytes([0] * len(attributes))
    values = []
    for att_type, data_type, nc in attributes:
        dt = np.dtype(dtypes[data_type])
        v = (rng.integers(0, 200, (num_points, nc)).astype(dt) if dt.kind != "f" else rng.uniform(-1, 1, (num_points, nc)).astype(dt))
        values.append(v)
        out += v.tobytes()
    return bytes(out), values
