"""Helpers shared by the parity tests: quantisation of source meshes exactly as the
stream writer does it, and order-independent comparison of decoded meshes."""
import numpy as np


def quantize(vals, qmin, qrange, bits):
    """float32 replica of AttributeQuantizationTransform / Quantizer."""
    vals = np.asarray(vals, np.float32)
    mn = np.asarray(qmin, np.float32)
    inv = np.float32(np.float32((1 << bits) - 1) / np.float32(qrange))
    return np.floor(((vals - mn).astype(np.float32) * inv).astype(np.float32) + np.float32(0.5)).astype(np.int64)


def source_quantization(vals, bits):
    """(min[c], range, quantised values) of a float32 source array, derived from the source alone by the rules of
    AttributeQuantizationTransform.cs:66-108,136-177 + Core/Quantizer.cs:12-21 (SURVEY.md App. D "Quantise"): per-component
    minimum, range = largest component extent (1 if 0), q = floor((v - min) * (max_q / range) + 0.5) in float32."""
    vals = np.asarray(vals, np.float32)
    mn = vals.min(axis=0).astype(np.float32)
    rng = np.float32((vals.max(axis=0).astype(np.float32) - mn).max())
    if rng == 0:
        rng = np.float32(1.0)
    return mn, rng, quantize(vals, mn, rng, bits)


def oct_quantize(normals, bits):
    """Octahedral quantisation of float normals, restated in numpy from OctahedronToolBox.cs:28-119 (SURVEY.md App. D
    "Octahedral quantise"): scale to L1 norm 1 (double arithmetic), x and y rounded times the centre value, z by the
    remainder with the two fix-ups, then (s, t) with the canonicalisation of the edge cases.  Independent of the stream
    writer's C++ (draco-sharp_amd/csrc/dsa_encode_host.h)."""
    v = np.asarray(normals, np.float64)
    max_value = (1 << bits) - 2
    center = max_value // 2
    abs_sum = np.abs(v).sum(axis=1)
    ok = abs_sum > 1e-6
    sv = np.where(ok[:, None], v * (1.0 / np.where(ok, abs_sum, 1.0))[:, None], np.array([1.0, 0.0, 0.0]))
    i0 = np.floor(sv[:, 0] * center + 0.5).astype(np.int64)
    i1 = np.floor(sv[:, 1] * center + 0.5).astype(np.int64)
    i2 = center - np.abs(i0) - np.abs(i1)
    neg = i2 < 0
    i1 = np.where(neg, np.where(i1 > 0, i1 + i2, i1 - i2), i1)
    i2 = np.where(neg, 0, i2)
    i2 = np.where(sv[:, 2] < 0, -i2, i2)
    s = np.where(i0 >= 0, i1 + center, np.where(i1 < 0, np.abs(i2), max_value - np.abs(i2)))
    t = np.where(i0 >= 0, i2 + center, np.where(i2 < 0, np.abs(i1), max_value - np.abs(i1)))
    # CanonicalizeOctahedralCoords, first matching rule wins
    out_s, out_t = s.copy(), t.copy()
    done = np.zeros(len(s), bool)
    r = ((s == 0) & (t == 0)) | ((s == 0) & (t == max_value)) | ((s == max_value) & (t == 0))
    out_s[r], out_t[r] = max_value, max_value
    done |= r
    r = ~done & (s == 0) & (t > center)
    out_t[r] = center - (t[r] - center)
    done |= r
    r = ~done & (s == max_value) & (t < center)
    out_t[r] = center + (center - t[r])
    done |= r
    r = ~done & (t == max_value) & (s < center)
    out_s[r] = center + (center - s[r])
    done |= r
    r = ~done & (t == 0) & (s > center)
    out_s[r] = center - (s[r] - center)
    return np.stack([out_s, out_t], axis=1)


def source_corner_faces(pos, normals, uvs, faces, pos_bits=11, normal_bits=8, uv_bits=10):
    """The quantised mesh a conformant codec must reproduce, from the INPUT alone: per vertex the key (quantised
    position, octahedral normal, quantised texture coordinate), per face its three keys up to rotation, as a sorted
    list.  Also returns the quantisation parameters the stream has to carry."""
    pmin, prange, qp = source_quantization(pos, pos_bits)
    umin, urange, qu = source_quantization(uvs, uv_bits)
    keys = np.concatenate([qp, oct_quantize(normals, normal_bits), qu], axis=1)
    return face_multiset_fast(faces, keys), (pmin, prange, umin, urange)


def face_multiset_fast(faces, per_point_keys):
    """Sorted rotation-canonical faces as an int64 array [F, 3 * K] (numpy throughout: 64k-triangle meshes)."""
    k = np.asarray(per_point_keys, np.int64)[np.asarray(faces, np.int64)]            # [F, 3, K]
    rots = np.stack([k.reshape(len(k), -1), np.roll(k, -1, axis=1).reshape(len(k), -1), np.roll(k, -2, axis=1).reshape(len(k), -1)], axis=1)   # [F, 3, 3K]
    # lexicographically smallest rotation per face
    best = rots[:, 0]
    for j in (1, 2):
        cand = rots[:, j]
        diff = cand != best
        first = np.where(diff.any(axis=1), diff.argmax(axis=1), 0)
        less = diff.any(axis=1) & (cand[np.arange(len(cand)), first] < best[np.arange(len(best)), first])
        best = np.where(less[:, None], cand, best)
    order = np.lexsort(best.T[::-1])
    return best[order]


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


# ---------------------------------------------------------------- attribute seams
def chart_of_faces(pos, faces, how, seed=0):
    """A chart id per face, from the face centroids in the bounding box of the mesh: the edges between charts become
    attribute seams.  how: 'stripes' (seams from boundary to boundary / closed loops on closed meshes), 'island' (one
    closed seam loop in the interior), 'checker' (seams crossing each other), 'random' (almost every edge a seam),
    'single' (one lone face cut out), 'none' (one chart: ids per corner but no seam)."""
    pos = np.asarray(pos, np.float64)
    c = pos[np.asarray(faces, np.int64)].mean(axis=1)
    lo, hi = pos.min(axis=0), pos.max(axis=0)
    u = (c - lo) / np.where(hi > lo, hi - lo, 1.0)
    if how == "stripes":
        return np.minimum((u[:, 0] * 3).astype(np.int64), 2)
    if how == "island":
        return (((u[:, 0] - 0.5) ** 2 + (u[:, 1] - 0.5) ** 2) < 0.07).astype(np.int64)
    if how == "checker":
        return (np.minimum((u[:, 0] * 4).astype(np.int64), 3) + np.minimum((u[:, 1] * 3).astype(np.int64), 2)) % 2 + \
            2 * (u[:, 2] > 0.5)
    if how == "random":
        return np.random.default_rng(seed).integers(0, 3, len(faces))
    if how == "single":
        r = np.zeros(len(faces), np.int64)
        r[len(faces) // 2] = 1
        return r
    if how == "none":
        return np.zeros(len(faces), np.int64)
    raise ValueError(how)


def split_by_chart(faces, values, chart, shift):
    """Per-corner value ids for an attribute that is continuous inside a chart and jumps between charts: one value row
    per (vertex, chart) pair in use, the vertex's value moved by chart * shift.  Returns (ids[F,3] uint32, rows float32)."""
    faces = np.asarray(faces, np.int64)
    chart = np.asarray(chart, np.int64)
    n = int(chart.max()) + 1
    key = faces * n + chart[:, None]
    uniq, inv = np.unique(key.ravel(), return_inverse=True)
    rows = np.asarray(values, np.float32)[uniq // n] + (uniq % n)[:, None].astype(np.float32) * np.asarray(shift, np.float32)[None, :]
    return inv.reshape(faces.shape).astype(np.uint32), rows.astype(np.float32)


def source_corner_faces_seamed(pos, faces, normals, normal_ids, uvs, uv_ids, pos_bits=11, normal_bits=8, uv_bits=10):
    """source_corner_faces for attributes given per corner (ids None: per vertex): the key of a corner is (quantised
    position of its vertex, octahedral normal of its normal row, quantised texture coordinate of its uv row)."""
    faces = np.asarray(faces, np.int64)
    pmin, prange, qp = source_quantization(pos, pos_bits)
    cols = [qp[faces.ravel()]]
    if normals is not None:
        qn = oct_quantize(normals, normal_bits)
        cols.append(qn[(faces if normal_ids is None else np.asarray(normal_ids, np.int64)).ravel()])
    umin = urange = None
    if uvs is not None:
        umin, urange, qu = source_quantization(uvs, uv_bits)
        cols.append(qu[(faces if uv_ids is None else np.asarray(uv_ids, np.int64)).ravel()])
    keys = np.concatenate(cols, axis=1)
    return face_multiset_fast(np.arange(len(keys)).reshape(-1, 3), keys), (pmin, prange, umin, urange)


def seamed_mesh(synth, kind, nx, ny, seed, normal_charts=None, uv_charts="stripes"):
    """A synthetic mesh with its normals and / or texture coordinates given per corner (chart_of_faces patterns; None:
    that attribute stays per vertex).  Returns the arguments of synth.encode_mesh_corners:
    (pos, faces, normal rows, normal ids or None, uv rows, uv ids or None)."""
    pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, seed)
    nid = uid = None
    if normal_charts:
        nid, nrm = split_by_chart(faces, nrm, chart_of_faces(pos, faces, normal_charts, seed + 1), [0.3, -0.2, 0.1])
    if uv_charts:
        uid, uv = split_by_chart(faces, uv, chart_of_faces(pos, faces, uv_charts, seed + 2), [1.25, 0.5])
    return pos, faces, nrm, nid, uv, uid
