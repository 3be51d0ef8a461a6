"""Host-side mirror of draco-sharp's decode surface over the C-ABI.

Names, argument meaning and error behaviour follow the reference:
  DracoDecoder.Decode(path | stream | bytes)      src/Draco/IO/DracoDecoder.cs:8-42
  Draco{Header, Metadata, ConnectedData, Attributes}   src/Draco/Draco.cs:9-15
  DracoHeader                                      src/Draco/DracoHeader.cs:5-23
  Mesh : PointCloud                                src/Draco/IO/Mesh/Mesh.cs:15-69, IO/PointCloud/PointCloud.cs:11-133
  PointAttribute : GeometryAttribute               src/Draco/IO/Attributes/PointAttribute.cs:5-63
Malformed streams raise InvalidDataException, unsupported stream features
NotImplementedException (src/Draco/IO/Extensions/Assertions.cs:5-24, DracoDecoder.cs:70).
The C# binding of the same C-ABI is in csharp/ (it cannot be compiled in this image)."""
import ctypes as C
import io

import numpy as np

from . import native

_DT_NUMPY = {1: np.int8, 2: np.uint8, 3: np.int16, 4: np.uint16, 5: np.int32, 6: np.uint32,
             7: np.int64, 8: np.uint64, 9: np.float32, 10: np.float64, 11: np.uint8}


class InvalidDataException(Exception):
    """System.IO.InvalidDataException"""


class DeviceException(RuntimeError):
    """HIP runtime failure / missing GPU library"""


def _raise(status, msg):
    if status == native.DSA_ERR_INVALID_DATA:
        raise InvalidDataException(msg)
    if status == native.DSA_ERR_NOT_IMPLEMENTED:
        raise NotImplementedError(msg)
    if status == native.DSA_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    if status == native.DSA_ERR_OUT_OF_MEMORY:
        raise MemoryError(msg)
    raise DeviceException(msg)


class DracoHeader:
    def __init__(self, info):
        self.MajorVersion = info.major_version
        self.MinorVersion = info.minor_version
        self.EncoderType = info.encoder_type
        self.EncoderMethod = info.encoder_method
        self.Flags = info.flags

    @property
    def Version(self):
        return (self.MajorVersion << 8) | self.MinorVersion


class DataBuffer:
    """Core/DataBuffer.cs:5-111 -- byte store of an attribute; here a numpy view of the decoded values."""

    def __init__(self, values):
        self._values = values

    @property
    def DataSize(self):
        return self._values.nbytes

    def Read(self, dtype, byte_offset):
        return np.frombuffer(self._values.tobytes(), dtype=dtype, count=1, offset=byte_offset)[0]

    def AsArray(self):
        return self._values


class PointAttribute:
    def __init__(self, info, values, point_map, portable):
        self.AttributeType = info.attribute_type
        self.DataType = info.data_type
        self.NumComponents = info.num_components
        self.Normalized = bool(info.normalized)
        self.ByteStride = info.byte_stride
        self.ByteOffset = 0
        self.UniqueId = info.unique_id
        self.UniqueEntriesCount = info.num_entries
        self.IsMappingIdentity = False          # mesh attributes always carry an explicit map (MeshTraversalSequencer.cs:35)
        self.DecoderType = info.decoder_type
        self.PredictionMethod = info.prediction_method
        self.PredictionTransform = info.prediction_transform
        self.QuantizationBits = info.quantization_bits
        self.Range = info.range
        self.MinValues = list(info.min_values)[: info.num_components]
        self.Buffer = DataBuffer(values)
        self.Values = values                    # (entries, components) in traversal order
        self.PointMap = point_map               # point -> entry
        self.PortableValues = portable          # int32 (entries, portable components) or None

    def MappedIndex(self, point):
        return int(self.PointMap[point])

    def GetValue(self, entry):
        return self.Values[entry]


class PointCloud:
    def __init__(self, attributes, num_points):
        self.Attributes = attributes
        self.PointsCount = num_points

    def GetNamedAttributeId(self, attribute_type, i=0):
        ids = [k for k, a in enumerate(self.Attributes) if a.AttributeType == attribute_type]
        return ids[i] if i < len(ids) else -1

    def GetNamedAttribute(self, attribute_type, i=0):
        k = self.GetNamedAttributeId(attribute_type, i)
        return None if k < 0 else self.Attributes[k]

    def GetAttributeById(self, k):
        return self.Attributes[k]

    def GetAttributeByUniqueId(self, uid):
        for a in self.Attributes:
            if a.UniqueId == uid:
                return a
        return None


class Mesh(PointCloud):
    def __init__(self, attributes, num_points, faces):
        super().__init__(attributes, num_points)
        self.Faces = faces                      # (F,3) int32 point ids

    @property
    def FacesCount(self):
        return len(self.Faces)

    def GetFace(self, f):
        return [int(x) for x in self.Faces[f]]

    def CornerToPointId(self, corner):
        return int(self.Faces[corner // 3][corner % 3])


class MetadataElement:
    """src/Draco/IO/Metadata/MetadataElement.cs:3-10: keys and values as byte strings, nested elements by key."""
    def __init__(self):
        self.Id = None
        self.Keys = []
        self.Values = []
        self.SubMetadataKeys = []
        self.SubMetadata = []

    def GetEntry(self, key):
        key = key.encode() if isinstance(key, str) else bytes(key)
        for k, v in zip(self.Keys, self.Values):
            if k == key:
                return v
        return None


class DracoMetadata:
    """src/Draco/DracoMetadata.cs:5-9"""
    def __init__(self, attributes, file):
        self.Attributes = attributes
        self.File = file


def parse_metadata(block):
    """The metadata block of a stream (Metadata/MetadataDecoder.cs:5-49; value sizes are varints as the bitstream
    writes them, where the C# reads one byte) -> DracoMetadata.  Host-side: the decode path only skips the block."""
    data = bytes(block)
    pos = 0

    def u8():
        nonlocal pos
        if pos >= len(data):
            raise InvalidDataException("metadata block truncated")
        pos += 1
        return data[pos - 1]

    def varint():
        r, shift = 0, 0
        while True:
            b = u8()
            r |= (b & 0x7F) << shift
            if not b & 0x80:
                return r
            shift += 7
            if shift > 63:
                raise InvalidDataException("metadata varint too long")

    def take(k):
        nonlocal pos
        if k > len(data) - pos:
            raise InvalidDataException("metadata block truncated")
        pos += k
        return data[pos - k:pos]

    def element(depth):
        if depth > 15:
            raise InvalidDataException("metadata nesting too deep")
        e = MetadataElement()
        for _ in range(varint()):
            e.Keys.append(take(u8()))
            e.Values.append(take(varint()))
        for _ in range(varint()):
            e.SubMetadataKeys.append(take(u8()))
            e.SubMetadata.append(element(depth + 1))
        return e

    atts = []
    for _ in range(varint()):
        att_id = varint()
        e = element(0)
        e.Id = att_id
        atts.append(e)
    md = DracoMetadata(atts, element(0))
    if pos != len(data):
        raise InvalidDataException("metadata block has trailing bytes")
    return md


class Draco:
    def __init__(self, header, connected, metadata=None):
        self.Header = header
        self.Metadata = metadata
        self.ConnectedData = connected
        self.Attributes = connected.Attributes


class Context:
    """One GPU.  stream: optional hipStream_t handle (int) to run on."""

    def __init__(self, device=0, stream=None):
        self._L = native.lib()
        h = C.c_void_p()
        st = self._L.dsa_context_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if st != 0:
            raise DeviceException("dsa_context_create(device=%d) failed with status %d (is a GPU visible?)" % (device, st))
        self._h = h
        self.device = device

    def error(self):
        return self._L.dsa_last_error(self._h).decode()

    def set_profiling(self, on=True):
        self._L.dsa_context_set_profiling(self._h, 1 if on else 0)

    def schedule_note(self):
        """What the context found when it checked its schedule's assumptions (register counts behind k_register_gate)."""
        return self._L.dsa_context_schedule_note(self._h).decode()

    def trim(self):
        """Releases the arenas, pinned mirrors and encoder lanes the context keeps between calls."""
        self._L.dsa_context_trim(self._h)

    def close(self):
        if self._h:
            self._L.dsa_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """A batch of independent .drc streams resident on one GPU."""

    def __init__(self, ctx, streams=None, blob=None, offsets=None):
        self.ctx = ctx
        L = self._L = ctx._L
        h = C.c_void_p()
        if blob is not None:
            blob = np.ascontiguousarray(blob, np.uint8)
            offsets = np.ascontiguousarray(offsets, np.uint64)
            n = len(offsets) - 1
            st = L.dsa_batch_create_packed(ctx._h, n, blob.ctypes.data, offsets.ctypes.data, C.byref(h))
        else:
            n = len(streams)
            bufs = [(C.c_uint8 * max(1, len(s))).from_buffer_copy(s if len(s) else b"\0") for s in streams]
            ptrs = (C.c_void_p * max(1, n))(*[C.addressof(b) for b in bufs])
            lens = (C.c_size_t * max(1, n))(*[len(s) for s in streams])
            st = L.dsa_batch_create(ctx._h, n, ptrs, lens, C.byref(h))
        if st != 0:
            _raise(st, ctx.error())
        self._h = h
        self.n = n

    def decode(self, wait=True):
        st = self._L.dsa_batch_decode(self._h)
        if st != 0:
            _raise(st, self.ctx.error())
        if wait:
            self.wait()

    def wait(self):
        st = self._L.dsa_batch_wait(self._h)
        if st != 0:
            _raise(st, self.ctx.error())

    def download(self, wait=True, compact=False):
        """Queues ONE device -> host transfer of every output array of the batch (faces, attribute values, point maps) into a
        pinned mirror the library owns, behind the batch's kernels (dsa_batch_download); result() and host_views() then read the
        mirror instead of copying array by array."""
        # compact=True (dsa_batch_download_compact): less on the link -- faces as uint16 where a mesh has at most 65 536 points, one
        # point map per distinct map; host_views() then returns the arrays as they were stored (see there), result() widens.
        st = (self._L.dsa_batch_download_compact if compact else self._L.dsa_batch_download)(self._h, None, 0)
        if st != 0:
            _raise(st, self.ctx.error())
        if wait:
            self.wait()

    @property
    def output_bytes(self):
        return int(self._L.dsa_batch_output_bytes(self._h))

    @property
    def compact_bytes(self):
        return int(self._L.dsa_batch_compact_bytes(self._h))

    def host_views(self, i):
        """Zero-copy numpy views of mesh i's arrays in the downloaded output block: {"faces": int32[F, 3], "attributes":
        [{"values": [entries, components], "point_map": uint32[points]}]}.  Valid until the batch is closed or decoded again."""
        L = self._L
        info = self.mesh_info(i)
        if info.status != 0:
            _raise(info.status, "stream %d: decode failed (status %d, site %d)" % (i, info.status, info.detail))
        lay = native.MeshOutput()
        st = L.dsa_batch_output_layout(self._h, i, C.byref(lay))
        if st != 0:
            _raise(st, self.ctx.error())
        base = L.dsa_batch_host_output(self._h, lay.block)
        if not base:
            raise RuntimeError("the batch has not been downloaded (Batch.download)")

        def view(off, count, dtype, shape):
            if count == 0:
                return np.zeros(shape, dtype)
            nbytes = count * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_uint8 * nbytes).from_address(base + off), dtype, count).reshape(shape)

        # after a compact download: faces may be uint16; attributes decoded in one order share one map array; a point cloud's
        # identity map is not stored ("point_map": None)
        fdt = np.uint16 if lay.flags & native.DSA_OUTPUT_FACES_U16 else np.int32
        out = {"faces": view(lay.faces, info.num_faces * 3, fdt, (info.num_faces, 3)), "attributes": []}
        for a in range(info.num_attributes):
            ai = native.AttributeInfo()
            st = L.dsa_batch_attribute_info(self._h, i, a, C.byref(ai))
            if st != 0:
                _raise(st, self.ctx.error())
            out["attributes"].append({
                "info": ai,
                "values": view(lay.values[a], ai.num_entries * ai.num_components, _DT_NUMPY[ai.data_type], (ai.num_entries, ai.num_components)),
                "point_map": None if lay.point_map[a] == native.NO_MAP else view(lay.point_map[a], info.num_points, np.uint32, (info.num_points,))})
        return out

    @property
    def algorithmic_bytes(self):
        return int(self._L.dsa_batch_algorithmic_bytes(self._h))

    @property
    def arena_bytes(self):
        return int(self._L.dsa_batch_arena_bytes(self._h))

    def stage_times(self):
        ms = (C.c_float * native.DSA_NUM_STAGES)()
        names = (C.c_char_p * native.DSA_NUM_STAGES)()
        self._L.dsa_batch_stage_times(self._h, C.byref(ms), C.byref(names))
        return {names[i].decode(): float(ms[i]) for i in range(native.DSA_NUM_STAGES)}

    def kernel_times(self):
        """{kernel name: ms} of the step's main kernels, each from its own event pair on its own stream (profiling on)."""
        cap = 32
        ms = (C.c_float * cap)()
        names = (C.c_char_p * cap)()
        count = C.c_uint32()
        self._L.dsa_batch_kernel_times(self._h, ms, names, cap, C.byref(count))
        return {names[i].decode(): float(ms[i]) for i in range(min(cap, count.value))}

    def mesh_info(self, i):
        info = native.MeshInfo()
        st = self._L.dsa_batch_mesh_info(self._h, i, C.byref(info))
        if st != 0:
            _raise(st, self.ctx.error())
        return info

    def status(self, i):
        return self.mesh_info(i).status

    def debug_array(self, i, what, dtype, count):
        out = np.zeros(count, dtype)
        written = C.c_size_t()
        st = self._L.dsa_batch_copy_debug(self._h, i, what, out.ctypes.data, out.nbytes, C.byref(written))
        if st != 0:
            _raise(st, self.ctx.error())
        return out[: written.value // out.itemsize]

    def result(self, i):
        """Draco object of mesh i; raises what the reference would for a bad stream."""
        L = self._L
        info = self.mesh_info(i)
        if info.status != 0:
            _raise(info.status, "stream %d: decode failed (status %d, site %d)" % (i, info.status, info.detail))
        faces = np.zeros((info.num_faces, 3), np.int32)
        if info.num_faces:
            st = L.dsa_batch_copy_faces(self._h, i, faces.ctypes.data)
            if st != 0:
                _raise(st, self.ctx.error())
        atts = []
        for a in range(info.num_attributes):
            ai = native.AttributeInfo()
            st = L.dsa_batch_attribute_info(self._h, i, a, C.byref(ai))
            if st != 0:
                _raise(st, self.ctx.error())
            vals = np.zeros((ai.num_entries, ai.num_components), _DT_NUMPY[ai.data_type])
            pmap = np.zeros(info.num_points, np.uint32)
            if ai.num_entries:
                st = L.dsa_batch_copy_attribute_values(self._h, i, a, vals.ctypes.data)
                if st != 0:
                    _raise(st, self.ctx.error())
            if info.num_points:
                st = L.dsa_batch_copy_point_map(self._h, i, a, pmap.ctypes.data)
                if st != 0:
                    _raise(st, self.ctx.error())
            portable = None
            if ai.decoder_type != 0:
                ncp = 2 if ai.decoder_type == 3 else ai.num_components
                portable = np.zeros((ai.num_entries, ncp), np.int32)
                if ai.num_entries:
                    st = L.dsa_batch_copy_portable_values(self._h, i, a, portable.ctypes.data)
                    if st != 0:
                        _raise(st, self.ctx.error())
            atts.append(PointAttribute(ai, vals, pmap, portable))
        metadata = None
        if info.flags & 0x8000:         # DracoDecoder.cs:23-28
            n = C.c_size_t(0)
            st = L.dsa_batch_copy_metadata(self._h, i, None, 0, C.byref(n))
            if st != 0:
                _raise(st, self.ctx.error())
            buf = (C.c_uint8 * max(1, n.value))()
            st = L.dsa_batch_copy_metadata(self._h, i, buf, n.value, C.byref(n))
            if st != 0:
                _raise(st, self.ctx.error())
            metadata = parse_metadata(bytes(buf)[:n.value])
        if info.encoder_type == 0:      # EncodedGeometryType.PointCloud (Constants.cs): no faces
            return Draco(DracoHeader(info), PointCloud(atts, info.num_points), metadata)
        return Draco(DracoHeader(info), Mesh(atts, info.num_points, faces), metadata)

    def device_views(self, i):
        """Zero-copy views of mesh i's results in the batch arena (dsa_batch_device_*), for consumers that stay on the
        GPU: {"faces": int32[F, 3] point ids, "attributes": [{"values": [entries, components], "point_map": uint32 as
        int32[points]}]} as torch tensors on the context's device.  Valid until the batch is closed or decoded again.
        (torch.cuda must have been initialised before the first Context of the process: its bundled HIP runtime does
        not come up behind the library's.)"""
        import torch
        L = self._L
        info = self.mesh_info(i)
        if info.status != 0:
            _raise(info.status, "stream %d: decode failed (status %d, site %d)" % (i, info.status, info.detail))

        class _Ptr:                                    # __cuda_array_interface__ v2 carrier
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}

        def view(ptr, shape, typestr):
            if ptr is None or 0 in shape:
                return torch.empty(shape, dtype=torch.int32 if typestr == "<i4" else None, device="cuda:%d" % self.ctx.device)
            return torch.as_tensor(_Ptr(ptr, shape, typestr), device="cuda:%d" % self.ctx.device)

        typestr = {1: "|i1", 2: "|u1", 3: "<i2", 4: "<u2", 5: "<i4", 6: "<u4", 9: "<f4"}
        out = {"faces": view(L.dsa_batch_device_faces(self._h, i), (info.num_faces, 3), "<i4"), "attributes": []}
        for a in range(info.num_attributes):
            ai = native.AttributeInfo()
            st = L.dsa_batch_attribute_info(self._h, i, a, C.byref(ai))
            if st != 0:
                _raise(st, self.ctx.error())
            ts = typestr.get(ai.data_type)
            if ts in (None, "<u2", "<u4"):           # torch has no unsigned 16 / 32 bit tensors: same bits, signed view
                ts = {"<u2": "<i2", "<u4": "<i4"}.get(ts)
            values = view(L.dsa_batch_device_attribute_values(self._h, i, a), (ai.num_entries, ai.num_components), ts) if ts else None
            pmap = view(L.dsa_batch_device_point_map(self._h, i, a), (info.num_points,), "<i4")
            out["attributes"].append({"values": values, "point_map": pmap})
        return out

    def close(self):
        if self._h:
            self._L.dsa_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class DracoDecoder:
    """DracoDecoder.Decode: one stream -> Draco (a batch of one on the default GPU)."""

    def __init__(self, context=None):
        self._ctx = context

    def Decode(self, source):
        if isinstance(source, str):
            with open(source, "rb") as f:
                data = f.read()
        elif isinstance(source, (bytes, bytearray, memoryview)):
            data = bytes(source)
        elif isinstance(source, io.IOBase) or hasattr(source, "read"):
            data = source.read()
            if hasattr(source, "close"):
                source.close()      # the reference disposes the caller's reader (DecoderBuffer.cs:186-189)
        else:
            raise TypeError("Decode expects a path, bytes or a binary stream")
        return self.DecodeBatch([data])[0]

    def DecodeBatch(self, streams):
        ctx = self._ctx or default_context()
        b = Batch(ctx, streams)
        try:
            b.decode()
            return [b.result(i) for i in range(b.n)]
        finally:
            b.close()


class _PoolErrors:
    """What a Batch view of a pool job asks its `ctx` for."""

    def __init__(self, pool):
        self._pool = pool
        self.device = None

    def error(self):
        return self._pool.error()


class PoolJob:
    """Results of Pool.decode: every stream was decoded in some chunk on some worker; result(i) / status(i) follow
    dsa_pool_job_locate to the batch that holds stream i."""

    def __init__(self, pool, handle, n):
        self._pool, self._L, self._h, self.n = pool, pool._L, handle, n
        self._views = {}

    def locate(self, i):
        b, m, w = C.c_void_p(), C.c_uint32(), C.c_uint32()
        st = self._L.dsa_pool_job_locate(self._h, i, C.byref(b), C.byref(m), C.byref(w))
        if st != 0:
            _raise(st, "stream index %d out of range" % i)
        return b.value, int(m.value), int(w.value)

    def _view(self, handle, worker):
        v = self._views.get(handle)
        if v is None:
            v = Batch.__new__(Batch)                     # a view: the job owns the batch
            v.ctx = _PoolErrors(self._pool)
            v.ctx.device = self._pool.devices[worker]
            v._L, v._h, v.n = self._L, C.c_void_p(handle), int(self._L.dsa_batch_size(C.c_void_p(handle)))
            v.close = lambda: None
            self._views[handle] = v
        return v

    def status(self, i):
        h, m, w = self.locate(i)
        return self._view(h, w).status(m)

    def worker(self, i):
        return self.locate(i)[2]

    def result(self, i):
        h, m, w = self.locate(i)
        return self._view(h, w).result(m)

    @property
    def chunks(self):
        return int(self._L.dsa_pool_job_chunks(self._h))

    def close(self):
        if self._h:
            for v in self._views.values():
                v._h = None
            self._views = {}
            self._L.dsa_pool_job_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pool:
    """One context + worker thread per listed GPU inside the library (dsa_pool_*): what a single-process host (the C#
    one) uses instead of one process per GPU.  Streams are handed out longest first in chunks of `chunk_meshes`
    through one atomic queue; there is no collective."""

    def __init__(self, devices, chunk_meshes=256):
        self._L = native.lib()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        st = self._L.dsa_pool_create(arr, len(self.devices), chunk_meshes, C.byref(h))
        if st != 0:
            raise DeviceException("dsa_pool_create(devices=%s) failed with status %d (is a GPU visible?)" % (self.devices, st))
        self._h = h

    def error(self):
        return self._L.dsa_pool_last_error(self._h).decode() if self._h else "pool closed"

    def decode(self, streams=None, blob=None, offsets=None):
        if blob is not None:                    # streams back to back in one array: no per-stream copies
            blob = np.ascontiguousarray(blob, np.uint8)
            offsets = np.ascontiguousarray(offsets, np.uint64)
            n = len(offsets) - 1
            addr = (offsets[:-1] + np.uint64(blob.ctypes.data)).astype(np.uint64)
            size = np.diff(offsets).astype(np.uint64)
            ptrs = C.cast(addr.ctypes.data, C.POINTER(C.c_void_p))
            lens = C.cast(size.ctypes.data, C.POINTER(C.c_size_t))
        else:
            n = len(streams)
            bufs = [(C.c_uint8 * max(1, len(s))).from_buffer_copy(s if len(s) else b"\0") for s in streams]
            ptrs = (C.c_void_p * max(1, n))(*[C.addressof(b) for b in bufs])
            lens = (C.c_size_t * max(1, n))(*[len(s) for s in streams])
        h = C.c_void_p()
        st = self._L.dsa_pool_decode(self._h, n, ptrs, lens, C.byref(h))
        if st != 0:
            _raise(st, self.error())
        return PoolJob(self, h, n)

    def close(self):
        if self._h:
            self._L.dsa_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pool_plan(lengths, chunk_meshes):
    """The queue order dsa_pool_decode uses (no GPU needed): (order, chunk_begin)."""
    L = native.lib()
    n = len(lengths)
    lens = (C.c_size_t * max(1, n))(*[int(x) for x in lengths])
    order = (C.c_uint32 * max(1, n))()
    begin = (C.c_uint32 * (n + 1))()
    chunks = L.dsa_pool_plan(n, lens, chunk_meshes, order, begin)
    return list(order[:n]), list(begin[:chunks + 1])
