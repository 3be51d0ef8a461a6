"""Host-side mirror of the reference's encode surface for the GPU path:

  DracoEncoder.Encode(BinaryWriter, Config, PointCloud, attributes)     src/Draco/IO/DracoEncoder.cs:22-41
  Config (quantisation bits, speed, prediction overrides)              src/Draco/IO/Config.cs

bound to the dsa_encode_* entry points of libdraco_mi355x.so.  Connectivity (corner table, Edgebreaker symbols, attribute
order), attribute quantisation / prediction / rANS coding are HIP kernels; the library's host side chooses the symbol schemes
and lays the bytes out; there is no CPU fallback."""
import ctypes as C
import os
import time

import numpy as np

from . import native
from .decoder import DeviceException, _raise, default_context


class Config:
    """Subset of src/Draco/IO/Config.cs that the device path honours."""

    def __init__(self, position_bits=11, texcoord_bits=10, normal_bits=8, speed=5, single_connectivity=False,
                 symbol_scheme=-1, position_prediction=1, texcoord_prediction=1):
        self.position_bits, self.texcoord_bits, self.normal_bits = position_bits, texcoord_bits, normal_bits
        self.speed = speed                      # compression level = 10 - speed (DracoEncoder.cs:50-56)
        self.single_connectivity = single_connectivity
        self.symbol_scheme = symbol_scheme
        self.position_prediction, self.texcoord_prediction = position_prediction, texcoord_prediction

    def _native(self):
        o = native.EncodeOptions()
        native.lib().dsa_encode_default_options(C.byref(o))
        o.position_bits, o.texcoord_bits, o.normal_bits = self.position_bits, self.texcoord_bits, self.normal_bits
        o.single_connectivity = 1 if self.single_connectivity else 0
        o.symbol_scheme = self.symbol_scheme
        o.compression_level = 10 - self.speed
        o.position_prediction, o.texcoord_prediction = self.position_prediction, self.texcoord_prediction
        return o


class MeshData:
    """Triangle mesh with per-vertex attributes: positions (V,3) f32, faces (F,3) u32, optional normals (V,3), uvs (V,2) and one
    generic uint8 attribute of 1 - 4 components (V,) or (V,C): vertex colours, ids (ABI 4)."""

    def __init__(self, positions, faces, normals=None, texcoords=None, generic=None):
        self.positions = np.ascontiguousarray(positions, np.float32)
        self.faces = np.ascontiguousarray(faces, np.uint32)
        self.normals = None if normals is None else np.ascontiguousarray(normals, np.float32)
        self.texcoords = None if texcoords is None else np.ascontiguousarray(texcoords, np.float32)
        self.generic = None
        if generic is not None:
            g = np.ascontiguousarray(generic, np.uint8)
            g = g.reshape(len(g), -1)
            if len(g) != len(self.positions) or not 1 <= g.shape[1] <= 4:
                raise ValueError("generic attribute: one row of 1 - 4 uint8 components per vertex")
            self.generic = g


class EncodedStreams:
    """The .drc streams of a batch, a sequence of `bytes`.  The bytes stay in the library's buffers until a stream is asked
    for (indexing, iteration), so that a caller who hands the batch on -- to a file, a socket, dsa.Batch -- pays for one copy of
    what it touches instead of for 4096 fresh allocations up front; `sizes` needs none.  A mesh that could not be encoded raises
    when the batch is made, like the reference's encoder does for its one mesh."""

    def __init__(self, ctx, handle, n):
        L = native.lib()
        self._h, self._free = handle, L.dsa_encoded_free
        self._ptr, self._len, self._cache = [0] * n, [0] * n, [None] * n
        p, ln = C.c_void_p(), C.c_size_t()
        try:
            for i in range(n):
                st = L.dsa_encoded_stream(handle, i, C.byref(p), C.byref(ln))
                if st != 0:
                    _raise(st, ctx.error())
                self._ptr[i], self._len[i] = p.value, ln.value
        except Exception:
            self.close()
            raise

    @property
    def sizes(self):
        return list(self._len)

    def __len__(self):
        return len(self._len)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        b = self._cache[i]
        if b is None:
            if self._h is None:
                raise ValueError("the encoded batch was closed")
            b = self._cache[i] = C.string_at(self._ptr[i], self._len[i])
        return b

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __eq__(self, other):
        return len(self) == len(other) and all(a == b for a, b in zip(self, other))

    def close(self):
        """Releases the library's buffers (streams already taken stay valid)."""
        if self._h is not None:
            self._free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class DracoEncoder:
    def __init__(self, context=None):
        self._ctx = context

    def EncodeBatch(self, meshes, config=None):
        """meshes: list of MeshData -> sequence of bytes (.drc streams, EncodedStreams).  A mesh that cannot be encoded raises."""
        ctx = self._ctx or default_context()
        L = native.lib()
        n = len(meshes)
        arr = (native.MeshInput * max(1, n))()
        for i, m in enumerate(meshes):
            arr[i].num_vertices, arr[i].num_faces = len(m.positions), len(m.faces)
            arr[i].positions, arr[i].faces = m.positions.ctypes.data, m.faces.ctypes.data
            arr[i].normals = m.normals.ctypes.data if m.normals is not None else None
            arr[i].texcoords = m.texcoords.ctypes.data if m.texcoords is not None else None
            g = getattr(m, "generic", None)
            arr[i].generic = g.ctypes.data if g is not None else None
            arr[i].generic_components = g.shape[1] if g is not None else 0
        opt = (config or Config())._native()
        h = C.c_void_p()
        t0 = time.perf_counter()
        st = L.dsa_encode_batch(ctx._h, n, arr, C.byref(opt), C.byref(h))
        t1 = time.perf_counter()
        if st != 0:
            _raise(st, ctx.error())
        r = EncodedStreams(ctx, h, n)
        if os.environ.get("DSA_ENC_TIMING"):            # diagnostics, like the library's own phase clocks
            print("[EncodeBatch] native call %.1f ms, result handles %.1f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3), flush=True)
        return r

    def Encode(self, mesh, config=None):
        return self.EncodeBatch([mesh], config)[0]
