"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(_ROOT, "oracle", "liboracle.so")


class AttrInfo(C.Structure):
    _fields_ = [
        ("att_type", C.c_int32), ("data_type", C.c_int32), ("num_components", C.c_int32),
        ("normalized", C.c_int32), ("unique_id", C.c_uint32), ("seq_type", C.c_int32),
        ("decoder_id", C.c_int32), ("pred_method", C.c_int32), ("pred_transform", C.c_int32),
        ("num_entries", C.c_uint32), ("nc_portable", C.c_int32), ("q_bits", C.c_int32),
        ("q_range", C.c_float), ("q_min", C.c_float * 4), ("oct_bits", C.c_int32),
        ("value_bytes", C.c_uint64),
    ]


def build(force=False):
    src = os.path.join(_ROOT, "oracle", "drc_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_decode.restype = C.c_void_p
        L.orc_decode.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.orc_free.argtypes = [C.c_void_p]
        for name in ("orc_num_faces", "orc_num_points", "orc_num_vertices", "orc_num_attributes", "orc_num_decoders"):
            getattr(L, name).restype = C.c_uint32
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_header.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_faces.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_corner_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_eb_symbols.restype = C.c_uint32
        L.orc_eb_symbols.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_decoder_num_entries.restype = C.c_uint32
        L.orc_decoder_num_entries.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_decoder_info.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_decoder_sequence.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_attr_get_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(AttrInfo)]
        L.orc_attr_values.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_attr_portable.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_attr_symbols.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_attr_point_map.restype = C.c_uint32
        L.orc_attr_point_map.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_varint.restype = C.c_uint64
        L.orc_varint.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_bits.restype = C.c_uint32
        L.orc_bits.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_int_sqrt.restype = C.c_uint64
        L.orc_int_sqrt.argtypes = [C.c_uint64]
        L.orc_zigzag.restype = C.c_int32
        L.orc_zigzag.argtypes = [C.c_uint32]
        L.orc_decode_symbols.restype = C.c_int64
        L.orc_decode_symbols.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_decode_rabs.restype = C.c_int64
        L.orc_decode_rabs.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.orc_oct_to_unit.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_dequantize.restype = C.c_float
        L.orc_dequantize.argtypes = [C.c_int32, C.c_float, C.c_int, C.c_float]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


_DT_NUMPY = {1: np.int8, 2: np.uint8, 3: np.int16, 4: np.uint16, 5: np.int32, 6: np.uint32,
             7: np.int64, 8: np.uint64, 9: np.float32, 10: np.float64, 11: np.uint8}


class OracleAttribute:
    pass


class OracleMesh:
    """Everything the oracle decoded, as numpy arrays."""

    def __init__(self, data: bytes):
        L = lib()
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        code = C.c_int(0)
        err = C.create_string_buffer(256)
        h = L.orc_decode(buf, len(data), C.byref(code), err, 256)
        if not h:
            raise OracleError(code.value, err.value.decode())
        try:
            hdr = np.zeros(8, np.int32)
            L.orc_header(h, hdr.ctypes.data)
            (self.major, self.minor, self.encoder_type, self.encoder_method, self.flags,
             self.traversal_type, self.num_attribute_data, self.end_pos) = [int(x) for x in hdr]
            self.num_faces = L.orc_num_faces(h)
            self.num_points = L.orc_num_points(h)
            self.num_vertices = L.orc_num_vertices(h)
            nf = self.num_faces
            self.faces = np.zeros((nf, 3), np.int32)
            self.opposite = np.zeros(nf * 3, np.uint32)
            self.corner_to_vertex = np.zeros(nf * 3, np.uint32)
            self.vertex_corners = np.zeros(self.num_vertices, np.uint32)
            if nf:
                L.orc_faces(h, self.faces.ctypes.data)
                L.orc_corner_table(h, self.opposite.ctypes.data, self.corner_to_vertex.ctypes.data,
                                   self.vertex_corners.ctypes.data)
            ns = L.orc_eb_symbols(h, None)
            self.eb_symbols = np.zeros(ns, np.uint8)
            if ns:
                L.orc_eb_symbols(h, self.eb_symbols.ctypes.data)
            self.decoders = []
            for d in range(L.orc_num_decoders(h)):
                info = np.zeros(4, np.int32)
                L.orc_decoder_info(h, d, info.ctypes.data)
                n = L.orc_decoder_num_entries(h, d)
                pids = np.zeros(n, np.uint32)
                d2c = np.zeros(n, np.uint32)
                L.orc_decoder_sequence(h, d, pids.ctypes.data, d2c.ctypes.data if nf else None)
                self.decoders.append(dict(att_data_id=int(info[0]), element_type=int(info[1]),
                                          traversal_method=int(info[2]), num_attributes=int(info[3]),
                                          point_ids=pids, data_to_corner=d2c))
            self.attributes = []
            for a in range(L.orc_num_attributes(h)):
                ai = AttrInfo()
                L.orc_attr_get_info(h, a, C.byref(ai))
                o = OracleAttribute()
                for f, _ in AttrInfo._fields_:
                    v = getattr(ai, f)
                    setattr(o, f, list(v) if f == "q_min" else v)
                dt = _DT_NUMPY[ai.data_type]
                raw = np.zeros(ai.value_bytes, np.uint8)
                if ai.value_bytes:
                    L.orc_attr_values(h, a, raw.ctypes.data)
                o.values = raw.view(dt).reshape(ai.num_entries, ai.num_components) if ai.num_entries else raw.view(dt)
                if ai.seq_type != 0:
                    o.portable = np.zeros((ai.num_entries, ai.nc_portable), np.int32)
                    o.symbols = np.zeros((ai.num_entries, ai.nc_portable), np.uint32)
                    if ai.num_entries:
                        L.orc_attr_portable(h, a, o.portable.ctypes.data)
                        L.orc_attr_symbols(h, a, o.symbols.ctypes.data)
                else:
                    o.portable = None
                    o.symbols = None
                n = L.orc_attr_point_map(h, a, None)
                o.point_map = np.zeros(n, np.uint32)
                if n:
                    L.orc_attr_point_map(h, a, o.point_map.ctypes.data)
                self.attributes.append(o)
        finally:
            L.orc_free(h)


def _decoders_of_attributes(self):
    """Per attribute the index of the attributes decoder it belongs to."""
    out = []
    for d, dec in enumerate(self.decoders):
        out += [d if dec["element_type"] else 0] * dec["num_attributes"]
    return out


OracleMesh.decoders_of_attributes = _decoders_of_attributes


def decode(data: bytes) -> OracleMesh:
    return OracleMesh(data)
