"""Synthetic inputs: procedural meshes and a CPU writer of Draco v2.2 streams
(ctypes binding of libdsa_synth.so).  Used by tests and bench.py to make .drc
inputs; not part of the decode path."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "libdsa_synth.so")

GRID, TORUS, SPHERE, HOLES, TWO_PARTS = 0, 1, 2, 3, 4


class Options(C.Structure):
    _fields_ = [("pos_bits", C.c_int32), ("uv_bits", C.c_int32), ("normal_bits", C.c_int32),
                ("single_connectivity", C.c_int32), ("force_scheme", C.c_int32),
                ("compression_level", C.c_int32), ("pos_prediction", C.c_int32), ("uv_prediction", C.c_int32),
                ("normal_prediction", C.c_int32), ("traversal_method", C.c_int32), ("predictive_connectivity", C.c_int32),
                ("normal_transform", C.c_int32), ("raw_integers", C.c_int32), ("no_prediction", C.c_int32),
                ("generic_components", C.c_int32)]


def build(force=False):
    src = os.path.join(_DIR, "synth_encoder.cpp")
    deps = [src, os.path.join(_DIR, "..", "csrc", "dsa_encode_host.h")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(_LIB_PATH) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.synth_last_error.restype = C.c_char_p
        L.synth_default_options.argtypes = [C.POINTER(Options)]
        L.synth_encode_mesh.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.POINTER(Options), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.synth_encode_mesh_corners.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                                C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Options), C.POINTER(C.c_void_p),
                                                C.POINTER(C.c_size_t)]
        L.synth_encode_point_cloud.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Options), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_size_t)]
        L.synth_encode_mesh_sequential.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int,
                                                   C.POINTER(Options), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.synth_free.argtypes = [C.c_void_p]
        L.synth_make_mesh.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.synth_make_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int,
                                       C.POINTER(Options), C.c_int, C.POINTER(C.c_void_p), C.c_void_p]
        L.synth_encode_symbols.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_size_t)]
        L.synth_encode_rabs.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def options(**kw):
    o = Options()
    lib().synth_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown synth option %r" % k)
        setattr(o, k, v)
    return o


def _err():
    return lib().synth_last_error().decode()


def make_mesh(kind, nx, ny, seed):
    """Returns (pos[V,3] f32, normals[V,3] f32, uv[V,2] f32, faces[F,3] u32)."""
    L = lib()
    nv, nf = C.c_uint32(), C.c_uint32()
    if L.synth_make_mesh(kind, nx, ny, seed, C.byref(nv), C.byref(nf), None, None, None, None):
        raise RuntimeError(_err())
    pos = np.zeros((nv.value, 3), np.float32)
    nrm = np.zeros((nv.value, 3), np.float32)
    uv = np.zeros((nv.value, 2), np.float32)
    faces = np.zeros((nf.value, 3), np.uint32)
    L.synth_make_mesh(kind, nx, ny, seed, C.byref(nv), C.byref(nf), pos.ctypes.data, nrm.ctypes.data,
                      uv.ctypes.data, faces.ctypes.data)
    return pos, nrm, uv, faces


def encode_mesh(pos, faces, normals=None, uvs=None, generic=None, opt=None):
    L = lib()
    pos = np.ascontiguousarray(pos, np.float32)
    faces = np.ascontiguousarray(faces, np.uint32)
    nrm = None if normals is None else np.ascontiguousarray(normals, np.float32)
    uv = None if uvs is None else np.ascontiguousarray(uvs, np.float32)
    gen = None if generic is None else np.ascontiguousarray(generic, np.uint8)
    out, n = C.c_void_p(), C.c_size_t()
    opt = opt or options()
    rc = L.synth_encode_mesh(pos.ctypes.data, len(pos), faces.ctypes.data, len(faces),
                             None if nrm is None else nrm.ctypes.data, None if uv is None else uv.ctypes.data,
                             None if gen is None else gen.ctypes.data, C.byref(opt), C.byref(out), C.byref(n))
    if rc:
        raise RuntimeError(_err())
    data = C.string_at(out, n.value)
    L.synth_free(out)
    return data


def encode_mesh_corners(pos, faces, normals=None, normal_corners=None, uvs=None, uv_corners=None, opt=None):
    """Mesh whose normals / texture coordinates are given per corner: `faces` [F,3] index `pos`, `normal_corners` /
    `uv_corners` [F,3] index the rows of `normals` / `uvs` (None: that attribute has one row per vertex).  Edges across
    which the ids differ become attribute seams in the stream (seam bits, attribute corner table, corner attribute)."""
    L = lib()
    pos = np.ascontiguousarray(pos, np.float32)
    faces = np.ascontiguousarray(faces, np.uint32)
    nrm = None if normals is None else np.ascontiguousarray(normals, np.float32)
    uv = None if uvs is None else np.ascontiguousarray(uvs, np.float32)
    nci = None if normal_corners is None else np.ascontiguousarray(normal_corners, np.uint32)
    uci = None if uv_corners is None else np.ascontiguousarray(uv_corners, np.uint32)
    for ids, vals, name in ((nci, nrm, "normal"), (uci, uv, "uv")):
        if ids is not None and (vals is None or ids.shape != faces.shape):
            raise ValueError("%s_corners needs %ss and one id per corner of `faces`" % (name, name))
        if ids is None and vals is not None and len(vals) != len(pos):
            raise ValueError("per-vertex %ss need one row per vertex" % name)
    out, n = C.c_void_p(), C.c_size_t()
    opt = opt or options()
    rc = L.synth_encode_mesh_corners(pos.ctypes.data, len(pos), faces.ctypes.data, len(faces),
                                     None if nrm is None else nrm.ctypes.data, 0 if nrm is None else len(nrm),
                                     None if nci is None else nci.ctypes.data,
                                     None if uv is None else uv.ctypes.data, 0 if uv is None else len(uv),
                                     None if uci is None else uci.ctypes.data, C.byref(opt), C.byref(out), C.byref(n))
    if rc:
        raise RuntimeError(_err())
    data = C.string_at(out, n.value)
    L.synth_free(out)
    return data


def encode_mesh_sequential(pos, faces, normals=None, uvs=None, compressed=True, opt=None):
    """Sequential mesh stream (MeshSequentialEncoder): faces as point indices, attributes in point order."""
    L = lib()
    pos = np.ascontiguousarray(pos, np.float32)
    faces = np.ascontiguousarray(faces, np.uint32)
    nrm = None if normals is None else np.ascontiguousarray(normals, np.float32)
    uv = None if uvs is None else np.ascontiguousarray(uvs, np.float32)
    out, n = C.c_void_p(), C.c_size_t()
    opt = opt or options()
    rc = L.synth_encode_mesh_sequential(pos.ctypes.data, len(pos), faces.ctypes.data, len(faces),
                                        None if nrm is None else nrm.ctypes.data, None if uv is None else uv.ctypes.data,
                                        1 if compressed else 0, C.byref(opt), C.byref(out), C.byref(n))
    if rc:
        raise RuntimeError(_err())
    data = C.string_at(out, n.value)
    L.synth_free(out)
    return data


def encode_point_cloud(pos, opt=None):
    L = lib()
    pos = np.ascontiguousarray(pos, np.float32)
    out, n = C.c_void_p(), C.c_size_t()
    opt = opt or options()
    if L.synth_encode_point_cloud(pos.ctypes.data, len(pos), C.byref(opt), C.byref(out), C.byref(n)):
        raise RuntimeError(_err())
    data = C.string_at(out, n.value)
    L.synth_free(out)
    return data


def make_batch(kind, nx, ny, seed0, count, normals=True, uvs=True, opt=None, threads=None):
    """Encodes `count` meshes (seeds seed0..) and returns (blob uint8[...], offsets uint64[count+1])."""
    L = lib()
    opt = opt or options()
    threads = threads or min(os.cpu_count() or 1, 32)
    blob = C.c_void_p()
    offsets = np.zeros(count + 1, np.uint64)
    mask = (1 if normals else 0) | (2 if uvs else 0)
    if L.synth_make_batch(kind, nx, ny, seed0, count, mask, C.byref(opt), threads, C.byref(blob), offsets.ctypes.data):
        raise RuntimeError(_err())
    total = int(offsets[-1])
    arr = np.frombuffer(C.string_at(blob, total), np.uint8).copy() if total else np.zeros(0, np.uint8)
    L.synth_free(blob)
    return arr, offsets


def encode_symbols(values, nc=1, force_scheme=-1, compression_level=5):
    """DecodeSymbols()-compatible block (scheme byte first) for uint32 `values`."""
    L = lib()
    v = np.ascontiguousarray(values, np.uint32).ravel()
    out, n = C.c_void_p(), C.c_size_t()
    if L.synth_encode_symbols(v.ctypes.data, v.size, nc, force_scheme, compression_level, C.byref(out), C.byref(n)):
        raise RuntimeError(_err())
    data = C.string_at(out, n.value)
    L.synth_free(out)
    return data


def encode_rabs(bits):
    """rABS bit block {prob_zero, size varint, bytes} for a 0/1 array."""
    L = lib()
    b = np.ascontiguousarray(bits, np.uint8).ravel()
    out, n = C.c_void_p(), C.c_size_t()
    if L.synth_encode_rabs(b.ctypes.data, b.size, C.byref(out), C.byref(n)):
        raise RuntimeError(_err())
    data = C.string_at(out, n.value)
    L.synth_free(out)
    return data
