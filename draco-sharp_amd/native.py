"""ctypes binding of libdraco_mi355x.so (include/draco_mi355x.h).  There is no CPU
fallback: if the HIP library is missing or no GPU is present the calls raise."""
import ctypes as C
import os
import subprocess

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("DSA_LIB") or os.path.join(_DIR, "libdraco_mi355x.so")   # DSA_LIB: kernel-ablation builds

DSA_OK, DSA_ERR_INVALID_DATA, DSA_ERR_NOT_IMPLEMENTED, DSA_ERR_INVALID_ARGUMENT, DSA_ERR_DEVICE, DSA_ERR_OUT_OF_MEMORY = range(6)
DSA_NUM_STAGES = 8
DSA_OUTPUT_FACES_U16 = 1
NO_MAP = 0xFFFFFFFFFFFFFFFF   # dsa_mesh_output.point_map: the identity map, not stored (compact download)

# every symbol include/draco_mi355x.h declares
EXPORTS = [
    "dsa_abi_version", "dsa_device_count", "dsa_context_create", "dsa_context_destroy", "dsa_last_error",
    "dsa_batch_create", "dsa_batch_create_packed", "dsa_batch_decode", "dsa_batch_wait", "dsa_batch_free",
    "dsa_batch_size", "dsa_batch_algorithmic_bytes", "dsa_batch_arena_bytes", "dsa_batch_mesh_info",
    "dsa_batch_attribute_info", "dsa_batch_copy_faces", "dsa_batch_copy_attribute_values", "dsa_batch_copy_point_map",
    "dsa_batch_copy_portable_values", "dsa_batch_device_faces", "dsa_batch_device_attribute_values",
    "dsa_batch_device_point_map", "dsa_batch_output_bytes", "dsa_batch_download", "dsa_batch_compact_bytes", "dsa_batch_download_compact", "dsa_batch_host_output", "dsa_batch_output_layout",
    "dsa_host_alloc", "dsa_host_free", "dsa_host_register", "dsa_host_unregister", "dsa_batch_copy_metadata", "dsa_batch_copy_debug", "dsa_context_set_profiling", "dsa_batch_stage_times",
    "dsa_batch_kernel_times", "dsa_context_trim", "dsa_context_schedule_note",
    "dsa_encode_default_options", "dsa_encode_batch", "dsa_encoded_size", "dsa_encoded_stream", "dsa_encoded_free",
    "dsa_pool_create", "dsa_pool_destroy", "dsa_pool_size", "dsa_pool_last_error", "dsa_pool_decode", "dsa_pool_job_locate",
    "dsa_pool_job_chunks", "dsa_pool_job_free", "dsa_pool_plan",
]


class EncodeOptions(C.Structure):
    _fields_ = [("position_bits", C.c_int32), ("texcoord_bits", C.c_int32), ("normal_bits", C.c_int32),
                ("single_connectivity", C.c_int32), ("symbol_scheme", C.c_int32), ("compression_level", C.c_int32),
                ("position_prediction", C.c_int32), ("texcoord_prediction", C.c_int32)]


class MeshInput(C.Structure):
    _fields_ = [("num_vertices", C.c_uint32), ("num_faces", C.c_uint32), ("positions", C.c_void_p), ("faces", C.c_void_p),
                ("normals", C.c_void_p), ("texcoords", C.c_void_p), ("generic", C.c_void_p), ("generic_components", C.c_uint32),
                ("reserved", C.c_uint32)]


class MeshInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("detail", C.c_int32), ("major_version", C.c_uint8), ("minor_version", C.c_uint8),
                ("encoder_type", C.c_uint8), ("encoder_method", C.c_uint8), ("flags", C.c_uint16), ("decode_path", C.c_uint16),
                ("num_faces", C.c_uint32), ("num_points", C.c_uint32), ("num_attributes", C.c_uint32),
                ("drc_bytes", C.c_uint64)]


class MeshOutput(C.Structure):
    _fields_ = [("block", C.c_uint32), ("flags", C.c_uint32), ("faces", C.c_uint64), ("values", C.c_uint64 * 16), ("point_map", C.c_uint64 * 16)]


class AttributeInfo(C.Structure):
    _fields_ = [("attribute_type", C.c_int32), ("data_type", C.c_int32), ("num_components", C.c_int32),
                ("normalized", C.c_int32), ("unique_id", C.c_uint32), ("num_entries", C.c_uint32),
                ("byte_stride", C.c_uint32), ("decoder_type", C.c_int32), ("prediction_method", C.c_int32),
                ("prediction_transform", C.c_int32), ("quantization_bits", C.c_int32), ("range", C.c_float),
                ("min_values", C.c_float * 4)]


def build(force=False):
    """Compiles the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    # make knows the prerequisites (compiler-written depfile)
    subprocess.check_call(["make", "-C", _DIR, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libdraco_mi355x.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, u32 = C.c_void_p, C.c_uint32
        L.dsa_abi_version.restype = C.c_int
        L.dsa_device_count.restype = C.c_int
        L.dsa_context_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
        L.dsa_context_destroy.argtypes = [vp]
        L.dsa_last_error.restype = C.c_char_p
        L.dsa_last_error.argtypes = [vp]
        L.dsa_batch_create.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp)]
        L.dsa_batch_create_packed.argtypes = [vp, u32, vp, vp, C.POINTER(vp)]
        L.dsa_batch_decode.argtypes = [vp]
        L.dsa_batch_wait.argtypes = [vp]
        L.dsa_batch_free.argtypes = [vp]
        L.dsa_batch_size.restype = u32
        L.dsa_batch_size.argtypes = [vp]
        L.dsa_batch_algorithmic_bytes.restype = C.c_uint64
        L.dsa_batch_algorithmic_bytes.argtypes = [vp]
        L.dsa_batch_arena_bytes.restype = C.c_uint64
        L.dsa_batch_arena_bytes.argtypes = [vp]
        L.dsa_batch_mesh_info.argtypes = [vp, u32, C.POINTER(MeshInfo)]
        L.dsa_batch_attribute_info.argtypes = [vp, u32, u32, C.POINTER(AttributeInfo)]
        L.dsa_batch_copy_faces.argtypes = [vp, u32, vp]
        L.dsa_batch_copy_attribute_values.argtypes = [vp, u32, u32, vp]
        L.dsa_batch_copy_point_map.argtypes = [vp, u32, u32, vp]
        L.dsa_batch_copy_portable_values.argtypes = [vp, u32, u32, vp]
        for f in ("dsa_batch_device_faces",):
            getattr(L, f).restype = vp
            getattr(L, f).argtypes = [vp, u32]
        for f in ("dsa_batch_device_attribute_values", "dsa_batch_device_point_map"):
            getattr(L, f).restype = vp
            getattr(L, f).argtypes = [vp, u32, u32]
        L.dsa_batch_output_bytes.restype = C.c_uint64
        L.dsa_batch_output_bytes.argtypes = [vp]
        L.dsa_batch_download.argtypes = [vp, vp, C.c_size_t]
        L.dsa_batch_download_compact.argtypes = [vp, vp, C.c_size_t]
        L.dsa_batch_compact_bytes.restype = C.c_uint64
        L.dsa_batch_compact_bytes.argtypes = [vp]
        L.dsa_batch_host_output.restype = vp
        L.dsa_batch_host_output.argtypes = [vp, u32]
        L.dsa_batch_output_layout.argtypes = [vp, u32, C.POINTER(MeshOutput)]
        L.dsa_host_alloc.restype = vp
        L.dsa_host_alloc.argtypes = [C.c_size_t]
        L.dsa_host_free.restype = None
        L.dsa_host_free.argtypes = [vp]
        L.dsa_host_register.argtypes = [vp, C.c_size_t]
        L.dsa_host_unregister.argtypes = [vp]
        L.dsa_batch_copy_metadata.argtypes = [vp, u32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.dsa_batch_copy_debug.argtypes = [vp, u32, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.dsa_context_set_profiling.argtypes = [vp, C.c_int]
        L.dsa_batch_stage_times.argtypes = [vp, C.POINTER(C.c_float * DSA_NUM_STAGES), C.POINTER(C.c_char_p * DSA_NUM_STAGES)]
        L.dsa_batch_kernel_times.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_uint32)]
        L.dsa_context_trim.argtypes = [vp]
        L.dsa_context_schedule_note.restype = C.c_char_p
        L.dsa_context_schedule_note.argtypes = [vp]
        L.dsa_encode_default_options.argtypes = [C.POINTER(EncodeOptions)]
        L.dsa_encode_default_options.restype = None
        L.dsa_encode_batch.argtypes = [vp, u32, C.POINTER(MeshInput), C.POINTER(EncodeOptions), C.POINTER(vp)]
        L.dsa_encoded_size.restype = u32
        L.dsa_encoded_size.argtypes = [vp]
        L.dsa_encoded_stream.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.dsa_encoded_free.argtypes = [vp]
        L.dsa_encoded_free.restype = None
        L.dsa_pool_create.argtypes = [C.POINTER(C.c_int), u32, u32, C.POINTER(vp)]
        L.dsa_pool_destroy.argtypes = [vp]
        L.dsa_pool_destroy.restype = None
        L.dsa_pool_size.argtypes = [vp]
        L.dsa_pool_size.restype = u32
        L.dsa_pool_last_error.argtypes = [vp]
        L.dsa_pool_last_error.restype = C.c_char_p
        L.dsa_pool_decode.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp)]
        L.dsa_pool_job_locate.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u32), C.POINTER(u32)]
        L.dsa_pool_job_chunks.argtypes = [vp]
        L.dsa_pool_job_chunks.restype = u32
        L.dsa_pool_job_free.argtypes = [vp]
        L.dsa_pool_job_free.restype = None
        L.dsa_pool_plan.argtypes = [u32, C.POINTER(C.c_size_t), u32, C.POINTER(u32), C.POINTER(u32)]
        L.dsa_pool_plan.restype = u32
        _lib = L
    return _lib
