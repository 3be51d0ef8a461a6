"""The C-ABI library loads without a GPU and exports every symbol include/draco_mi355x.h declares
(no compute calls here)."""
import os
import re

import pytest

import draco_sharp_amd as dsa
from draco_sharp_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "draco_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsa_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    L = native.lib()
    names = declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(names) == sorted(native.EXPORTS)
    assert L.dsa_abi_version() == 4          # 4: dsa_mesh_input carries a generic uint8 attribute (3: dsa_batch_kernel_times, dsa_context_trim, two stream sets)


def test_no_gpu_means_loud_failure_not_fallback():
    L = native.lib()
    if L.dsa_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(dsa.DeviceException):
        dsa.Context(0)
    with pytest.raises(dsa.DeviceException):
        dsa.DracoDecoder().Decode(b"DRACO")


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(native.MeshInfo) == 40
    assert C.sizeof(native.AttributeInfo) == 64
    assert C.sizeof(native.MeshOutput) == 8 + 8 + 2 * 16 * 8
