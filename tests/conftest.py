import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The suite does not depend on a previous `__graft_entry__.build()`: missing native pieces (the HIP library, the
    # synthetic-input writer, the oracle) are built here, in-tree, exactly as build() does.  hipcc cross-compiles for
    # gfx950 without a GPU.
    needed = [os.path.join(ROOT, "draco-sharp_amd", "csrc", "libdraco_mi355x.so"),
              os.path.join(ROOT, "draco-sharp_amd", "synth", "libdsa_synth.so"),
              os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(f) for f in needed):
        import __graft_entry__
        __graft_entry__.build()
    # torch's bundled HIP runtime has to come up before the library's (the other order leaves torch without a GPU):
    # the device-view test wraps arena pointers as torch tensors
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def house04_bytes():
    with open(os.path.join(ROOT, "tests", "golden", "house_04.obj.drc"), "rb") as f:
        return f.read()
