"""draco-sharp_amd/csrc/dsa_symbol_plan.h -- the symbol-scheme choice and rANS table normalisation that the host coder and
the device kernel k_enc_plan share -- against a restatement with the standard library (std::log2, std::stable_sort,
std::floor) on random histograms, under ASan / UBSan (tests/hostcheck/plan_host.cpp)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostcheck", "plan_host.cpp")
EXE = os.path.join(HERE, "hostcheck", "plan_host")
DEP = os.path.join(HERE, "..", "draco-sharp_amd", "csrc", "dsa_symbol_plan.h")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in (SRC, DEP)):
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-ffp-contract=off", "-o", EXE, SRC], check=True)
    return EXE


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_shared_symbol_plan_equals_the_library_restatement(exe, seed):
    r = subprocess.run([exe, str(seed), "3000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
