"""-m gpu: a plain C++ program (no Python, no torch) drives libgqhip.so through include/gqhip.h."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_consumer_of_the_c_abi(tmp_path):
    from oracle import gq_oracle
    from pit_hip import _lib

    gq_oracle.build()
    csrc = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "cabi_smoke")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi_smoke.cpp"), "-L", csrc, "-lgqhip",
                           "-L", os.path.join(ROOT, "oracle"), "-lgq_oracle",
                           f"-Wl,-rpath,{csrc}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "index mismatches 0" in out.stdout
