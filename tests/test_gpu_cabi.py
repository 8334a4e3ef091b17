"""-m gpu: a plain C++ program (no Python, no torch) drives libgqhip.so through include/gqhip.h."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name):
    from oracle import gq_oracle
    from pit_hip import _lib

    gq_oracle.build()
    csrc = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", name + ".cpp"), "-L", csrc, "-lgqhip",
                           "-L", os.path.join(ROOT, "oracle"), "-lgq_oracle",
                           f"-Wl,-rpath,{csrc}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-o", exe])
    return exe


def test_cpp_consumer_of_the_c_abi(tmp_path):
    out = subprocess.run([_build(tmp_path, "cabi_smoke")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "index mismatches 0" in out.stdout


def test_cpp_consumer_of_the_module_level_c_abi(tmp_path):
    """ABI 7 / 8 through C++ only: gq_quantize_z_f32 in both layouts / groupings, the dim-4 search with a caller-owned codebook cache
    (garbage at first, then the codebook edited in place), vq_argmin_f32, vq_quantize_z_f32, lfq_pack_f32 and
    gq_quantize_z_gauss_f32 (two calls: the lambda state advances on the device) -- all against libgq_oracle.so."""
    out = subprocess.run([_build(tmp_path, "cabi_modules")], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "total mismatches 0" in out.stdout
