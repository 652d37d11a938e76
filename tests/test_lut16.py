"""The two-lookup form of the kernels' 16-entry byte LUTs (csrc/common.hpp: lut16_xor_form) against a host model of
v_perm_b32's selector rules (tests/cpp/test_lut16.cpp).  CPU only; the hardware's behaviour is pinned by the GPU parity
tests (all 256 byte values through every context string)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lut16_xor_form(tmp_path):
    exe = str(tmp_path / "test_lut16")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I", os.path.join(ROOT, "epialleler_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_lut16.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "lut16 ok" in r.stdout
