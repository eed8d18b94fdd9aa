"""The C++ host classes (gpslam_amd/host/gpslam_host.hpp): compile the reference-style test program against the
C-ABI library with plain g++ (CPU), run it on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_api_tests")


def build_exe():
    import gpslam_amd
    gpslam_amd.load_library()
    libdir = os.path.join(ROOT, "gpslam_amd", "lib")
    src = os.path.join(ROOT, "tests", "cpp", "host_api_tests.cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", ROOT, src, "-o", EXE, "-L", libdir, "-lgpslam_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_host_header_compiles_and_links_against_the_c_abi():
    exe = build_exe()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_reference_style_cpp_tests_pass_on_gpu():
    exe = build_exe()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout
