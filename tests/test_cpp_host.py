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


def test_equals_print_clone_of_the_host_factor_classes():
    """equals(expected, tol) / print(s, keyFormatter) / clone() as the reference's factors define them
    (gpslam/gp/GaussianProcessPriorPose3.h:104-115): no device call, runs on the CPU."""
    exe = build_exe()
    out = subprocess.run([exe, "--host-only"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all host-only tests passed" in out.stdout
    assert "factor 1: 4-way Gaussian Process Factor Pose3" in out.stdout and "<x1>" in out.stdout
    lines = out.stdout.splitlines()
    for tag, keys in (("zero landmark key: ", "[k7] [k0]"), ("zero second key: ", "[k3] [k0]")):      # a key VALUE of 0 is printed
        at = [i for i, l in enumerate(lines) if l.startswith(tag)]
        assert at and keys in lines[at[0] + 1], (tag, lines[at[0]:at[0] + 3] if at else None)


@pytest.mark.gpu
def test_reference_style_cpp_tests_pass_on_gpu():
    exe = build_exe()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


def test_fp32_math_on_the_host():
    """tests/cpp/fp32_math_tests.cpp: blended coefficients and the exact derivative that replaces the reference's h = 1e-6
    finite difference in the fp32 mode; hipcc's host pass, runs on the CPU."""
    exe = os.path.join(ROOT, "tests", "cpp", "fp32_math_tests")
    src = os.path.join(ROOT, "tests", "cpp", "fp32_math_tests.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "gpslam_amd", "csrc", "factors.hpp")),
                                                              os.path.getmtime(os.path.join(ROOT, "gpslam_amd", "csrc", "lie.hpp"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all fp32 math tests passed" in out.stdout


def test_gtsam_adapter_header_is_a_valid_translation_unit_without_gtsam():
    """gpslam_amd/host/gtsam_adapter.hpp is guarded by __has_include(<gtsam/...>): in this image (no GTSAM) it must
    preprocess to nothing and compile cleanly next to the ABI header."""
    src = os.path.join(ROOT, "tests", "cpp", "_adapter_tu.cpp")
    with open(src, "w") as f:
        f.write('#include "../../gpslam_amd/host/gtsam_adapter.hpp"\n#include "../../include/gpslam_hip.h"\n'
                '#ifdef GPSLAM_HIP_HAVE_GTSAM\n#error "GTSAM unexpectedly present: run the adapter tests instead"\n#endif\nint main() { return 0; }\n')
    try:
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", src])
    finally:
        os.remove(src)


def _build_sharded_exe():
    import gpslam_amd
    gpslam_amd.load_library()
    libdir = os.path.join(ROOT, "gpslam_amd", "lib")
    exe = os.path.join(ROOT, "tests", "cpp", "sharded_rccl_test")
    src = os.path.join(ROOT, "tests", "cpp", "sharded_rccl_test.cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", ROOT, "-I", "/opt/rocm/include", src, "-o", exe,
           "-L", libdir, "-lgpslam_hip", "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_cpp_sharded_host_compiles_against_rccl():
    assert os.path.exists(_build_sharded_exe())


@pytest.mark.gpu
def test_cpp_sharded_host_runs_rccl_on_every_visible_gpu():
    """gpslam_amd/host/sharded_host.hpp: phase1 -> ncclAllGather -> phase2 from C++, one rank per visible GPU (a forced-
    sharded single rank on a 1-GPU box), against the unsharded solve."""
    out = subprocess.run([_build_sharded_exe()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


def _build_split_exe():
    import gpslam_amd
    gpslam_amd.load_library()
    libdir = os.path.join(ROOT, "gpslam_amd", "lib")
    exe = os.path.join(ROOT, "tests", "cpp", "split_rccl_test")
    src = os.path.join(ROOT, "tests", "cpp", "split_rccl_test.cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", ROOT, "-I", "/opt/rocm/include", src, "-o", exe,
           "-L", libdir, "-lgpslam_hip", "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_cpp_split_host_compiles_against_rccl():
    assert os.path.exists(_build_split_exe())


@pytest.mark.gpu
def test_cpp_split_host_runs_config4_pieces():
    """gpslam_amd/host/sharded_host.hpp: SplitDriver -- BASELINE config 4's graph cut into pieces from C++ (one per visible GPU
    over RCCL, and three on device 0 with device copies) against the unsplit solve."""
    out = subprocess.run([_build_split_exe()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all tests passed" in out.stdout


def test_gtsam_only_files_type_check_against_declaration_headers():
    """gtsam_adapter.hpp and bench/gtsam_reference.cpp have no compiler anywhere this project is built (no GTSAM, no Boost, no
    reference install).  tests/cpp/gtsam_decl/ declares -- bodies none, behaviour none -- the GTSAM / gpslam names they use,
    so that both translation units are at least type-checked: both optimizer instantiations of the
    adapter, and the reference benchmark's live branch."""
    decl = os.path.join(ROOT, "tests", "cpp", "gtsam_decl")
    src = os.path.join(ROOT, "tests", "cpp", "_adapter_live_tu.cpp")
    with open(src, "w") as f:
        f.write('#include "../../gpslam_amd/host/gtsam_adapter.hpp"\n#ifndef GPSLAM_HIP_HAVE_GTSAM\n#error "declaration headers not found"\n#endif\n'
                'template class gpslam_hip::HipChainOptimizerT<gtsam::Pose3>;\ntemplate class gpslam_hip::HipChainOptimizerT<gtsam::Pose2>;\n'
                'int main() { return 0; }\n')
    try:
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", decl, src])
    finally:
        os.remove(src)
    ref = os.path.join(ROOT, "bench", "gtsam_reference.cpp")
    pre = subprocess.run(["g++", "-std=c++17", "-E", "-dM", "-I", decl, ref], capture_output=True, text=True, check=True).stdout
    assert "HAVE_REFERENCE" in pre                       # the live branch is the one being checked
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", decl, ref])
