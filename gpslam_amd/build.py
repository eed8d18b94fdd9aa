"""Builds gpslam_amd/lib/libgpslam_hip.so (hand-written HIP for gfx950) with hipcc.

hipcc cross-compiles for gfx950 without a GPU present, so this runs in the CPU-only container as well;
the built .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgpslam_hip.so")
SOURCES = ["api.hip"]
HEADERS = ["kernels.hpp", "factors.hpp", "lie.hpp", "devbuf.hpp", "fatsep.hpp", "api_impl.inc", os.path.join("..", "..", "include", "gpslam_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    extra = os.environ.get("GPSLAM_HIPCC_FLAGS", "").split()      # e.g. -DGPS_ABLATE_ASM for the timing ablations of DESIGN.md
    cmd = [_hipcc()] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
