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
# one object each, compiled side by side, linked into the one library: the C ABI, the fp64-row and the fp32-row half of the
# optimiser (both are api_impl.inc; round 3: they were one 3.5-minute translation unit), the upper solver levels
SOURCES = ["api.hip", "api_impl64.hip", "api_impl32.hip", "upper.hip"]
HEADERS = ["kernels.hpp", "factors.hpp", "lie.hpp", "devbuf.hpp", "fatsep.hpp", "dpp.hpp", "cr_step.hpp", "upper.hpp", "api_common.hpp",
           "api_decl.inc", os.path.join("..", "..", "include", "gpslam_hip.h")]
# what each translation unit includes (an edit to upper.hip does not recompile the others)
DEPS = {"api.hip": HEADERS, "api_impl64.hip": HEADERS + ["api_impl.inc"], "api_impl32.hip": HEADERS + ["api_impl.inc"],
        "upper.hip": ["dpp.hpp", "cr_step.hpp", "upper.hpp"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]
OBJDIR = os.path.join(LIBDIR, "obj")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def _obj(src):
    return os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _src_deps(src):
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, d) for d in DEPS[src]] + [os.path.abspath(__file__)]


def up_to_date():
    if any(_stale(_obj(s), _src_deps(s)) for s in SOURCES):
        return False
    return not _stale(LIB, [_obj(s) for s in SOURCES])


def build(force=False, verbose=False):
    extra = os.environ.get("GPSLAM_HIPCC_FLAGS", "").split()      # e.g. -DGPS_ABLATE_ASM for the timing ablations of DESIGN.md
    if extra:
        force = True
    if not force and up_to_date():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    procs = []
    for src in SOURCES:
        if not force and not _stale(_obj(src), _src_deps(src)):
            continue
        cmd = [_hipcc()] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
