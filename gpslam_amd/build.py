"""Builds gpslam_amd/lib/libgpslam_hip.so (hand-written HIP for gfx950) with hipcc.

hipcc cross-compiles for gfx950 without a GPU present, so this runs in the CPU-only container as well;
the built .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import fcntl
import hashlib
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
HEADERS = ["kernels.hpp", "factors.hpp", "lie.hpp", "devbuf.hpp", "fatsep.hpp", "dpp.hpp", "cr_step.hpp", "cr_quad.hpp", "upper.hpp", "api_common.hpp",
           "api_decl.inc", os.path.join("..", "..", "include", "gpslam_hip.h")]
# what each translation unit includes (an edit to upper.hip does not recompile the others)
DEPS = {"api.hip": HEADERS, "api_impl64.hip": HEADERS + ["api_impl.inc"], "api_impl32.hip": HEADERS + ["api_impl.inc"],
        "upper.hip": ["dpp.hpp", "cr_step.hpp", "cr_quad.hpp", "upper.hpp"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"]


def _variant(extra):
    """Ablation / A-B builds (GPSLAM_HIPCC_FLAGS=-DGPS_ABLATE_*: results wrong on purpose) never share file names with the
    product build: their objects and library are keyed by a hash of the extra flags."""
    if not extra:
        return ""
    return "_" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10]


def _paths(extra):
    tag = _variant(extra)
    return os.path.join(LIBDIR, "obj" + tag), os.path.join(LIBDIR, "libgpslam_hip%s.so" % tag)


OBJDIR = _paths([])[0]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def _obj(src, objdir=None):
    return os.path.join(objdir or OBJDIR, os.path.splitext(src)[0] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _src_deps(src):
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, d) for d in DEPS[src]] + [os.path.abspath(__file__)]


def up_to_date(extra=()):
    objdir, lib = _paths(list(extra))
    if not _stale(lib, [d for s in SOURCES for d in _src_deps(s)]):
        return True            # the objects are intermediates: a library newer than every source needs none of them
    if any(_stale(_obj(s, objdir), _src_deps(s)) for s in SOURCES):
        return False
    return not _stale(lib, [_obj(s, objdir) for s in SOURCES])


def build(force=False, verbose=False, extra=None):
    """Returns the path of the library.  Several processes may call this at once (pytest children, one rank per GPU): the
    build runs under a file lock, every output is written to a temporary name and renamed into place."""
    if extra is None:
        extra = os.environ.get("GPSLAM_HIPCC_FLAGS", "").split()   # e.g. -DGPS_ABLATE_ASM for the timing ablations of DESIGN.md
    objdir, lib = _paths(extra)
    if not force and up_to_date(extra):
        return lib
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and up_to_date(extra):        # another process built it while this one waited
            return lib
        procs = []
        for src in SOURCES:
            if not force and not _stale(_obj(src, objdir), _src_deps(src)):
                continue
            tmp = _obj(src, objdir) + ".tmp%d" % os.getpid()
            cmd = [_hipcc()] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, tmp, _obj(src, objdir), subprocess.Popen(cmd, cwd=CSRC)))
        failed = None
        for cmd, tmp, final, p in procs:
            if p.wait() != 0:
                failed = failed or subprocess.CalledProcessError(p.returncode, cmd)
                if os.path.exists(tmp):
                    os.remove(tmp)
            else:
                os.replace(tmp, final)
        if failed:
            raise failed
        tmp = lib + ".tmp%d" % os.getpid()
        link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s, objdir) for s in SOURCES] + ["-o", tmp]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link, cwd=CSRC)
        os.replace(tmp, lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
