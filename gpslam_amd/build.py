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


def _src_deps(src):
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, d) for d in DEPS[src]] + [os.path.abspath(__file__)]


def _digest(src, extra):
    """Content hash of everything one object is made from: the source, the headers it includes, this file (the flags live
    here) and the extra flags.  Round 5: modification times are not evidence -- an edit made WHILE hipcc runs leaves a
    library newer than every source and built from none of them (it happened), and a `touch` makes a stale one look fresh."""
    h = hashlib.sha256(" ".join(FLAGS + list(extra)).encode())
    for d in _src_deps(src):
        h.update(b"\0" + os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _recorded(path):
    try:
        with open(path + ".srchash") as f:
            return f.read().strip()
    except OSError:
        return None


def _record(path, digest):
    tmp = path + ".srchash.tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        f.write(digest + "\n")
    os.replace(tmp, path + ".srchash")


def _lib_digest(extra):
    return hashlib.sha256("".join(_digest(s, extra) for s in SOURCES).encode()).hexdigest()


def up_to_date(extra=()):
    """True when the library on disk was linked from objects compiled from exactly the current sources and flags (the
    digests are written next to the outputs, taken from the file contents read BEFORE the compiler starts)."""
    extra = list(extra)
    _objdir, lib = _paths(extra)
    return os.path.exists(lib) and _recorded(lib) == _lib_digest(extra)


def build(force=False, verbose=False, extra=None):
    """Returns the path of the library.  Several processes may call this at once (pytest children, one rank per GPU): the
    build runs under a file lock, every output is written to a temporary name and renamed into place."""
    if extra is None:
        extra = os.environ.get("GPSLAM_HIPCC_FLAGS", "").split()   # A/B builds of kernel variants under their own file names
    objdir, lib = _paths(extra)
    if not force and up_to_date(extra):
        return lib
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "a") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and up_to_date(extra):        # another process built it while this one waited
            return lib
        digests = {src: _digest(src, extra) for src in SOURCES}      # of the contents as they are NOW, before hipcc reads them
        procs = []
        for src in SOURCES:
            obj = _obj(src, objdir)
            if not force and os.path.exists(obj) and _recorded(obj) == digests[src]:
                continue
            tmp = obj + ".tmp%d" % os.getpid()
            cmd = [_hipcc()] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, tmp, obj, src, subprocess.Popen(cmd, cwd=CSRC)))
        failed = None
        for cmd, tmp, final, src, p in procs:
            if p.wait() != 0:
                failed = failed or subprocess.CalledProcessError(p.returncode, cmd)
                if os.path.exists(tmp):
                    os.remove(tmp)
            elif _digest(src, extra) != digests[src]:
                # the source changed under the compiler: the object is of neither version for sure -- drop it, build again
                os.remove(tmp)
                failed = failed or RuntimeError("%s (or a header it includes) was edited while it was being compiled; run the build again" % src)
            else:
                os.replace(tmp, final)
                _record(final, digests[src])
        if failed:
            raise failed
        tmp = lib + ".tmp%d" % os.getpid()
        link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s, objdir) for s in SOURCES] + ["-o", tmp]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link, cwd=CSRC)
        os.replace(tmp, lib)
        _record(lib, hashlib.sha256("".join(digests[s] for s in SOURCES).encode()).hexdigest())
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
