"""Builds sccav_cbf_amd/lib/libsccbf.so (HIP kernels + C ABI) for gfx950 with hipcc."""
import fcntl
import hashlib
import os
import shutil
import subprocess
import tempfile

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
SRC = os.path.join(CSRC, "sccbf.hip")      # the one translation unit; it includes the *.hip.h next to it
HDR = os.path.join(_ROOT, "include", "sccbf.h")
LIB_DIR = os.path.join(_PKG, "lib")
LIB = os.path.join(LIB_DIR, "libsccbf.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",   # -shared: link step only
    # packed-fp32 forms have produced sporadic wrong results on this hardware pool
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
    # leading scalar kernel arguments arrive in SGPRs at wave launch instead of through s_load
    "-mllvm", "-amdgpu-kernarg-preload-count=16",
]


# parts of the one source file (sccbf.hip says which entry points each holds) and their extra flags:
# the fp64 rollout is compiled without machine-level loop-invariant hoisting, which would keep the
# double-precision constants of the library transcendentals in registers across its tick loop
PARTS = [("F32", []), ("F64", []), ("ROLLOUT_F64", ["-mllvm", "-disable-machine-licm"]), ("BATCHES", [])]


def sources():
    return [HDR] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hip.h")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in sources())


def source_hash():
    """What the library is built from (the header, the kernels, the compiler flags), as 16 hex digits: counter
    runs record it, and a number measured on other sources is not shown next to this library."""
    h = hashlib.sha256()
    for f in sources():
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(repr((HIPCC_FLAGS, PARTS)).encode())
    return h.hexdigest()[:16]


def build_library(force=False, verbose=False, out=None, extra_flags=()):
    """Compile the shared library in-tree; returns its path.  `out` / `extra_flags`: a variant build beside
    the library (for A/B measurements: load it with SCCBF_LIBRARY=<path>), always compiled."""
    if out is None and not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    # Several processes may find the library stale at once (the ranks of a launcher, test workers): one of them
    # builds, the others wait at the lock and find the work done.  Objects go to a directory of this process's
    # own and the finished library is moved into place in one step, so nobody maps a half-written file.
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if out is None and not force and not needs_build():
            return LIB
        return _build_locked(out or LIB, verbose, extra_flags)


def _build_locked(out, verbose, extra_flags):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # the one source file in parts, side by side; then one link
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-I" + os.path.join(_ROOT, "include"), "-c"]
    with tempfile.TemporaryDirectory(prefix="build.%d." % os.getpid(), dir=LIB_DIR) as tmp:
        objs, procs = [], []
        for tag, extra in PARTS:
            obj = os.path.join(tmp, "sccbf_%s.o" % tag.lower())
            cmd = [hipcc] + cflags + ["-DSCCBF_ONLY_" + tag] + extra + list(extra_flags) + ["-o", obj, SRC]
            if verbose:
                print(" ".join(cmd))
            objs.append(obj)
            procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        logs = [p.communicate()[0] for p in procs]
        if any(p.returncode != 0 for p in procs):
            raise RuntimeError("hipcc failed:\n" + "\n".join(logs))
        lib_tmp = os.path.join(tmp, "libsccbf.so")
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_tmp] + objs
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc link failed:\n" + res.stdout)
        os.replace(lib_tmp, out)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
