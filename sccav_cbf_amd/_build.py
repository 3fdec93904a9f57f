"""Builds sccav_cbf_amd/lib/libsccbf.so (HIP kernels + C ABI) for gfx950 with hipcc."""
import os
import shutil
import subprocess

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


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [HDR] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hip.h"))]
    return any(os.path.getmtime(f) > t for f in deps)


def build_library(force=False, verbose=False, out=None, extra_flags=()):
    """Compile the shared library in-tree; returns its path.  `out` / `extra_flags`: a variant build beside
    the library (for A/B measurements: load it with SCCBF_LIBRARY=<path>), always compiled."""
    if out is None and not force and not needs_build():
        return LIB
    variant = out is not None
    out = out or LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIB_DIR, exist_ok=True)
    # the one source file in three parts, side by side; then one link
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-I" + os.path.join(_ROOT, "include"), "-c"]
    objs, procs = [], []
    for tag, extra in PARTS:
        obj = os.path.join(LIB_DIR, "sccbf_%s%s.o" % (tag.lower(), "_variant" if variant else ""))
        cmd = [hipcc] + cflags + ["-DSCCBF_ONLY_" + tag] + extra + list(extra_flags) + ["-o", obj, SRC]
        if verbose:
            print(" ".join(cmd))
        objs.append(obj)
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate()[0] for p in procs]
    if any(p.returncode != 0 for p in procs):
        raise RuntimeError("hipcc failed:\n" + "\n".join(logs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout)
    for obj in objs:
        os.remove(obj)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
