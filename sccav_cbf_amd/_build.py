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
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # packed-fp32 forms have produced sporadic wrong results on this hardware pool
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
    # leading scalar kernel arguments arrive in SGPRs at wave launch instead of through s_load
    "-mllvm", "-amdgpu-kernarg-preload-count=16",
]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [HDR] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hip.h"))]
    return any(os.path.getmtime(f) > t for f in deps)


def build_library(force=False, verbose=False):
    """Compile the shared library in-tree; returns its path."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + ["-I" + os.path.join(_ROOT, "include"), "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
