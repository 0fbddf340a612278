"""Build the native libraries in-tree:  python -m pytorchltr_amd.build [--force]

Cross-compiles for gfx950 with hipcc (no GPU needed to build).  The resulting
``pytorchltr_amd/csrc/libltr_hip.so`` is git-ignored but travels with the tree.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libltr_hip.so")
SOURCES = [os.path.join(CSRC, "ltr_kernels.hip")]
# every .inc the translation unit includes, and the public header
DEPENDS = SOURCES + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".inc")) + \
    [os.path.join(_ROOT, "include", "ltr_hip.h")]
ARCH = "gfx950"
IO_LIB_PATH = os.path.join(CSRC, "libltr_io.so")
IO_SOURCES = [os.path.join(CSRC, "svmrank_parser.cpp")]
IO_DEPENDS = IO_SOURCES + [os.path.join(_ROOT, "include", "ltr_io.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > built for d in DEPENDS)


def build_extension(force=False, verbose=False):
    """Compile every HIP source into libltr_hip.so for gfx950.  Returns the .so path."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
           "-I", os.path.join(_ROOT, "include"), "-I", CSRC,
           "-o", LIB_PATH] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


def build_io(force=False, verbose=False):
    """Compile the host-side ingestion library (include/ltr_io.h) with g++.  Returns its path."""
    stale = (not os.path.exists(IO_LIB_PATH)
             or any(os.path.getmtime(d) > os.path.getmtime(IO_LIB_PATH) for d in IO_DEPENDS))
    if force or stale:
        cmd = [os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
               "-Wall", "-I", os.path.join(_ROOT, "include"), "-o", IO_LIB_PATH] + IO_SOURCES
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return IO_LIB_PATH


if __name__ == "__main__":
    path = build_extension(force="--force" in sys.argv, verbose=True)
    print("built", path)
    print("built", build_io(force="--force" in sys.argv, verbose=True))
