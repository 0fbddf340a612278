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
# three translation units, compiled side by side and linked into one library: the loss / metric /
# helper kernels with the C ABI, the fused Linear scorer + loss kernels, the fused MLP + scorer layer
SOURCES = [os.path.join(CSRC, f) for f in ("ltr_kernels.hip", "ltr_linear.hip", "ltr_mlp.hip")]
SOURCE_FLAGS = {}          # per-source compiler flags (none at present)
OBJ_DIR = os.path.join(_ROOT, "build", "obj")
# every .inc the translation units include, and the public header
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


def build_extension(force=False, verbose=False, extra_flags=(), lib_path=None):
    """Compile every HIP source into libltr_hip.so for gfx950.  Returns the .so path."""
    lib_path = lib_path or LIB_PATH
    if not force and lib_path == LIB_PATH and not is_stale():
        return LIB_PATH
    obj_dir = OBJ_DIR if lib_path == LIB_PATH else lib_path + ".obj"
    os.makedirs(obj_dir, exist_ok=True)
    common = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
              "-pthread", "-Wall", "-Wno-unused-function", "-I", os.path.join(_ROOT, "include"), "-I", CSRC] + list(extra_flags)
    if os.environ.get("LTR_NO_DEBUG_HOOKS") == "1":      # production build: the ltr_debug_* test hooks are not exported
        common.append("-DLTR_NO_DEBUG_HOOKS")
    jobs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        cmd = common + SOURCE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd), obj))
    objs = []
    failed = None
    for cmd, proc, obj in jobs:                  # every compiler is waited for, whatever the others returned
        if proc.wait() != 0 and failed is None:
            failed = (proc.returncode, cmd)
        objs.append(obj)
    if failed is not None:
        for obj in objs:                         # (no stale object may be linked by a later, partial build)
            try:
                os.remove(obj)
            except OSError:
                pass
        raise subprocess.CalledProcessError(*failed)
    link = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc", "-pthread", "-o", lib_path] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return lib_path


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
