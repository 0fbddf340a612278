"""Build recipe for ``oracle/_ref/`` -- TEST INFRASTRUCTURE ONLY, never imported by the product.

The reference's loss/metric path is pure Python, so the only compiled reference code relevant
to SURVEY.md section 8 is its SVMrank parser (row f-3), one self-contained C header.  This
recipe compiles it with gcc straight from ``/root/reference`` (sources are NOT copied) through
``oracle/svmrank_ref_shim.c`` into ``oracle/_ref/libsvmrank_ref.so``.  ``oracle/_ref/`` is
git-ignored (binaries stay out of history) but travels to the GPU box with the tree.

    python -m oracle.build_ref          # build if the reference is present
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("LTR_REFERENCE_ROOT", "/root/reference")
REF_PARSER_DIR = os.path.join(REF_ROOT, "pytorchltr", "datasets", "svmrank", "parser")
REF_DIR = os.path.join(_HERE, "_ref")
SHIM = os.path.join(_HERE, "svmrank_ref_shim.c")
LIB_PATH = os.path.join(REF_DIR, "libsvmrank_ref.so")


def reference_present():
    return os.path.exists(os.path.join(REF_PARSER_DIR, "svmrank_parser.h"))


def build(force=False):
    """Compile the reference parser when its sources are here; returns the .so path or None
    (None = neither the sources nor a prebuilt library are available)."""
    if reference_present():
        header = os.path.join(REF_PARSER_DIR, "svmrank_parser.h")
        stale = (not os.path.exists(LIB_PATH)
                 or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(header), os.path.getmtime(SHIM)))
        if force or stale:
            os.makedirs(REF_DIR, exist_ok=True)
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu99", "-w",
                                   "-I", REF_PARSER_DIR, "-o", LIB_PATH, SHIM, "-lm"])
    return LIB_PATH if os.path.exists(LIB_PATH) else None


_lib = None


def _load():
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            raise FileNotFoundError("oracle/_ref/libsvmrank_ref.so is not built and the reference "
                                    "sources are not present")
        _lib = ctypes.CDLL(path)
        _lib.ref_svmrank_parse.restype = ctypes.c_int
        _lib.ref_svmrank_parse.argtypes = [
            ctypes.c_char_p, ctypes.POINTER(ctypes.POINTER(ctypes.c_double)),
            ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t),
            ctypes.POINTER(ctypes.POINTER(ctypes.c_int)), ctypes.POINTER(ctypes.POINTER(ctypes.c_long))]
        _lib.ref_svmrank_free.restype = None
        _lib.ref_svmrank_free.argtypes = [ctypes.c_void_p]
    return _lib


def parse_svmrank_file(path):
    """The reference parser's (status, xs float64 (rows, cols), ys int32, qids int64)."""
    lib = _load()
    xs = ctypes.POINTER(ctypes.c_double)()
    ys = ctypes.POINTER(ctypes.c_int)()
    qids = ctypes.POINTER(ctypes.c_long)()
    rows, cols = ctypes.c_size_t(0), ctypes.c_size_t(0)
    rc = lib.ref_svmrank_parse(os.fsencode(path), ctypes.byref(xs), ctypes.byref(rows),
                               ctypes.byref(cols), ctypes.byref(ys), ctypes.byref(qids))
    if rc != 0:
        return rc, None, None, None
    r, c = rows.value, cols.value
    X = np.ctypeslib.as_array(xs, shape=(r, c)).copy() if r * c else np.zeros((r, c))
    Y = np.ctypeslib.as_array(ys, shape=(r,)).copy() if r else np.zeros(0, np.int32)
    Q = np.ctypeslib.as_array(qids, shape=(r,)).copy() if r else np.zeros(0, np.int64)
    for p in (xs, ys, qids):
        lib.ref_svmrank_free(ctypes.cast(p, ctypes.c_void_p))
    return 0, X, Y.astype(np.int32), Q.astype(np.int64)


if __name__ == "__main__":
    print("reference parser:", build(force=True))
