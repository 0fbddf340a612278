"""ctypes binding of ``include/ltr_io.h`` (pytorchltr_amd/csrc/libltr_io.so): host-side dataset
ingestion.  No torch, no GPU; like the device library it must have been built -- no fallback."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LTR_IO_LIB: another build of the same library (e.g. the sanitizer build of scripts/sanitize_host.sh)
LIB_PATH = os.environ.get("LTR_IO_LIB") or os.path.join(_HERE, "csrc", "libltr_io.so")

OK, FILE_ERROR, FORMAT_ERROR, MEMORY_ERROR, ARG_ERROR = range(5)

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/ltr_io.h line by line
SIGNATURES = {
    "ltr_svmrank_open": (_i, [ctypes.c_char_p, _i, ctypes.POINTER(_vp), ctypes.POINTER(_sz),
                              ctypes.POINTER(_sz)]),
    "ltr_svmrank_read": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "ltr_svmrank_close": (None, [_vp]),
    "ltr_io_error_string": (ctypes.c_char_p, [_i]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "pytorchltr_amd: %s is missing. Build it with `python -m pytorchltr_amd.build`."
                % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib
