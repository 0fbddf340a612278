"""ctypes binding of the C ABI in ``include/ltr_hip.h`` (pytorchltr_amd/csrc/libltr_hip.so).

PyTorch supplies device memory and the current HIP stream; every call here hands raw device
pointers to the library.  The library must exist -- there is deliberately no fallback.
"""
import ctypes
import os

import torch  # noqa: F401  (loads the HIP runtime the extension binds to, before dlopen)

_HERE = os.path.dirname(os.path.abspath(__file__))
# LTR_HIP_LIB: another build of the same library (tuning variants, scripts/build_variants.sh)
LIB_PATH = os.environ.get("LTR_HIP_LIB") or os.path.join(_HERE, "csrc", "libltr_hip.so")

# enum ltr_loss_kind
HINGE, DCG_HINGE, LOGISTIC, ARP1, ARP2, NDCG1, NDCG2 = range(7)
# enum ltr_label_dtype
LABEL_I64, LABEL_F32, LABEL_I32 = 0, 1, 2
ERR_CONFIG = -6
ERR_TIMEOUT = -7
# ltr_linear_fused_plan (include/ltr_hip.h)
PLAN_NONE, PLAN_REGISTER_TILE, PLAN_CLUSTER, PLAN_GENERAL, PLAN_PARTS = 0, 1, 2, 3, 4

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/ltr_hip.h line by line
SIGNATURES = {
    "ltr_version": (_i, []),
    "ltr_error_string": (ctypes.c_char_p, [_i]),
    "ltr_max_list_len": (_i, []),
    "ltr_max_list_len_f64": (_i, []),
    "ltr_device_status": (_i, [_i]),
    "ltr_debug_force_timeout": (None, [_i]),
    "ltr_debug_cluster_mode": (None, [_i]),
    "ltr_debug_parts_all": (_i, [_i]),
    "ltr_exchange_release": (_i, []),
    "ltr_debug_set_exchange_tag": (_i, [_vp, ctypes.c_uint32]),
    "ltr_debug_stream_probe_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ltr_debug_kernel_events": (_i, [_vp, _vp]),
    "ltr_debug_mlp_layout": (None, [_i]),
    "ltr_debug_mlp_probe_f32": (_i, [_vp] * 8 + [_i] * 5 + [_vp, _vp]),
    "ltr_pairwise_loss_f32": (_i, [_i, _f, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "ltr_pairwise_loss_f32_cfg": (_i, [_i, _f, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    "ltr_pairwise_loss_workspace_bytes": (_sz, [_i, _i, _i]),
    "ltr_pairwise_loss_ws_f32": (_i, [_i, _f, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "ltr_scale_rows_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ltr_scale_rows_uniform_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ltr_pairwise_loss_f64": (_i, [_i, ctypes.c_double, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "ltr_scale_rows_f64": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ltr_rank_by_score_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ltr_dcg_f32": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ltr_arp_f32": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "ltr_rank_by_score_tie_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "ltr_dcg_tie_f32": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ltr_arp_tie_f32": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "ltr_rank_by_score_seed_f32": (_i, [_vp, _vp, ctypes.c_uint64, _vp, _i, _i, _vp, _vp]),
    "ltr_dcg_seed_f32": (_i, [_vp, _vp, _i, _vp, ctypes.c_uint64, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ltr_arp_seed_f32": (_i, [_vp, _vp, _i, _vp, ctypes.c_uint64, _vp, _i, _i, _vp, _vp]),
    "ltr_tie_hash_word": (ctypes.c_uint32, [ctypes.c_uint64, ctypes.c_uint32]),
    "ltr_listwise_softmax_f32": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "ltr_mask_padded_values_f32": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp]),
    "ltr_batch_pairs": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ltr_plackettluce_keys_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "ltr_pbm_clicks": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "ltr_collate_pad_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ltr_collate_pad_csr_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ltr_linear_workspace_bytes": (_sz, [_i, _i, _i]),
    "ltr_linear_fused_plan": (_i, [_i, _i, _i, _i]),
    "ltr_overlap_create": (_i, [_vp, _vp, _i, _vp]),
    "ltr_overlap_destroy": (_i, [_vp]),
    "ltr_overlap_wait": (_i, [_vp, _i, _vp]),
    "ltr_overlap_flush": (_i, [_vp]),
    "ltr_linear_step_f32": (_i, [_i, _f, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _sz,
                                _vp, _i, _vp]),
    "ltr_linear_sgd_step_f32": (_i, [_i, _f, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _sz,
                                    _vp, _vp]),
    "ltr_linear_sgd_lazy_step_f32": (_i, [_i, _f, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _sz, _i, _vp]),
    "ltr_linear_sgd_flush_f32": (_i, [_i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "ltr_linear_sgd_lazy_step_dp_f32": (_i, [_i, _f, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _sz, _i, _f, _vp, _i, _vp, _vp]),
    "ltr_linear_sgd_flush_dp_f32": (_i, [_i, _vp, _vp, _i, _i, _i, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "ltr_linear_lazy_rows_reduce_f32": (_i, [_i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "ltr_debug_fake_allreduce": (_i, [_vp, _vp, _sz, _i, _i, _vp, _vp]),
    "ltr_mailbox_create": (_i, [_i, _i, _i, _vp, _vp]),
    "ltr_mailbox_connect": (_i, [_vp, _vp]),
    "ltr_mailbox_destroy": (_i, [_vp]),
    "ltr_mailbox_allreduce": (_i, [_vp, _vp, _sz, _i, _i, _vp, _vp]),
    "ltr_debug_mailbox_state": (_i, [_vp, ctypes.c_longlong, ctypes.c_longlong]),
    "ltr_mailbox_set_timeout_ms": (ctypes.c_longlong, [ctypes.c_longlong]),
    "ltr_linear_pairwise_f32": (_i, [_i, _f, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i,
                                     _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ltr_linear_partials_f32": (_i, [_i, _f, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i,
                                     _vp, _vp, _vp, _vp]),
    "ltr_linear_reduce_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ltr_linear_reduce_bcast_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ltr_linear_reduce_loss_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "ltr_linear_reduce_accum_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "ltr_linear_scores_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ltr_linear_grad_workspace_bytes": (_sz, [_i, _i, _i]),
    "ltr_linear_grad_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ltr_mlp_max_list_len": (_i, [_i]),
    "ltr_mlp_param_count": (_sz, [_i, _i, _i]),
    "ltr_mlp_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ltr_mlp_scores_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ltr_mlp_pairwise_f32": (_i, [_i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp,
                                  _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None


class ExtensionMissingError(ImportError):
    pass


class _Library(ctypes.CDLL):
    """ctypes.CDLL that says WHY a test / tuning hook is missing: a production build (LTR_NO_DEBUG_HOOKS=1) does not
    export the ltr_debug_* entry points (the bare AttributeError came up far from its cause)."""

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name.startswith("ltr_debug_"):
                raise AttributeError(
                    "%s is a test / tuning hook and this libltr_hip.so was built without them (LTR_NO_DEBUG_HOOKS=1); "
                    "rebuild with `python -m pytorchltr_amd.build --force` and the variable unset" % name) from None
            raise


def lib():
    """The loaded extension.  Raises loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ExtensionMissingError(
                "pytorchltr_amd HIP extension not found at %s; build it with "
                "`python -m pytorchltr_amd.build` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        handle = _Library(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            if name.startswith("ltr_debug_") and not hasattr(handle, name):
                continue                     # a production build (-DLTR_NO_DEBUG_HOOKS) leaves the test hooks out
            fn = getattr(handle, name)       # AttributeError if the symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().ltr_error_string(int(rc))
        raise RuntimeError("pytorchltr_amd: %s (code %d)" % (msg.decode() if msg else "?", rc))


def ptr(t):
    return None if t is None else t.data_ptr()


try:                                    # raw stream handle without building a Stream object
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                  # pragma: no cover - older/newer torch
    _raw_stream = None


def stream_of(t):
    """hipStream_t (as int) of torch's current stream on t's device."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _NullCtx()


def device_ctx(t):
    """Context that makes t's device current for the launch; free when it already is."""
    idx = t.device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NULL
    return torch.cuda.device(idx)


_max_len = None
_max_len_f64 = None


def max_list_len():
    global _max_len
    if _max_len is None:
        _max_len = int(lib().ltr_max_list_len())
    return _max_len


def max_list_len_f64():
    global _max_len_f64
    if _max_len_f64 is None:
        _max_len_f64 = int(lib().ltr_max_list_len_f64())
    return _max_len_f64


def device_status(clear=True, synchronize=True):
    """The sticky device status word (include/ltr_hip.h: ltr_device_status): raises RuntimeError if
    a multi-workgroup kernel of an earlier launch gave up waiting for its partners (its outputs are
    NaN-poisoned).  `synchronize` waits for the current device first, so the answer covers every
    launch issued so far."""
    if synchronize and torch.cuda.is_available():
        torch.cuda.synchronize()
    check(lib().ltr_device_status(1 if clear else 0))


def require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            "pytorchltr_amd: `%s` is on %s; this library only runs its HIP kernels on a ROCm "
            "device (no CPU fallback). Move the batch with .to('cuda')." % (what, t.device))


def label_dtype(rel):
    if rel.dtype == torch.int64:
        return LABEL_I64
    if rel.dtype == torch.float32:
        return LABEL_F32
    if rel.dtype == torch.int32:
        return LABEL_I32
    raise TypeError("unsupported relevance dtype %s" % rel.dtype)
