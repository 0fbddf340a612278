"""SVMrank files -> device-resident query storage.

``parse_svmrank_file`` has the reference's name, return value and errors
(pytorchltr/datasets/svmrank/parser/svmrank_parser.pyx:22-58): ``(xs float64 (rows, cols),
ys int32 (rows,), qids int64 (rows,))``, ``OSError`` for an unreadable file, ``ValueError`` for
a malformed one.  It is backed by the multi-threaded C++ parser in csrc/svmrank_parser.cpp
(C ABI: include/ltr_io.h) instead of the reference's single-threaded DFA.

``load_svmrank`` is the ``SVMRankDataset(file, normalize=, filter_queries=)`` constructor
(pytorchltr/datasets/svmrank/svmrank.py:46-103, dense path) producing a ``RaggedQueries`` --
the whole split on the GPU -- whose ``collate_fn`` yields the reference's padded batches.
"""
import ctypes
import errno as _errno
import os

import numpy as _np

from pytorchltr_amd import _io


def parse_svmrank_file(path, n_threads=0, dtype=_np.float64):
    """Parses an SVMrank file into dense arrays.

    Args:
        path: file to read.
        n_threads: parser threads (<= 0: hardware concurrency, at most one per MiB of input).
        dtype: ``numpy.float64`` (the reference's) or ``numpy.float32`` for the feature matrix.

    Returns:
        ``(xs, ys, qids)`` -- (rows, cols) features, (rows,) int32 labels, (rows,) int64 qids.
    """
    dtype = _np.dtype(dtype)
    if dtype not in (_np.dtype(_np.float64), _np.dtype(_np.float32)):
        raise TypeError("dtype must be float64 or float32")
    lib = _io.lib()
    handle = ctypes.c_void_p()
    rows, cols = ctypes.c_size_t(0), ctypes.c_size_t(0)
    rc = lib.ltr_svmrank_open(os.fsencode(path), int(n_threads), ctypes.byref(handle),
                              ctypes.byref(rows), ctypes.byref(cols))
    if rc == _io.FILE_ERROR:
        raise OSError(_errno.ENOENT if not os.path.exists(path) else _errno.EIO,
                      "could not open file %s" % path)
    if rc == _io.FORMAT_ERROR:
        raise ValueError("could not parse file %s, not in SVMrank format" % path)
    if rc == _io.MEMORY_ERROR:
        raise OSError(_errno.ENOMEM, "could not allocate memory")
    if rc != _io.OK:
        raise RuntimeError(lib.ltr_io_error_string(rc).decode())
    try:
        xs = _np.empty((rows.value, cols.value), dtype=dtype)
        ys = _np.empty(rows.value, dtype=_np.int32)
        qids = _np.empty(rows.value, dtype=_np.int64)
        x64 = xs.ctypes.data if dtype == _np.float64 else None
        x32 = xs.ctypes.data if dtype == _np.float32 else None
        rc = lib.ltr_svmrank_read(handle, x64, x32, ys.ctypes.data, qids.ctypes.data)
        if rc != _io.OK:
            raise RuntimeError(lib.ltr_io_error_string(rc).decode())
    finally:
        lib.ltr_svmrank_close(handle)
    return xs, ys, qids


def query_offsets(qids):
    """Row offsets of the runs of equal qid (svmrank.py:72-74): (Q + 1,) int64."""
    qids = _np.asarray(qids)
    if qids.shape[0] == 0:
        return _np.zeros(1, dtype=_np.int64)
    cuts = _np.where(qids[1:] != qids[:-1])[0] + 1
    return _np.hstack([[0], cuts, [qids.shape[0]]]).astype(_np.int64)


def normalize_queries(xs, offsets):
    """Query-level min-max normalisation in place (svmrank.py:105-111): per query and feature,
    x <- (x - min) / (max - min), constant features -> 0.  Same two roundings per element as
    the reference (a subtraction, then a division), vectorised over all queries."""
    if xs.shape[0] == 0 or xs.shape[1] == 0:
        return xs
    starts = offsets[:-1]
    counts = _np.diff(offsets)
    lo = _np.minimum.reduceat(xs, starts, axis=0)
    xs -= _np.repeat(lo, counts, axis=0)
    hi = _np.maximum.reduceat(xs, starts, axis=0)
    hi[hi == 0.0] = 1.0
    xs /= _np.repeat(hi, counts, axis=0)
    return xs


def load_svmrank(file, normalize=False, filter_queries=False, device="cuda", n_threads=0, pad_features_to=1):
    """Loads an SVMrank file as a device-resident :class:`RaggedQueries`.

    Args:
        file: path of the dataset split.
        normalize: query-level feature normalisation (done in float64 like the reference, then
            rounded once to float32).
        filter_queries: drop queries without any relevant document (svmrank.py:87-96).
        device: ROCm device that will hold the split.
        n_threads: parser threads.
        pad_features_to: 4 keeps the rows of every collated batch a multiple of four floats apart (see RaggedQueries).
    """
    from pytorchltr_amd.datasets.ragged import RaggedQueries
    xs, ys, qids = parse_svmrank_file(file, n_threads=n_threads)
    offsets = query_offsets(qids)
    unique_qids = qids[offsets[:-1]]
    if normalize:
        normalize_queries(xs, offsets)
    if filter_queries and unique_qids.shape[0] > 0:
        label_sums = _np.add.reduceat(ys.astype(_np.int64), offsets[:-1])
        keep = label_sums > 0
        if not keep.all():
            counts = _np.diff(offsets)
            row_keep = _np.repeat(keep, counts)
            xs, ys = xs[row_keep], ys[row_keep]
            unique_qids = unique_qids[keep]
            offsets = _np.hstack([[0], _np.cumsum(counts[keep])]).astype(_np.int64)
    return RaggedQueries(xs.astype(_np.float32), ys.astype(_np.int64), offsets, qids=unique_qids,
                         device=device, pad_features_to=pad_features_to)
