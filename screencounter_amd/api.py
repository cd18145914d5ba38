"""Host-side mirror of the reference's operator interface for the hot path, at the Rcpp level.

Same names, positional arguments and return tuples as the functions R reaches through ``.Call`` (R/RcppExports.R:4-26
of the reference): ``count_single_barcodes``, ``count_combo_barcodes_single``, ``count_dual_barcodes``,
``match_barcodes``, their siblings and the many-files entries -- 0-based indices, int32 counts, scalar totals, exactly
what the ``Rcpp::List`` results hold.  The R functions above that level (R/count*.R) stay what they are in the
reference and are not part of this package; the tests keep a small mirror of them (tests/rlevel.py) so that the
reference's own test vectors can be written the way its tests write them.

Everything is computed by libscg on the GPU; errors the reference raises as R errors surface as ``ScgError``.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Sequence

import numpy as np

from . import _lib
from ._lib import ScgError, check, cstr_array, errbuf

# =============================================================================================
# Rcpp level
# =============================================================================================
def count_single_barcodes(path: str, constant: str, strand: int, pool: Sequence[str], mismatches: int,
                          use_first: bool, nthreads: int = 1):
    """src/count_single_barcodes.cpp:28-50 -> (counts int32[len(pool)], total)."""
    L = _lib.load()
    counts = np.zeros(max(len(pool), 1), dtype=np.int32)
    total = C.c_int32(0)
    err = errbuf()
    parr, _keep = cstr_array(pool)
    check(L.scg_count_single_barcodes(os.fspath(path).encode(), constant.encode(), int(strand), parr, len(pool),
                                      int(mismatches), int(bool(use_first)), int(nthreads),
                                      counts.ctypes.data_as(_lib.i32_p), C.byref(total), err, _lib.ERRCAP), err)
    return counts[:len(pool)].copy(), int(total.value)


def count_combo_barcodes_single(path: str, constant: str, strand: int, pool: Sequence[Sequence[str]], mismatches: int,
                                use_first: bool, nthreads: int = 1):
    """src/count_combo_barcodes_single.cpp:39-70 -> (indices int32[2, K] 0-based, freq int32[K], total)."""
    if len(pool) != 2:
        # src/count_combo_barcodes_single.cpp:44-46
        raise ScgError(_lib.SCG_ERR_INVALID, "currently expecting only 2 variable regions for single-end combinatorial barcodes")
    L = _lib.load()
    idx_p = _lib.i32_p()
    freq_p = _lib.i32_p()
    k = C.c_int64(0)
    total = C.c_int32(0)
    err = errbuf()
    p0, _k0 = cstr_array(pool[0])
    p1, _k1 = cstr_array(pool[1])
    check(L.scg_count_combo_barcodes_single(os.fspath(path).encode(), constant.encode(), int(strand), p0, len(pool[0]), p1, len(pool[1]),
                                            int(mismatches), int(bool(use_first)), int(nthreads),
                                            C.byref(idx_p), C.byref(freq_p), C.byref(k), C.byref(total), err, _lib.ERRCAP), err)
    K = int(k.value)
    try:
        idx = np.ctypeslib.as_array(idx_p, shape=(max(2 * K, 1),))[:2 * K].reshape(K, 2).T.copy()
        freq = np.ctypeslib.as_array(freq_p, shape=(max(K, 1),))[:K].copy()
    finally:
        L.scg_free(idx_p)
        L.scg_free(freq_p)
    return idx.astype(np.int32), freq.astype(np.int32), int(total.value)


@contextlib.contextmanager
def _devices_env(devices):
    """`devices=` of the matrixOf* mirrors -> scg_set_devices for the duration of one native call on this thread (the C
    ABI's device list; an id may repeat to keep several files in flight on one card).  Nothing touches the process
    environment: libscg's worker threads read it while a call runs."""
    if devices is None:
        yield
        return
    L = _lib.load()
    err = errbuf()
    ids = [int(d) for d in devices]
    arr = (C.c_int * max(len(ids), 1))(*ids)
    check(L.scg_set_devices(arr, len(ids), err, _lib.ERRCAP), err)
    try:
        yield
    finally:
        L.scg_set_devices(None, 0, err, _lib.ERRCAP)


def count_single_barcodes_files(paths: Sequence[str], constant: str, strand: int, pool: Sequence[str], mismatches: int,
                                use_first: bool, nthreads: int = 1, devices=None):
    """scg_count_single_barcodes_files: every file of matrixOfSingleBarcodes in one native call
    -> (counts int32[len(pool), len(paths)], totals list)."""
    L = _lib.load()
    n = len(paths)
    counts = np.zeros((max(n, 1), max(len(pool), 1)), dtype=np.int32)      # row f = column f of the column-major output
    totals = (C.c_int32 * max(n, 1))()
    err = errbuf()
    parr, _keep = cstr_array(pool)
    farr, _fk = cstr_array([os.fspath(x) for x in paths])
    with _devices_env(devices):
        check(L.scg_count_single_barcodes_files(farr, n, constant.encode(), int(strand), parr, len(pool), int(mismatches),
                                                int(bool(use_first)), int(nthreads), _files_out(counts, len(pool)), totals,
                                                err, _lib.ERRCAP), err)
    return _files_matrix(counts, len(pool), n), [int(totals[i]) for i in range(n)]


def _files_out(buf: np.ndarray, n_pool: int):
    """A contiguous int32 buffer of n_pool * n_files entries inside `buf` (allocated with >= 1 row / column)."""
    return buf.reshape(-1).ctypes.data_as(_lib.i32_p)


def _files_matrix(buf: np.ndarray, n_pool: int, n_files: int) -> np.ndarray:
    flat = buf.reshape(-1)[: n_pool * n_files]
    return flat.reshape(n_files, n_pool).T.copy()          # column-major n_pool x n_files -> [n_pool, n_files]


def count_combo_barcodes_single_files(paths: Sequence[str], constant: str, strand: int, pool: Sequence[Sequence[str]], mismatches: int,
                                      use_first: bool, nthreads: int = 1, devices=None):
    """scg_count_combo_barcodes_single_files -> list of (indices int32[2, K], freq int32[K], total), one per file."""
    if len(pool) != 2:
        raise ScgError(_lib.SCG_ERR_INVALID, "currently expecting only 2 variable regions for single-end combinatorial barcodes")
    L = _lib.load()
    n = len(paths)
    idx = (_lib.i32_p * max(n, 1))()
    freq = (_lib.i32_p * max(n, 1))()
    ks = (C.c_int64 * max(n, 1))()
    totals = (C.c_int32 * max(n, 1))()
    err = errbuf()
    p0, _k0 = cstr_array(pool[0])
    p1, _k1 = cstr_array(pool[1])
    farr, _fk = cstr_array([os.fspath(x) for x in paths])
    with _devices_env(devices):
        check(L.scg_count_combo_barcodes_single_files(farr, n, constant.encode(), int(strand), p0, len(pool[0]), p1, len(pool[1]),
                                                      int(mismatches), int(bool(use_first)), int(nthreads), idx, freq, ks, totals,
                                                      err, _lib.ERRCAP), err)
    out = []
    try:
        for f in range(n):
            K = int(ks[f])
            i2 = np.ctypeslib.as_array(idx[f], shape=(max(2 * K, 1),))[:2 * K].reshape(K, 2).T.copy() if K else np.zeros((2, 0), dtype=np.int32)
            fr = np.ctypeslib.as_array(freq[f], shape=(max(K, 1),))[:K].copy() if K else np.zeros(0, dtype=np.int32)
            out.append((i2.astype(np.int32), fr.astype(np.int32), int(totals[f])))
    finally:
        for f in range(n):
            L.scg_free(idx[f])
            L.scg_free(freq[f])
    return out


def count_dual_barcodes_files(paths1: Sequence[str], constant1: str, reverse1: bool, mismatches1: int, pool1: Sequence[str],
                              paths2: Sequence[str], constant2: str, reverse2: bool, mismatches2: int, pool2: Sequence[str],
                              randomized: bool, use_first: bool, nthreads: int = 1, devices=None):
    """scg_count_dual_barcodes_files -> (counts int32[len(pool1), n_files], totals list)."""
    if len(pool1) != len(pool2):
        raise ScgError(_lib.SCG_ERR_INVALID, "both barcode pools should be of the same length")
    if len(paths1) != len(paths2):
        raise ValueError("paths1 and paths2 differ in length")
    L = _lib.load()
    n = len(paths1)
    counts = np.zeros((max(n, 1), max(len(pool1), 1)), dtype=np.int32)
    totals = (C.c_int32 * max(n, 1))()
    err = errbuf()
    a1, _k1 = cstr_array(pool1)
    a2, _k2 = cstr_array(pool2)
    f1, _fk1 = cstr_array([os.fspath(x) for x in paths1])
    f2, _fk2 = cstr_array([os.fspath(x) for x in paths2])
    with _devices_env(devices):
        check(L.scg_count_dual_barcodes_files(f1, constant1.encode(), int(bool(reverse1)), int(mismatches1), a1,
                                              f2, constant2.encode(), int(bool(reverse2)), int(mismatches2), a2, len(pool1), n,
                                              int(bool(randomized)), int(bool(use_first)), int(nthreads),
                                              _files_out(counts, len(pool1)), totals, err, _lib.ERRCAP), err)
    return _files_matrix(counts, len(pool1), n), [int(totals[i]) for i in range(n)]


def count_dual_barcodes(path1: str, constant1: str, reverse1: bool, mismatches1: int, pool1: Sequence[str],
                        path2: str, constant2: str, reverse2: bool, mismatches2: int, pool2: Sequence[str],
                        randomized: bool, use_first: bool, diagnostics: bool = False, nthreads: int = 1):
    """src/count_dual_barcodes.cpp:74-117 -> (counts int32[n pairs], total), or with diagnostics=True the
    five outputs of the include.invalid=TRUE branch: (counts, (indices int32[2, K] 0-based, freq), total,
    barcode1_only, barcode2_only)."""
    if len(pool1) != len(pool2):
        # kaori/handlers/DualBarcodesPairedEnd.hpp:106-109
        raise ScgError(_lib.SCG_ERR_INVALID, "both barcode pools should be of the same length")
    L = _lib.load()
    if diagnostics:
        counts = np.zeros(max(len(pool1), 1), dtype=np.int32)
        idx_p, freq_p = _lib.i32_p(), _lib.i32_p()
        k = C.c_int64(0)
        total, b1, b2 = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        err = errbuf()
        p1, _k1 = cstr_array(pool1)
        p2, _k2 = cstr_array(pool2)
        check(L.scg_count_dual_barcodes_diagnostics(os.fspath(path1).encode(), constant1.encode(), int(bool(reverse1)), int(mismatches1), p1,
                                                    os.fspath(path2).encode(), constant2.encode(), int(bool(reverse2)), int(mismatches2), p2,
                                                    len(pool1), int(bool(randomized)), int(bool(use_first)), int(nthreads),
                                                    counts.ctypes.data_as(_lib.i32_p), C.byref(idx_p), C.byref(freq_p), C.byref(k),
                                                    C.byref(total), C.byref(b1), C.byref(b2), err, _lib.ERRCAP), err)
        K = int(k.value)
        try:
            idx = np.ctypeslib.as_array(idx_p, shape=(max(2 * K, 1),))[:2 * K].reshape(K, 2).T.copy()
            freq = np.ctypeslib.as_array(freq_p, shape=(max(K, 1),))[:K].copy()
        finally:
            L.scg_free(idx_p)
            L.scg_free(freq_p)
        return counts[:len(pool1)].copy(), (idx.astype(np.int32), freq.astype(np.int32)), int(total.value), int(b1.value), int(b2.value)
    counts = np.zeros(max(len(pool1), 1), dtype=np.int32)
    total = C.c_int32(0)
    err = errbuf()
    p1, _k1 = cstr_array(pool1)
    p2, _k2 = cstr_array(pool2)
    check(L.scg_count_dual_barcodes(os.fspath(path1).encode(), constant1.encode(), int(bool(reverse1)), int(mismatches1), p1,
                                    os.fspath(path2).encode(), constant2.encode(), int(bool(reverse2)), int(mismatches2), p2,
                                    len(pool1), int(bool(randomized)), int(bool(use_first)), int(bool(diagnostics)), int(nthreads),
                                    counts.ctypes.data_as(_lib.i32_p), C.byref(total), err, _lib.ERRCAP), err)
    return counts[:len(pool1)].copy(), int(total.value)


def count_random_barcodes(path: str, constant: str, strand: int, mismatches: int, use_first: bool, nthreads: int = 1):
    """src/count_random_barcodes.cpp:41-62 -> ((sequences list[str] sorted, freq int32[K]), total)."""
    L = _lib.load()
    seq_p = C.c_void_p()
    freq_p = _lib.i32_p()
    k, vlen, total = C.c_int64(0), C.c_int32(0), C.c_int32(0)
    err = errbuf()
    check(L.scg_count_random_barcodes(os.fspath(path).encode(), constant.encode(), int(strand), int(mismatches), int(bool(use_first)),
                                      int(nthreads), C.byref(seq_p), C.byref(freq_p), C.byref(k), C.byref(vlen), C.byref(total),
                                      err, _lib.ERRCAP), err)
    K, W = int(k.value), int(vlen.value)
    try:
        blob = C.string_at(seq_p, K * (W + 1)) if K else b""
        seqs = [blob[i * (W + 1): i * (W + 1) + W].decode("latin-1") for i in range(K)]
        freq = np.ctypeslib.as_array(freq_p, shape=(max(K, 1),))[:K].copy().astype(np.int32)
    finally:
        L.scg_free(seq_p)
        L.scg_free(freq_p)
    return (seqs, freq), int(total.value)


def count_dual_barcodes_single_end(path: str, constant: str, pools: Sequence[Sequence[str]], strand: int, mismatches: int,
                                   use_first: bool, diagnostics: bool = False, nthreads: int = 1):
    """src/count_dual_barcodes_single_end.cpp:53-87 -> (counts int32[n combinations], total), or with diagnostics=True
    (counts, (indices int32[2, K] 0-based, freq), total) as in its include.invalid=TRUE branch."""
    L = _lib.load()
    nch = len(pools[0]) if pools else 0
    counts = np.zeros(max(nch, 1), dtype=np.int32)
    total = C.c_int32(0)
    err = errbuf()
    rows, sizes, _keep = _lib.cstr_matrix(pools)
    if diagnostics:
        idx_p, freq_p = _lib.i32_p(), _lib.i32_p()
        k = C.c_int64(0)
        check(L.scg_count_dual_barcodes_single_end_diagnostics(os.fspath(path).encode(), constant.encode(), rows, sizes, len(pools),
                                                               int(strand), int(mismatches), int(bool(use_first)), int(nthreads),
                                                               counts.ctypes.data_as(_lib.i32_p), C.byref(idx_p), C.byref(freq_p), C.byref(k),
                                                               C.byref(total), err, _lib.ERRCAP), err)
        K = int(k.value)
        try:
            idx = np.ctypeslib.as_array(idx_p, shape=(max(2 * K, 1),))[:2 * K].reshape(K, 2).T.copy()
            freq = np.ctypeslib.as_array(freq_p, shape=(max(K, 1),))[:K].copy()
        finally:
            L.scg_free(idx_p)
            L.scg_free(freq_p)
        return counts[:nch].copy(), (idx.astype(np.int32), freq.astype(np.int32)), int(total.value)
    check(L.scg_count_dual_barcodes_single_end(os.fspath(path).encode(), constant.encode(), rows, sizes, len(pools),
                                               int(strand), int(mismatches), int(bool(use_first)), int(bool(diagnostics)), int(nthreads),
                                               counts.ctypes.data_as(_lib.i32_p), C.byref(total), err, _lib.ERRCAP), err)
    return counts[:nch].copy(), int(total.value)


def count_combo_barcodes_paired(path1: str, constant1: str, reverse1: bool, mismatches1: int, pool1: Sequence[str],
                                path2: str, constant2: str, reverse2: bool, mismatches2: int, pool2: Sequence[str],
                                randomized: bool, use_first: bool, nthreads: int = 1):
    """src/count_combo_barcodes_paired.cpp:57-95 -> (indices int32[2, K] 0-based sorted by (first, second),
    freq int32[K], total, barcode1_only, barcode2_only)."""
    L = _lib.load()
    idx_p, freq_p = _lib.i32_p(), _lib.i32_p()
    k = C.c_int64(0)
    total, b1, b2 = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    err = errbuf()
    p1, _k1 = cstr_array(pool1)
    p2, _k2 = cstr_array(pool2)
    check(L.scg_count_combo_barcodes_paired(os.fspath(path1).encode(), constant1.encode(), int(bool(reverse1)), int(mismatches1), p1, len(pool1),
                                            os.fspath(path2).encode(), constant2.encode(), int(bool(reverse2)), int(mismatches2), p2, len(pool2),
                                            int(bool(randomized)), int(bool(use_first)), int(nthreads),
                                            C.byref(idx_p), C.byref(freq_p), C.byref(k),
                                            C.byref(total), C.byref(b1), C.byref(b2), err, _lib.ERRCAP), err)
    K = int(k.value)
    try:
        idx = np.ctypeslib.as_array(idx_p, shape=(max(2 * K, 1),))[:2 * K].reshape(K, 2).T.copy()
        freq = np.ctypeslib.as_array(freq_p, shape=(max(K, 1),))[:K].copy()
    finally:
        L.scg_free(idx_p)
        L.scg_free(freq_p)
    return idx.astype(np.int32), freq.astype(np.int32), int(total.value), int(b1.value), int(b2.value)


def match_barcodes(sequences: Sequence[str], choices: Sequence[str], substitutions: int = 0, reverse: bool = False):
    """src/match_barcodes.cpp:6-37 -> (index int32[n] 0-based with -1 for NA, mismatches int32[n] with -1 for NA)."""
    L = _lib.load()
    n = len(sequences)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    mm = np.zeros(max(n, 1), dtype=np.int32)
    err = errbuf()
    sa, _k = cstr_array(sequences)
    ca, _k2 = cstr_array(choices)
    check(L.scg_match_barcodes(sa, n, ca, len(choices), int(substitutions), int(bool(reverse)),
                               idx.ctypes.data_as(_lib.i32_p), mm.ctypes.data_as(_lib.i32_p), err, _lib.ERRCAP), err)
    return idx[:n].copy(), mm[:n].copy()


def parse_fastq(path: str):
    """kaori/FastqReader.hpp:42-110 -> (uint8 sequence bytes, uint64 offsets[n + 1])."""
    L = _lib.load()
    seqs_p = C.c_void_p()
    offs_p = C.c_void_p()
    n = C.c_int64(0)
    err = errbuf()
    check(L.scg_parse_fastq(os.fspath(path).encode(), C.byref(seqs_p), C.byref(offs_p), C.byref(n), err, _lib.ERRCAP), err)
    try:
        offs = np.ctypeslib.as_array(C.cast(offs_p, C.POINTER(C.c_uint64)), shape=(n.value + 1,)).copy()
        nb = int(offs[-1])
        seqs = np.ctypeslib.as_array(C.cast(seqs_p, C.POINTER(C.c_uint8)), shape=(max(nb, 1),))[:nb].copy()
    finally:
        L.scg_free(seqs_p)
        L.scg_free(offs_p)
    return seqs, offs
