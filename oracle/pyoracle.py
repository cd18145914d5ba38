"""ctypes bindings for the test oracle.  TEST INFRASTRUCTURE ONLY.

Two checkers live here and neither is ever used by the product path (screencounter_amd/):

* ``Oracle``   -- oracle/liboracle.so, the C restatement of the reference algorithm
                  (oracle/scg_oracle.c).  Always available (build: ``make -C oracle``).
* ``KaoriRef`` -- oracle/_ref/libkaori_ref.so, the real kaori headers from /root/reference behind
                  a C ABI (oracle/kaori_ref.cpp).  Only buildable where /root/reference exists;
                  the built .so travels to the GPU box.  ``KaoriRef.available()`` says whether
                  it is there.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
KAORI_SO = os.path.join(_HERE, "_ref", "libkaori_ref.so")

_ERRCAP = 1024


def _cstr_matrix(pools):
    """list of pools -> (const char* const* const*, int[] sizes, keepalive)"""
    keep = []
    rows = (C.POINTER(C.c_char_p) * max(len(pools), 1))()
    for r, p in enumerate(pools):
        arr, k = _cstr_array(p)
        keep.append((arr, k))
        rows[r] = C.cast(arr, C.POINTER(C.c_char_p))
    sizes = (C.c_int * max(len(pools), 1))(*[len(p) for p in pools])
    return rows, sizes, keep


class OracleError(RuntimeError):
    """The checker reported an error (the reference would have thrown std::runtime_error)."""


def _cstr_array(strings: Sequence[str | bytes]):
    arr = (C.c_char_p * max(len(strings), 1))()
    keep = []
    for i, s in enumerate(strings):
        b = s.encode() if isinstance(s, str) else bytes(s)
        keep.append(b)
        arr[i] = b
    return arr, keep


def pack_reads(reads: Iterable[str | bytes]):
    """list of reads -> (concatenated uint8 array, uint64 offsets[n+1])"""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    seqs = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    return seqs, offs


def write_fastq(path: str, reads: Iterable[str | bytes], gz: bool = False) -> None:
    import gzip
    opener = gzip.open if gz else open
    with opener(path, "wb") as f:
        for i, r in enumerate(reads):
            b = r.encode() if isinstance(r, str) else bytes(r)
            f.write(b"@r%d\n" % i + b + b"\n+\n" + b"I" * len(b) + b"\n")


def _as_batch(reads):
    if isinstance(reads, tuple) and len(reads) == 2 and isinstance(reads[0], np.ndarray):
        seqs, offs = reads
        return np.ascontiguousarray(seqs, dtype=np.uint8), np.ascontiguousarray(offs, dtype=np.uint64)
    return pack_reads(reads)


def _ptr(a: np.ndarray, ty):
    if a.size == 0:
        # keep a valid pointer for zero-length arrays
        a = np.zeros(1, dtype=a.dtype)
    return a.ctypes.data_as(C.POINTER(ty)), a


class Oracle:
    """C restatement (oracle/scg_oracle.c)."""

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing -- run `make -C oracle`")
        L = C.CDLL(path)
        self.L = L
        L.scgo_free.argtypes = [C.c_void_p]
        L.scgo_combo_rle.restype = C.c_int64

    @staticmethod
    def available() -> bool:
        return os.path.exists(ORACLE_SO)

    def count_single(self, reads, template: str, strand: int, pool: Sequence[str], mismatches: int, use_first: bool):
        seqs, offs = _as_batch(reads)
        n = len(offs) - 1
        counts = np.zeros(max(len(pool), 1), dtype=np.int32)
        total = C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        parr, _k = _cstr_array(pool)
        sp, _s = _ptr(seqs, C.c_char)
        rc = self.L.scgo_count_single(
            sp, offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(n),
            template.encode(), C.c_int(len(template)), C.c_int(strand),
            parr, C.c_int(len(pool)), C.c_int(mismatches), C.c_int(int(use_first)),
            counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(total), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        return counts[:len(pool)].copy(), int(total.value)

    def count_combo(self, reads, template: str, strand: int, pool0: Sequence[str], pool1: Sequence[str],
                    mismatches: int, use_first: bool):
        """-> (idx int32[2,K] 0-based sorted by (first, second), freq int32[K], total)"""
        seqs, offs = _as_batch(reads)
        n = len(offs) - 1
        tuples = np.zeros(2 * max(n, 1), dtype=np.int32)
        nt = C.c_int64(0)
        total = C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        p0, _k0 = _cstr_array(pool0)
        p1, _k1 = _cstr_array(pool1)
        sp, _s = _ptr(seqs, C.c_char)
        rc = self.L.scgo_count_combo(
            sp, offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(n),
            template.encode(), C.c_int(len(template)), C.c_int(strand),
            p0, C.c_int(len(pool0)), p1, C.c_int(len(pool1)),
            C.c_int(mismatches), C.c_int(int(use_first)),
            tuples.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nt), C.byref(total), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        freq = np.zeros(max(int(nt.value), 1), dtype=np.int32)
        k = self.L.scgo_combo_rle(tuples.ctypes.data_as(C.POINTER(C.c_int32)), nt,
                                  freq.ctypes.data_as(C.POINTER(C.c_int32)))
        idx = tuples[:2 * k].reshape(k, 2).T.copy()
        return idx, freq[:k].copy(), int(total.value)

    def count_dual(self, reads1, reads2, template1: str, reverse1: bool, mm1: int, pool1: Sequence[str],
                   template2: str, reverse2: bool, mm2: int, pool2: Sequence[str],
                   randomized: bool, use_first: bool):
        s1, o1 = _as_batch(reads1)
        s2, o2 = _as_batch(reads2)
        n = len(o1) - 1
        if len(o2) - 1 != n:
            raise OracleError("different number of reads in paired FASTQ files")
        if len(pool1) != len(pool2):
            raise OracleError("both barcode pools should be of the same length")
        counts = np.zeros(max(len(pool1), 1), dtype=np.int32)
        total = C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        p1, _k1 = _cstr_array(pool1)
        p2, _k2 = _cstr_array(pool2)
        sp1, _a = _ptr(s1, C.c_char)
        sp2, _b = _ptr(s2, C.c_char)
        rc = self.L.scgo_count_dual(
            sp1, o1.ctypes.data_as(C.POINTER(C.c_uint64)), sp2, o2.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(n),
            template1.encode(), C.c_int(len(template1)), C.c_int(int(reverse1)), C.c_int(mm1), p1,
            template2.encode(), C.c_int(len(template2)), C.c_int(int(reverse2)), C.c_int(mm2), p2,
            C.c_int(len(pool1)), C.c_int(int(randomized)), C.c_int(int(use_first)),
            counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(total), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        return counts[:len(pool1)].copy(), int(total.value)

    def count_dual_diag(self, reads1, reads2, template1: str, reverse1: bool, mm1: int, pool1: Sequence[str],
                        template2: str, reverse2: bool, mm2: int, pool2: Sequence[str],
                        randomized: bool, use_first: bool):
        """include.invalid=TRUE -> dict(counts, indices int32[2,K], freq, total, barcode1_only, barcode2_only)"""
        s1, o1 = _as_batch(reads1)
        s2, o2 = _as_batch(reads2)
        n = len(o1) - 1
        if len(o2) - 1 != n:
            raise OracleError("different number of reads in paired FASTQ files")
        if len(pool1) != len(pool2):
            raise OracleError("both barcode pools should be of the same length")
        counts = np.zeros(max(len(pool1), 1), dtype=np.int32)
        tuples = np.zeros(2 * max(n, 1), dtype=np.int32)
        nt = C.c_int64(0)
        total, b1, b2 = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        p1, _k1 = _cstr_array(pool1)
        p2, _k2 = _cstr_array(pool2)
        sp1, _a = _ptr(s1, C.c_char)
        sp2, _b = _ptr(s2, C.c_char)
        rc = self.L.scgo_count_dual_diag(
            sp1, o1.ctypes.data_as(C.POINTER(C.c_uint64)), sp2, o2.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(n),
            template1.encode(), C.c_int(len(template1)), C.c_int(int(reverse1)), C.c_int(mm1), p1,
            template2.encode(), C.c_int(len(template2)), C.c_int(int(reverse2)), C.c_int(mm2), p2,
            C.c_int(len(pool1)), C.c_int(int(randomized)), C.c_int(int(use_first)),
            counts.ctypes.data_as(C.POINTER(C.c_int32)), tuples.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nt),
            C.byref(total), C.byref(b1), C.byref(b2), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        freq = np.zeros(max(int(nt.value), 1), dtype=np.int32)
        k = self.L.scgo_combo_rle(tuples.ctypes.data_as(C.POINTER(C.c_int32)), nt, freq.ctypes.data_as(C.POINTER(C.c_int32)))
        return dict(counts=counts[:len(pool1)].copy(), indices=tuples[:2 * k].reshape(k, 2).T.copy(), freq=freq[:k].copy(),
                    total=int(total.value), barcode1_only=int(b1.value), barcode2_only=int(b2.value))

    def count_dual_single_end_diag(self, reads, template: str, strand: int, pools: Sequence[Sequence[str]], mismatches: int, use_first: bool):
        """countDualBarcodesSingleEnd(include.invalid=TRUE) -> dict(counts, indices int32[2,K], freq, total)"""
        s, o = _as_batch(reads)
        n = len(o) - 1
        nch = len(pools[0]) if pools else 0
        counts = np.zeros(max(nch, 1), dtype=np.int32)
        tuples = np.zeros(2 * max(n, 1), dtype=np.int32)
        nt = C.c_int64(0)
        total = C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        rows, sizes, _keep = _cstr_matrix(pools)
        sp, _a = _ptr(s, C.c_char)
        rc = self.L.scgo_count_dual_single_end_diag(sp, o.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(n),
                                                    template.encode(), C.c_int(len(template)), C.c_int(strand),
                                                    rows, sizes, C.c_int(len(pools)), C.c_int(mismatches), C.c_int(int(use_first)),
                                                    counts.ctypes.data_as(C.POINTER(C.c_int32)), tuples.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    C.byref(nt), C.byref(total), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        freq = np.zeros(max(int(nt.value), 1), dtype=np.int32)
        k = self.L.scgo_combo_rle(tuples.ctypes.data_as(C.POINTER(C.c_int32)), nt, freq.ctypes.data_as(C.POINTER(C.c_int32)))
        return dict(counts=counts[:nch].copy(), indices=tuples[:2 * k].reshape(k, 2).T.copy(), freq=freq[:k].copy(), total=int(total.value))

    def count_random(self, reads, template: str, strand: int, mismatches: int, use_first: bool):
        """countRandomBarcodes -> (dict sequence -> count, total).  The C restatement decides the window of
        every read (scgo_random_hits); the strings are cut here exactly as RandomBarcodeSingleEnd.hpp:86-115 does."""
        s, o = _as_batch(reads)
        n = len(o) - 1
        hits = np.zeros(max(n, 1), dtype=np.int32)
        vstart, vlen = C.c_int32(0), C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        sp, _a = _ptr(s, C.c_char)
        rc = self.L.scgo_random_hits(sp, o.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(n), template.encode(), C.c_int(len(template)),
                                     C.c_int(strand), C.c_int(mismatches), C.c_int(int(use_first)),
                                     hits.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(vstart), C.byref(vlen), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
        raw = bytes(s)
        out = {}
        for i in range(n):
            h = int(hits[i])
            if h < 0:
                continue
            a = int(o[i]) + (h >> 1) + vstart.value
            piece = raw[a:a + vlen.value].decode("latin-1")
            if h & 1:
                try:
                    piece = "".join(comp[c.upper()] for c in reversed(piece))
                except KeyError as ex:
                    raise OracleError(f"cannot complement unknown base '{ex.args[0]}'")
            out[piece] = out.get(piece, 0) + 1
        return out, n

    def count_dual_single_end(self, reads, template: str, strand: int, pools: Sequence[Sequence[str]], mismatches: int, use_first: bool):
        """countDualBarcodesSingleEnd -> (counts int32[n], total)"""
        s, o = _as_batch(reads)
        n = len(o) - 1
        nch = len(pools[0]) if pools else 0
        counts = np.zeros(max(nch, 1), dtype=np.int32)
        total = C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        rows, sizes, _keep = _cstr_matrix(pools)
        sp, _a = _ptr(s, C.c_char)
        rc = self.L.scgo_count_dual_single_end(sp, o.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(n),
                                               template.encode(), C.c_int(len(template)), C.c_int(strand),
                                               rows, sizes, C.c_int(len(pools)), C.c_int(mismatches), C.c_int(int(use_first)),
                                               counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(total), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        return counts[:nch].copy(), int(total.value)

    def count_combo_paired(self, reads1, reads2, template1: str, reverse1: bool, mm1: int, pool1: Sequence[str],
                           template2: str, reverse2: bool, mm2: int, pool2: Sequence[str],
                           randomized: bool, use_first: bool):
        """countPairedComboBarcodes -> dict(indices int32[2,K], freq, total, barcode1_only, barcode2_only)"""
        s1, o1 = _as_batch(reads1)
        s2, o2 = _as_batch(reads2)
        n = len(o1) - 1
        if len(o2) - 1 != n:
            raise OracleError("different number of reads in paired FASTQ files")
        tuples = np.zeros(2 * max(n, 1), dtype=np.int32)
        nt = C.c_int64(0)
        total, b1, b2 = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        p1, _k1 = _cstr_array(pool1)
        p2, _k2 = _cstr_array(pool2)
        sp1, _a = _ptr(s1, C.c_char)
        sp2, _b = _ptr(s2, C.c_char)
        rc = self.L.scgo_count_combo_paired(
            sp1, o1.ctypes.data_as(C.POINTER(C.c_uint64)), sp2, o2.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(n),
            template1.encode(), C.c_int(len(template1)), C.c_int(int(reverse1)), C.c_int(mm1), p1, C.c_int(len(pool1)),
            template2.encode(), C.c_int(len(template2)), C.c_int(int(reverse2)), C.c_int(mm2), p2, C.c_int(len(pool2)),
            C.c_int(int(randomized)), C.c_int(int(use_first)),
            tuples.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nt),
            C.byref(total), C.byref(b1), C.byref(b2), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        freq = np.zeros(max(int(nt.value), 1), dtype=np.int32)
        k = self.L.scgo_combo_rle(tuples.ctypes.data_as(C.POINTER(C.c_int32)), nt, freq.ctypes.data_as(C.POINTER(C.c_int32)))
        return dict(indices=tuples[:2 * k].reshape(k, 2).T.copy(), freq=freq[:k].copy(),
                    total=int(total.value), barcode1_only=int(b1.value), barcode2_only=int(b2.value))

    def match_barcodes(self, sequences: Sequence[str], choices: Sequence[str], substitutions: int = 0, reverse: bool = False):
        """-> (index int32[n] 0-based, -1 = NA; mismatches int32[n], -1 = NA)"""
        n = len(sequences)
        idx = np.zeros(max(n, 1), dtype=np.int32)
        mm = np.zeros(max(n, 1), dtype=np.int32)
        err = C.create_string_buffer(_ERRCAP)
        sa, _k = _cstr_array(sequences)
        ca, _k2 = _cstr_array(choices)
        rc = self.L.scgo_match_barcodes(sa, C.c_int(n), ca, C.c_int(len(choices)), C.c_int(substitutions), C.c_int(int(reverse)),
                                        idx.ctypes.data_as(C.POINTER(C.c_int32)), mm.ctypes.data_as(C.POINTER(C.c_int32)),
                                        err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        return idx[:n].copy(), mm[:n].copy()

    def parse_fastq(self, path: str):
        """-> (uint8 seqs, uint64 offsets[n+1])"""
        seqs_p = C.c_void_p()
        offs_p = C.c_void_p()
        n = C.c_int64(0)
        err = C.create_string_buffer(_ERRCAP)
        rc = self.L.scgo_parse_fastq(path.encode(), C.byref(seqs_p), C.byref(offs_p), C.byref(n), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        try:
            offs = np.ctypeslib.as_array(C.cast(offs_p, C.POINTER(C.c_uint64)), shape=(n.value + 1,)).copy()
            nb = int(offs[-1])
            seqs = np.ctypeslib.as_array(C.cast(seqs_p, C.POINTER(C.c_uint8)), shape=(max(nb, 1),))[:nb].copy()
        finally:
            self.L.scgo_free(seqs_p)
            self.L.scgo_free(offs_p)
        return seqs, offs


class KaoriRef:
    """The real reference (kaori v1.1.1 from /root/reference) behind oracle/kaori_ref.cpp."""

    def __init__(self, path: str = KAORI_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing -- run `make -C oracle` where /root/reference exists")
        self.L = C.CDLL(path)
        self.L.kref_free.argtypes = [C.c_void_p]

    @staticmethod
    def available() -> bool:
        return os.path.exists(KAORI_SO)

    def count_single(self, fastq: str, template: str, strand: int, pool: Sequence[str], mismatches: int,
                     use_first: bool, nthreads: int = 1):
        counts = np.zeros(max(len(pool), 1), dtype=np.int32)
        total = C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        parr, _k = _cstr_array(pool)
        rc = self.L.kref_count_single(fastq.encode(), template.encode(), C.c_int(strand), parr, C.c_int(len(pool)),
                                      C.c_int(mismatches), C.c_int(int(use_first)), C.c_int(nthreads),
                                      counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(total), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        return counts[:len(pool)].copy(), int(total.value)

    def count_combo(self, fastq: str, template: str, strand: int, pool0: Sequence[str], pool1: Sequence[str],
                    mismatches: int, use_first: bool, nthreads: int = 1):
        idx_p = C.POINTER(C.c_int32)()
        freq_p = C.POINTER(C.c_int32)()
        k = C.c_int64(0)
        total = C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        p0, _k0 = _cstr_array(pool0)
        p1, _k1 = _cstr_array(pool1)
        rc = self.L.kref_count_combo(fastq.encode(), template.encode(), C.c_int(strand), p0, C.c_int(len(pool0)),
                                     p1, C.c_int(len(pool1)), C.c_int(mismatches), C.c_int(int(use_first)), C.c_int(nthreads),
                                     C.byref(idx_p), C.byref(freq_p), C.byref(k), C.byref(total), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        K = int(k.value)
        try:
            idx = np.array([idx_p[i] for i in range(2 * K)], dtype=np.int32).reshape(K, 2).T.copy()
            freq = np.array([freq_p[i] for i in range(K)], dtype=np.int32)
        finally:
            self.L.kref_free(idx_p)
            self.L.kref_free(freq_p)
        return idx, freq, int(total.value)

    def count_dual(self, fastq1: str, template1: str, reverse1: bool, mm1: int, pool1: Sequence[str],
                   fastq2: str, template2: str, reverse2: bool, mm2: int, pool2: Sequence[str],
                   randomized: bool, use_first: bool, nthreads: int = 1):
        if len(pool1) != len(pool2):
            raise OracleError("both barcode pools should be of the same length")
        counts = np.zeros(max(len(pool1), 1), dtype=np.int32)
        total = C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        p1, _k1 = _cstr_array(pool1)
        p2, _k2 = _cstr_array(pool2)
        rc = self.L.kref_count_dual(fastq1.encode(), template1.encode(), C.c_int(int(reverse1)), C.c_int(mm1), p1,
                                    fastq2.encode(), template2.encode(), C.c_int(int(reverse2)), C.c_int(mm2), p2,
                                    C.c_int(len(pool1)), C.c_int(int(randomized)), C.c_int(int(use_first)), C.c_int(nthreads),
                                    counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(total), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        return counts[:len(pool1)].copy(), int(total.value)

    def count_dual_diag(self, fastq1: str, template1: str, reverse1: bool, mm1: int, pool1: Sequence[str],
                        fastq2: str, template2: str, reverse2: bool, mm2: int, pool2: Sequence[str],
                        randomized: bool, use_first: bool, nthreads: int = 1):
        if len(pool1) != len(pool2):
            raise OracleError("both barcode pools should be of the same length")
        counts = np.zeros(max(len(pool1), 1), dtype=np.int32)
        idx_p = C.POINTER(C.c_int32)()
        freq_p = C.POINTER(C.c_int32)()
        k = C.c_int64(0)
        total, b1, b2 = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        p1, _k1 = _cstr_array(pool1)
        p2, _k2 = _cstr_array(pool2)
        rc = self.L.kref_count_dual_diag(fastq1.encode(), template1.encode(), C.c_int(int(reverse1)), C.c_int(mm1), p1,
                                         fastq2.encode(), template2.encode(), C.c_int(int(reverse2)), C.c_int(mm2), p2,
                                         C.c_int(len(pool1)), C.c_int(int(randomized)), C.c_int(int(use_first)), C.c_int(nthreads),
                                         counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(idx_p), C.byref(freq_p), C.byref(k),
                                         C.byref(total), C.byref(b1), C.byref(b2), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        K = int(k.value)
        try:
            idx = np.array([idx_p[i] for i in range(2 * K)], dtype=np.int32).reshape(K, 2).T.copy()
            freq = np.array([freq_p[i] for i in range(K)], dtype=np.int32)
        finally:
            self.L.kref_free(idx_p)
            self.L.kref_free(freq_p)
        return dict(counts=counts[:len(pool1)].copy(), indices=idx, freq=freq, total=int(total.value),
                    barcode1_only=int(b1.value), barcode2_only=int(b2.value))

    def count_dual_single_end_diag(self, fastq: str, template: str, strand: int, pools: Sequence[Sequence[str]], mismatches: int,
                                   use_first: bool, nthreads: int = 1):
        nch = len(pools[0]) if pools else 0
        counts = np.zeros(max(nch, 1), dtype=np.int32)
        idx_p = C.POINTER(C.c_int32)()
        freq_p = C.POINTER(C.c_int32)()
        k, total = C.c_int64(0), C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        rows, sizes, _keep = _cstr_matrix(pools)
        rc = self.L.kref_count_dual_single_end_diag(fastq.encode(), template.encode(), C.c_int(strand), rows, sizes, C.c_int(len(pools)),
                                                    C.c_int(mismatches), C.c_int(int(use_first)), C.c_int(nthreads),
                                                    counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(idx_p), C.byref(freq_p), C.byref(k),
                                                    C.byref(total), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        K = int(k.value)
        try:
            idx = np.array([idx_p[i] for i in range(2 * K)], dtype=np.int32).reshape(K, 2).T.copy()
            freq = np.array([freq_p[i] for i in range(K)], dtype=np.int32)
        finally:
            self.L.kref_free(idx_p)
            self.L.kref_free(freq_p)
        return dict(counts=counts[:nch].copy(), indices=idx, freq=freq, total=int(total.value))

    def count_random(self, fastq: str, template: str, strand: int, mismatches: int, use_first: bool, nthreads: int = 1):
        seq_p = C.c_void_p()
        freq_p = C.POINTER(C.c_int32)()
        k, vlen, total = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        rc = self.L.kref_count_random(fastq.encode(), template.encode(), C.c_int(strand), C.c_int(mismatches), C.c_int(int(use_first)),
                                      C.c_int(nthreads), C.byref(seq_p), C.byref(freq_p), C.byref(k), C.byref(vlen), C.byref(total),
                                      err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        K, L = int(k.value), int(vlen.value)
        try:
            blob = C.string_at(seq_p, K * (L + 1)) if K else b""
            out = {blob[i * (L + 1): i * (L + 1) + L].decode("latin-1"): int(freq_p[i]) for i in range(K)}
        finally:
            self.L.kref_free(seq_p)
            self.L.kref_free(freq_p)
        return out, int(total.value)

    def count_dual_single_end(self, fastq: str, template: str, strand: int, pools: Sequence[Sequence[str]], mismatches: int,
                              use_first: bool, nthreads: int = 1):
        nch = len(pools[0]) if pools else 0
        counts = np.zeros(max(nch, 1), dtype=np.int32)
        total = C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        rows, sizes, _keep = _cstr_matrix(pools)
        rc = self.L.kref_count_dual_single_end(fastq.encode(), template.encode(), C.c_int(strand), rows, sizes, C.c_int(len(pools)),
                                               C.c_int(mismatches), C.c_int(int(use_first)), C.c_int(nthreads),
                                               counts.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(total), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        return counts[:nch].copy(), int(total.value)

    def count_combo_paired(self, fastq1: str, template1: str, reverse1: bool, mm1: int, pool1: Sequence[str],
                           fastq2: str, template2: str, reverse2: bool, mm2: int, pool2: Sequence[str],
                           randomized: bool, use_first: bool, nthreads: int = 1):
        idx_p = C.POINTER(C.c_int32)()
        freq_p = C.POINTER(C.c_int32)()
        k = C.c_int64(0)
        total, b1, b2 = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        err = C.create_string_buffer(_ERRCAP)
        p1, _k1 = _cstr_array(pool1)
        p2, _k2 = _cstr_array(pool2)
        rc = self.L.kref_count_combo_paired(fastq1.encode(), template1.encode(), C.c_int(int(reverse1)), C.c_int(mm1), p1, C.c_int(len(pool1)),
                                            fastq2.encode(), template2.encode(), C.c_int(int(reverse2)), C.c_int(mm2), p2, C.c_int(len(pool2)),
                                            C.c_int(int(randomized)), C.c_int(int(use_first)), C.c_int(nthreads),
                                            C.byref(idx_p), C.byref(freq_p), C.byref(k),
                                            C.byref(total), C.byref(b1), C.byref(b2), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        K = int(k.value)
        try:
            idx = np.array([idx_p[i] for i in range(2 * K)], dtype=np.int32).reshape(K, 2).T.copy()
            freq = np.array([freq_p[i] for i in range(K)], dtype=np.int32)
        finally:
            self.L.kref_free(idx_p)
            self.L.kref_free(freq_p)
        return dict(indices=idx, freq=freq, total=int(total.value), barcode1_only=int(b1.value), barcode2_only=int(b2.value))

    def match_barcodes(self, sequences: Sequence[str], choices: Sequence[str], substitutions: int = 0, reverse: bool = False):
        n = len(sequences)
        idx = np.zeros(max(n, 1), dtype=np.int32)
        mm = np.zeros(max(n, 1), dtype=np.int32)
        err = C.create_string_buffer(_ERRCAP)
        sa, _k = _cstr_array(sequences)
        ca, _k2 = _cstr_array(choices)
        rc = self.L.kref_match_barcodes(sa, C.c_int(n), ca, C.c_int(len(choices)), C.c_int(substitutions), C.c_int(int(reverse)),
                                        idx.ctypes.data_as(C.POINTER(C.c_int32)), mm.ctypes.data_as(C.POINTER(C.c_int32)),
                                        err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        return idx[:n].copy(), mm[:n].copy()

    def parse_fastq(self, path: str):
        seqs_p = C.c_void_p()
        offs_p = C.c_void_p()
        n = C.c_int64(0)
        err = C.create_string_buffer(_ERRCAP)
        rc = self.L.kref_parse_fastq(path.encode(), C.byref(seqs_p), C.byref(offs_p), C.byref(n), err, C.c_size_t(_ERRCAP))
        if rc:
            raise OracleError(err.value.decode())
        try:
            offs = np.ctypeslib.as_array(C.cast(offs_p, C.POINTER(C.c_uint64)), shape=(n.value + 1,)).copy()
            nb = int(offs[-1])
            seqs = np.ctypeslib.as_array(C.cast(seqs_p, C.POINTER(C.c_uint8)), shape=(max(nb, 1),))[:nb].copy()
        finally:
            self.L.kref_free(seqs_p)
            self.L.kref_free(offs_p)
        return seqs, offs
