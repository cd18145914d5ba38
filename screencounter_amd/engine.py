"""Plans over device-resident read batches (thin object layer over the C ABI of include/scg.h).

A ``Plan`` is the device counterpart of one kaori handler object
(``SingleBarcodeSingleEnd`` / ``CombinatorialBarcodesSingleEnd`` / ``DualBarcodesPairedEnd`` in
inst/include/kaori/handlers/ of the reference): construction validates the template and the
barcode pools exactly like the handler's constructor, ``count*`` is ``process()`` over a whole
batch, ``read`` is ``get_counts()`` / ``get_total()``.

torch is used only for device memory, streams and (in ``parallel``) torch.distributed.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import ScgError, check, cstr_array, errbuf

STRANDS = {"original": 0, "reverse": 1, "both": 2}


def _stream_handle(stream) -> int:
    """hipStream_t of a torch stream (None -> torch's current stream)."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream()
    return int(stream.cuda_stream)


def _dev_ptr(t) -> int:
    return 0 if t is None else int(t.data_ptr())


def _max_len(offsets, n: int) -> int:
    """Longest read of a ragged batch (one small device reduction + sync; callers that know the
    bound pass max_len themselves)."""
    import torch
    o = offsets[: n + 1].to(torch.int64) & 0xFFFFFFFF
    return int((o[1:] - o[:-1]).max().item()) if n > 0 else 0


def _check_u8(t, name):
    import torch
    if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous uint8 CUDA tensor")


def _check_offsets(t, name):
    import torch
    if t is None:
        return
    if t.dtype not in (torch.int32, torch.uint32) or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous int32/uint32 CUDA tensor of n_reads + 1 byte offsets")


class Plan:
    """A compiled (template, barcode library, options) bound to one GPU."""

    def __init__(self, handle: int, kind: str, n_pool: Sequence[int], device: int):
        self._lib = _lib.load()
        self._h = C.c_void_p(handle)
        self.kind = kind
        self.n_pool = tuple(n_pool)
        self.device = device
        self.num_counters = int(self._lib.scg_plan_num_counters(self._h))
        self._bound = None

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def single(cls, template: str, strand: int, pool: Sequence[str], mismatches: int = 0, use_first: bool = True,
               device: int = -1) -> "Plan":
        L = _lib.load()
        h = C.c_void_p()
        err = errbuf()
        parr, _keep = cstr_array(pool)
        check(L.scg_plan_single(C.byref(h), template.encode(), int(strand), parr, len(pool), int(mismatches),
                                int(bool(use_first)), int(device), err, _lib.ERRCAP), err)
        return cls(h.value, "single", (len(pool),), device)

    @classmethod
    def combo(cls, template: str, strand: int, pool0: Sequence[str], pool1: Sequence[str], mismatches: int = 0,
              use_first: bool = True, device: int = -1) -> "Plan":
        L = _lib.load()
        h = C.c_void_p()
        err = errbuf()
        p0, _k0 = cstr_array(pool0)
        p1, _k1 = cstr_array(pool1)
        check(L.scg_plan_combo(C.byref(h), template.encode(), int(strand), p0, len(pool0), p1, len(pool1), int(mismatches),
                               int(bool(use_first)), int(device), err, _lib.ERRCAP), err)
        return cls(h.value, "combo", (len(pool0), len(pool1)), device)

    @classmethod
    def dual(cls, template1: str, reverse1: bool, mismatches1: int, pool1: Sequence[str],
             template2: str, reverse2: bool, mismatches2: int, pool2: Sequence[str],
             randomized: bool = False, use_first: bool = True, device: int = -1, diagnostics: bool = False) -> "Plan":
        """diagnostics=True builds the include.invalid=TRUE variant (read with read_diagnostics())."""
        if len(pool1) != len(pool2):
            # kaori/handlers/DualBarcodesPairedEnd.hpp:106-109
            raise ScgError(_lib.SCG_ERR_INVALID, "both barcode pools should be of the same length")
        L = _lib.load()
        h = C.c_void_p()
        err = errbuf()
        p1, _k1 = cstr_array(pool1)
        p2, _k2 = cstr_array(pool2)
        check(L.scg_plan_dual(C.byref(h), template1.encode(), int(bool(reverse1)), int(mismatches1), p1,
                              template2.encode(), int(bool(reverse2)), int(mismatches2), p2, len(pool1),
                              int(bool(randomized)), int(bool(use_first)), int(bool(diagnostics)), int(device), err, _lib.ERRCAP), err)
        return cls(h.value, "dual", (len(pool1),), device)

    @classmethod
    def dual_single_end(cls, template: str, strand: int, pools: Sequence[Sequence[str]], mismatches: int = 0,
                        use_first: bool = True, device: int = -1) -> "Plan":
        """countDualBarcodesSingleEnd: counted with count(), read with read()."""
        L = _lib.load()
        h = C.c_void_p()
        err = errbuf()
        rows, sizes, _keep = _lib.cstr_matrix(pools)
        check(L.scg_plan_dual_single_end(C.byref(h), template.encode(), int(strand), rows, sizes, len(pools),
                                         int(mismatches), int(bool(use_first)), int(device), err, _lib.ERRCAP), err)
        return cls(h.value, "single", (len(pools[0]) if pools else 0,), device)

    @classmethod
    def paired_combo(cls, template1: str, reverse1: bool, mismatches1: int, pool1: Sequence[str],
                     template2: str, reverse2: bool, mismatches2: int, pool2: Sequence[str],
                     randomized: bool = False, use_first: bool = True, device: int = -1) -> "Plan":
        """countPairedComboBarcodes: counted with count_paired(), read with read_diagnostics()
        (indices/freq = the combinations; `counts` is empty)."""
        L = _lib.load()
        h = C.c_void_p()
        err = errbuf()
        p1, _k1 = cstr_array(pool1)
        p2, _k2 = cstr_array(pool2)
        check(L.scg_plan_paired_combo(C.byref(h), template1.encode(), int(bool(reverse1)), int(mismatches1), p1, len(pool1),
                                      template2.encode(), int(bool(reverse2)), int(mismatches2), p2, len(pool2),
                                      int(bool(randomized)), int(bool(use_first)), int(device), err, _lib.ERRCAP), err)
        return cls(h.value, "dual", (0,), device)

    def close(self) -> None:
        if self._h:
            self._lib.scg_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- counters -----------------------------------------------------------------------------
    def bind_counters(self, tensor) -> None:
        """Accumulate into a caller-owned int32 CUDA tensor (e.g. the buffer handed to the RCCL
        all-reduce) instead of the plan's own counters.  ``None`` restores the plan's buffer."""
        import torch
        err = errbuf()
        if tensor is None:
            check(self._lib.scg_plan_bind_counters(self._h, None, err, _lib.ERRCAP), err)
            self._bound = None
            return
        if tensor.dtype != torch.int32 or not tensor.is_cuda or not tensor.is_contiguous() or tensor.numel() < self.num_counters:
            raise ValueError(f"counters must be a contiguous int32 CUDA tensor with >= {self.num_counters} elements")
        check(self._lib.scg_plan_bind_counters(self._h, C.c_void_p(tensor.data_ptr()), err, _lib.ERRCAP), err)
        self._bound = tensor

    def reset(self, stream=None) -> None:
        err = errbuf()
        check(self._lib.scg_plan_reset(self._h, C.c_void_p(_stream_handle(stream)), err, _lib.ERRCAP), err)

    # ---- the hot path ---------------------------------------------------------------------------
    def count(self, seqs, offsets=None, fixed_len: int = 0, n_reads: Optional[int] = None, stream=None, max_len: int = 0) -> None:
        """One step: count a batch of single-end reads resident in HBM (asynchronous).
        max_len (ragged batches): upper bound on the read lengths if known; a tile-shape hint only."""
        _check_u8(seqs, "seqs")
        _check_offsets(offsets, "offsets")
        if offsets is not None:
            n = offsets.numel() - 1 if n_reads is None else n_reads
            if max_len <= 0 and n > 0:
                max_len = _max_len(offsets, n)
        else:
            if fixed_len <= 0:
                raise ValueError("either offsets or a positive fixed_len is required")
            n = seqs.numel() // fixed_len if n_reads is None else n_reads
            if n * fixed_len > seqs.numel():
                raise ValueError("seqs is shorter than n_reads * fixed_len")
        err = errbuf()
        check(self._lib.scg_count_batch(self._h, C.c_void_p(_dev_ptr(seqs)), C.c_void_p(_dev_ptr(offsets)), int(fixed_len),
                                        int(max_len), int(n), C.c_void_p(_stream_handle(stream)), err, _lib.ERRCAP), err)

    def count_paired(self, seqs1, seqs2, offsets1=None, offsets2=None, fixed_len1: int = 0, fixed_len2: int = 0,
                     n_pairs: Optional[int] = None, stream=None, max_len: int = 0) -> None:
        _check_u8(seqs1, "seqs1")
        _check_u8(seqs2, "seqs2")
        _check_offsets(offsets1, "offsets1")
        _check_offsets(offsets2, "offsets2")

        def count_of(seqs, offs, fl):
            if offs is not None:
                return offs.numel() - 1
            if fl <= 0:
                raise ValueError("either offsets or a positive fixed_len is required")
            return seqs.numel() // fl
        n1, n2 = count_of(seqs1, offsets1, fixed_len1), count_of(seqs2, offsets2, fixed_len2)
        if n_pairs is None:
            if n1 != n2:
                # kaori/process_data.hpp:284-285
                raise ScgError(_lib.SCG_ERR_IO, "different number of reads in paired FASTQ files")
            n_pairs = n1
        elif n_pairs > min(n1, n2):
            raise ValueError("n_pairs exceeds the batch")
        if max_len <= 0 and n_pairs > 0:
            m1 = fixed_len1 if offsets1 is None else _max_len(offsets1, n_pairs)
            m2 = fixed_len2 if offsets2 is None else _max_len(offsets2, n_pairs)
            max_len = max(m1, m2)
        err = errbuf()
        check(self._lib.scg_count_batch_paired(self._h, C.c_void_p(_dev_ptr(seqs1)), C.c_void_p(_dev_ptr(offsets1)), int(fixed_len1),
                                               C.c_void_p(_dev_ptr(seqs2)), C.c_void_p(_dev_ptr(offsets2)), int(fixed_len2),
                                               int(max_len), int(n_pairs), C.c_void_p(_stream_handle(stream)), err, _lib.ERRCAP), err)

    def read(self, stream=None):
        """Synchronise and fetch (counts int32[num_counters], total reads seen)."""
        counts = np.zeros(max(self.num_counters, 1), dtype=np.int32)
        total = C.c_int64(0)
        err = errbuf()
        check(self._lib.scg_plan_read(self._h, counts.ctypes.data_as(_lib.i32_p), C.byref(total),
                                      C.c_void_p(_stream_handle(stream)), err, _lib.ERRCAP), err)
        return counts[:self.num_counters], int(total.value)

    def read_diagnostics(self, stream=None):
        """Dual plans built with diagnostics=True -> dict(counts, indices int32[2, K], freq, total,
        barcode1_only, barcode2_only), the outputs of src/count_dual_barcodes.cpp:64-70."""
        counts = np.zeros(max(self.n_pool[0], 1), dtype=np.int32)
        idx_p, freq_p = _lib.i32_p(), _lib.i32_p()
        k, total = C.c_int64(0), C.c_int64(0)
        b1, b2 = C.c_int32(0), C.c_int32(0)
        err = errbuf()
        check(self._lib.scg_plan_read_diagnostics(self._h, counts.ctypes.data_as(_lib.i32_p), C.byref(idx_p), C.byref(freq_p),
                                                  C.byref(k), C.byref(total), C.byref(b1), C.byref(b2),
                                                  C.c_void_p(_stream_handle(stream)), err, _lib.ERRCAP), err)
        K = int(k.value)
        try:
            idx = np.ctypeslib.as_array(idx_p, shape=(max(2 * K, 1),))[:2 * K].reshape(K, 2).T.copy()
            freq = np.ctypeslib.as_array(freq_p, shape=(max(K, 1),))[:K].copy()
        finally:
            self._lib.scg_free(idx_p)
            self._lib.scg_free(freq_p)
        return dict(counts=counts[:self.n_pool[0]].copy(), indices=idx.astype(np.int32), freq=freq.astype(np.int32),
                    total=int(total.value), barcode1_only=int(b1.value), barcode2_only=int(b2.value))

    def read_combo(self, stream=None):
        """Combo plans: (indices int32[2, K] sorted by (first, second), freq int32[K], total) -- from the dense histogram or,
        for combination spaces beyond 2^26 cells, from the sorted, run-length encoded combination streams."""
        if self.kind != "combo":
            raise ValueError("read_combo needs a combo plan")
        idx_p, freq_p = _lib.i32_p(), _lib.i32_p()
        k, total = C.c_int64(0), C.c_int64(0)
        err = errbuf()
        check(self._lib.scg_plan_read_combinations(self._h, C.byref(idx_p), C.byref(freq_p), C.byref(k), C.byref(total),
                                                   C.c_void_p(_stream_handle(stream)), err, _lib.ERRCAP), err)
        K = int(k.value)
        try:
            idx = np.ctypeslib.as_array(idx_p, shape=(max(2 * K, 1),))[:2 * K].reshape(K, 2).T.copy()
            freq = np.ctypeslib.as_array(freq_p, shape=(max(K, 1),))[:K].copy()
        finally:
            self._lib.scg_free(idx_p)
            self._lib.scg_free(freq_p)
        return idx.astype(np.int32), freq.astype(np.int32), int(total.value)

    # ---- measurement ----------------------------------------------------------------------------
    def set_profiling(self, enabled: bool) -> None:
        self._lib.scg_plan_set_profiling(self._h, int(bool(enabled)))

    def kernel_stats(self):
        """(total kernel milliseconds, launches) since the last reset, from HIP events recorded on
        the launch stream around every counting kernel."""
        ms = C.c_double(0)
        n = C.c_int64(0)
        err = errbuf()
        check(self._lib.scg_plan_kernel_stats(self._h, C.byref(ms), C.byref(n), err, _lib.ERRCAP), err)
        return float(ms.value), int(n.value)


def combo_compact(cells: np.ndarray, n0: int, n1: int):
    """Dense histogram -> the reference's sorted run-length form (src/utils.h:14-45)."""
    L = _lib.load()
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    if cells.size < n0 * n1:
        raise ValueError("cells too short")
    idx_p = _lib.i32_p()
    freq_p = _lib.i32_p()
    k = C.c_int64(0)
    err = errbuf()
    src = cells if cells.size else np.zeros(1, dtype=np.int32)
    check(L.scg_combo_compact(src.ctypes.data_as(_lib.i32_p), int(n0), int(n1), C.byref(idx_p), C.byref(freq_p), C.byref(k),
                              err, _lib.ERRCAP), err)
    K = int(k.value)
    try:
        idx = np.ctypeslib.as_array(idx_p, shape=(max(2 * K, 1),))[:2 * K].reshape(K, 2).T.copy()
        freq = np.ctypeslib.as_array(freq_p, shape=(max(K, 1),))[:K].copy()
    finally:
        L.scg_free(idx_p)
        L.scg_free(freq_p)
    return idx.astype(np.int32), freq.astype(np.int32)


def upload_reads(reads, device="cuda"):
    """list of reads (str/bytes) or (uint8 array, offsets array) -> (seqs uint8 cuda, offsets int32 cuda)."""
    import torch
    if isinstance(reads, tuple):
        seqs, offs = reads
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
    else:
        bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        offs = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        seqs = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    if offs[-1] >= (1 << 32):
        raise ValueError("a batch must stay below 4 GiB of sequence bytes")
    o32 = offs.astype(np.uint32).view(np.int32)
    # keep at least one byte so that data_ptr() is valid for an empty batch
    s = torch.from_numpy(seqs if seqs.size else np.zeros(1, dtype=np.uint8)).to(device)
    o = torch.from_numpy(o32.copy()).to(device)
    return s, o
