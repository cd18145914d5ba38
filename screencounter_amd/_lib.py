"""ctypes binding of libscg.so (include/scg.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (or ``make -C
screencounter_amd/csrc``).  There is no fallback of any kind: if the library is missing the
import fails, and if no HIP device is present every counting call returns SCG_ERR_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCG_LIB") or os.path.join(_HERE, "libscg.so")     # SCG_LIB: A/B builds (tools/ab.sh)

SCG_OK, SCG_ERR_INVALID, SCG_ERR_IO, SCG_ERR_DEVICE, SCG_ERR_UNSUPPORTED = 0, 1, 2, 3, 4
ERRCAP = 1024

c_str_p = C.POINTER(C.c_char_p)
i32_p = C.POINTER(C.c_int32)
i64_p = C.POINTER(C.c_int64)


class SynthSpec(C.Structure):
    """struct scg_synth_spec (include/scg.h)"""
    _fields_ = [
        ("seed", C.c_uint64),
        ("first_read", C.c_int64),
        ("read_len", C.c_int32),
        ("template_len", C.c_int32),
        ("d_template", C.c_void_p),
        ("n_regions", C.c_int32),
        ("region_start", C.c_int32 * 2),
        ("region_len", C.c_int32 * 2),
        ("d_pool", C.c_void_p * 2),
        ("n_pool", C.c_int32 * 2),
        ("d_pair_index", C.c_void_p),
        ("n_pairs", C.c_int32),
        ("pair_column", C.c_int32),
        ("p_invalid_pair", C.c_float),
        ("p_sub", C.c_float),
        ("p_n", C.c_float),
        ("p_junk", C.c_float),
        ("p_reverse", C.c_float),
    ]


# name -> (restype, argtypes); must list every symbol declared in include/scg.h
SIGNATURES = {
    "scg_version": (C.c_char_p, []),
    "scg_device_count": (C.c_int, []),
    "scg_set_device": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "scg_set_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_size_t]),
    "scg_count_single_barcodes": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, c_str_p, C.c_int32, C.c_int, C.c_int, C.c_int,
                                            i32_p, i32_p, C.c_char_p, C.c_size_t]),
    "scg_count_combo_barcodes_single": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, c_str_p, C.c_int32, c_str_p, C.c_int32,
                                                  C.c_int, C.c_int, C.c_int, C.POINTER(i32_p), C.POINTER(i32_p), i64_p, i32_p,
                                                  C.c_char_p, C.c_size_t]),
    "scg_count_dual_barcodes": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, c_str_p,
                                          C.c_char_p, C.c_char_p, C.c_int, C.c_int, c_str_p, C.c_int32,
                                          C.c_int, C.c_int, C.c_int, C.c_int, i32_p, i32_p, C.c_char_p, C.c_size_t]),
    "scg_match_barcodes": (C.c_int, [c_str_p, C.c_int32, c_str_p, C.c_int32, C.c_int, C.c_int, i32_p, i32_p, C.c_char_p, C.c_size_t]),
    "scg_count_single_barcodes_files": (C.c_int, [c_str_p, C.c_int32, C.c_char_p, C.c_int, c_str_p, C.c_int32, C.c_int, C.c_int, C.c_int,
                                                  i32_p, i32_p, C.c_char_p, C.c_size_t]),
    "scg_count_combo_barcodes_single_files": (C.c_int, [c_str_p, C.c_int32, C.c_char_p, C.c_int, c_str_p, C.c_int32, c_str_p, C.c_int32,
                                                        C.c_int, C.c_int, C.c_int, C.POINTER(i32_p), C.POINTER(i32_p), i64_p, i32_p,
                                                        C.c_char_p, C.c_size_t]),
    "scg_count_dual_barcodes_files": (C.c_int, [c_str_p, C.c_char_p, C.c_int, C.c_int, c_str_p,
                                                c_str_p, C.c_char_p, C.c_int, C.c_int, c_str_p, C.c_int32, C.c_int32,
                                                C.c_int, C.c_int, C.c_int, i32_p, i32_p, C.c_char_p, C.c_size_t]),
    "scg_fastq_text_windows": (C.c_int, [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p), i64_p, C.POINTER(C.c_void_p), i64_p,
                                         C.c_char_p, C.c_char_p, C.c_size_t]),
    "scg_fastq_scan_windows": (C.c_int, [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), i64_p, i64_p,
                                         C.c_char_p, C.c_size_t]),
    "scg_bgzf_member_batches": (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p), i64_p, C.POINTER(C.c_void_p), i64_p, i64_p,
                                          C.c_char_p, C.c_size_t]),
    "scg_free": (None, [C.c_void_p]),
    "scg_release_buffers": (None, []),
    "scg_parse_fastq": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), i64_p, C.c_char_p, C.c_size_t]),
    "scg_plan_single": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, c_str_p, C.c_int32, C.c_int, C.c_int, C.c_int,
                                  C.c_char_p, C.c_size_t]),
    "scg_plan_combo": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, c_str_p, C.c_int32, c_str_p, C.c_int32, C.c_int, C.c_int,
                                 C.c_int, C.c_char_p, C.c_size_t]),
    "scg_plan_dual": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int, c_str_p, C.c_char_p, C.c_int, C.c_int, c_str_p,
                                C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "scg_count_dual_barcodes_diagnostics": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, c_str_p,
                                                      C.c_char_p, C.c_char_p, C.c_int, C.c_int, c_str_p, C.c_int32,
                                                      C.c_int, C.c_int, C.c_int, i32_p, C.POINTER(i32_p), C.POINTER(i32_p), i64_p,
                                                      i32_p, i32_p, i32_p, C.c_char_p, C.c_size_t]),
    "scg_count_random_barcodes": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                            C.POINTER(i32_p), i64_p, i32_p, i32_p, C.c_char_p, C.c_size_t]),
    "scg_count_dual_barcodes_single_end": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(c_str_p), i32_p, C.c_int32,
                                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32_p, i32_p, C.c_char_p, C.c_size_t]),
    "scg_count_dual_barcodes_single_end_diagnostics": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(c_str_p), i32_p, C.c_int32,
                                                                 C.c_int, C.c_int, C.c_int, C.c_int, i32_p, C.POINTER(i32_p), C.POINTER(i32_p),
                                                                 i64_p, i32_p, C.c_char_p, C.c_size_t]),
    "scg_plan_dual_single_end": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.POINTER(c_str_p), i32_p, C.c_int32,
                                           C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "scg_count_combo_barcodes_paired": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, c_str_p, C.c_int32,
                                                  C.c_char_p, C.c_char_p, C.c_int, C.c_int, c_str_p, C.c_int32,
                                                  C.c_int, C.c_int, C.c_int, C.POINTER(i32_p), C.POINTER(i32_p), i64_p,
                                                  i32_p, i32_p, i32_p, C.c_char_p, C.c_size_t]),
    "scg_plan_paired_combo": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int, c_str_p, C.c_int32,
                                        C.c_char_p, C.c_int, C.c_int, c_str_p, C.c_int32,
                                        C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "scg_plan_read_diagnostics": (C.c_int, [C.c_void_p, i32_p, C.POINTER(i32_p), C.POINTER(i32_p), i64_p, i64_p, i32_p, i32_p,
                                            C.c_void_p, C.c_char_p, C.c_size_t]),
    "scg_plan_destroy": (None, [C.c_void_p]),
    "scg_plan_num_counters": (C.c_int64, [C.c_void_p]),
    "scg_plan_device_counters": (C.c_void_p, [C.c_void_p]),
    "scg_plan_bind_counters": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]),
    "scg_plan_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]),
    "scg_count_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_char_p, C.c_size_t]),
    "scg_count_batch_paired": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_int64, C.c_void_p, C.c_char_p, C.c_size_t]),
    "scg_plan_read": (C.c_int, [C.c_void_p, i32_p, i64_p, C.c_void_p, C.c_char_p, C.c_size_t]),
    "scg_plan_read_combinations": (C.c_int, [C.c_void_p, C.POINTER(i32_p), C.POINTER(i32_p), i64_p, i64_p, C.c_void_p, C.c_char_p, C.c_size_t]),
    "scg_combo_compact": (C.c_int, [i32_p, C.c_int32, C.c_int32, C.POINTER(i32_p), C.POINTER(i32_p), i64_p, C.c_char_p, C.c_size_t]),
    "scg_plan_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "scg_plan_kernel_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), i64_p, C.c_char_p, C.c_size_t]),
    "scg_synth_reads": (C.c_int, [C.POINTER(SynthSpec), C.c_void_p, C.c_int64, C.c_void_p, C.c_char_p, C.c_size_t]),
}

_lib = None


def load() -> C.CDLL:
    """Load libscg.so once.  torch is imported first so that both share one HIP runtime
    (torch ships its own libamdhip64.so.7; whichever is loaded first serves both)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C screencounter_amd/csrc`.  screencounter_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads the HIP runtime torch tensors live in)
    except Exception:  # pragma: no cover - torch is plumbing only; the library works without it
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("SCG_LIB") and not hasattr(lib, name):
            continue              # an older build under tools/ab/ (A/B timing of entry points both builds have)
        fn = getattr(lib, name)   # AttributeError here means the .so is stale: rebuild
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class ScgError(RuntimeError):
    """Non-zero return from libscg; `.code` is one of the SCG_ERR_* values.  SCG_ERR_INVALID /
    SCG_ERR_IO correspond to the std::runtime_error the reference throws (an R error)."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def check(rc: int, err) -> None:
    if rc != SCG_OK:
        raise ScgError(rc, err.value.decode(errors="replace"))


def errbuf():
    return C.create_string_buffer(ERRCAP)


def cstr_matrix(pools):
    """list of pools -> (const char* const* const*, int32[] sizes, keepalive) for scg_*_dual_single_end"""
    keep = []
    rows = (c_str_p * max(len(pools), 1))()
    for r, p in enumerate(pools):
        arr, k = cstr_array(p)
        keep.append((arr, k))
        rows[r] = C.cast(arr, c_str_p)
    sizes = (C.c_int32 * max(len(pools), 1))(*[len(p) for p in pools])
    return rows, sizes, keep


class PreparedPool(list):
    """A barcode pool whose C string array has been built once (`prepare_pool`): passing it to the count_* functions
    skips the per-call marshalling of this Python mirror (~0.4 us per barcode; R hands the C side its CHARSXP pointers
    without such a copy)."""
    _carr = None
    _keep = None


def prepare_pool(strings) -> PreparedPool:
    p = PreparedPool(strings)
    p._carr, p._keep = cstr_array(list(strings))
    return p


def cstr_array(strings):
    if isinstance(strings, PreparedPool) and strings._carr is not None:
        return strings._carr, strings._keep
    arr = (C.c_char_p * max(len(strings), 1))()
    keep = []
    for i, s in enumerate(strings):
        b = s.encode() if isinstance(s, str) else bytes(s)
        keep.append(b)
        arr[i] = b
    return arr, keep
