"""R-level mirror of the reference's user functions, for the TESTS only (not part of the product package).

``countSingleBarcodes`` (R/countSingleBarcodes.R:82-105), ``countComboBarcodes`` (R/countComboBarcodes.R:87-124),
``countDualBarcodes`` (R/countDualBarcodes.R:118-160), ``matchBarcodes`` (R/matchBarcodes.R) and their siblings:
template construction from flanks, ``N`` -> ``-``, strand names, 1-based indices, the ``matrixOf*`` multi-file
wrappers.  R's DataFrame / SummarizedExperiment become plain dataclasses with the same column names.  The reference
keeps this layer as it is (R/*.R stays); it is restated here so that the reference's
test vectors (tests/testthat/*.R) can be transcribed literally.  Everything below calls the Rcpp-level mirror of
screencounter_amd.api, i.e. the C ABI.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from screencounter_amd import _lib
from screencounter_amd._lib import ScgError, check, errbuf
from screencounter_amd.api import (  # noqa: F401
    count_single_barcodes, count_combo_barcodes_single, count_dual_barcodes, count_combo_barcodes_paired,
    count_single_barcodes_files, count_combo_barcodes_single_files, count_dual_barcodes_files,
    count_dual_barcodes_single_end, count_random_barcodes, match_barcodes,
)

_STRAND = {"original": 0, "reverse": 1, "both": 2}


# =============================================================================================
# R level
# =============================================================================================
@dataclass
class BarcodeCounts:
    """countSingleBarcodes(): DataFrame(choices, counts) + metadata$nreads."""
    choices: List[str]
    counts: np.ndarray
    nreads: int


@dataclass
class ComboCounts:
    """countComboBarcodes(): DataFrame(combinations = DataFrame(first, second), counts) + metadata$nreads.
    `first`/`second` hold 1-based indices (indices=True) or the barcode sequences."""
    names: List[str]
    combinations: Dict[str, list]
    counts: np.ndarray
    nreads: int
    # countPairedComboBarcodes metadata (R/countPairedComboBarcodes.R:110)
    npairs: Optional[int] = None
    barcode1_only: Optional[int] = None
    barcode2_only: Optional[int] = None


@dataclass
class DualCounts:
    """countDualBarcodes(): `choices` with a counts column + metadata$npairs.  With include.invalid=TRUE the
    invalid combinations are appended as extra rows (`valid` False) and the metadata gains barcode1.only,
    barcode2.only, invalid.pair (R/countDualBarcodes.R:152-159)."""
    choices: Dict[str, List[str]]
    counts: np.ndarray
    npairs: int
    valid: Optional[List[bool]] = None
    barcode1_only: Optional[int] = None
    barcode2_only: Optional[int] = None
    invalid_pair: Optional[int] = None


@dataclass
class CountMatrix:
    """matrixOf*(): SummarizedExperiment(counts; rowData; colData)."""
    counts: np.ndarray                 # rows x files
    row_data: Dict[str, list]
    col_data: Dict[str, list]
    rownames: Optional[List[str]] = None
    colnames: Optional[List[str]] = None
    metadata: dict = field(default_factory=dict)


def _strand_code(strand: str) -> int:
    if strand not in _STRAND:
        raise ValueError("'strand' should be one of 'both', 'original', 'reverse'")
    return _STRAND[strand]


def countSingleBarcodes(fastq: str, choices: Sequence[str], flank5: str = "", flank3: str = "", template: Optional[str] = None,
                        substitutions: int = 0, find_best: bool = False, strand: str = "both", num_threads: int = 1) -> BarcodeCounts:
    """R/countSingleBarcodes.R:82-105."""
    choices = list(choices)
    if template is not None:
        template = template.replace("N", "-")                       # :93
    else:
        template = flank5 + "-" * (len(choices[0]) if choices else 0) + flank3   # :95-97
    counts, total = count_single_barcodes(fastq, template, _strand_code(strand), choices, substitutions, not find_best, num_threads)
    return BarcodeCounts(choices=choices, counts=counts, nreads=total)


def _map_files(fn, files, devices=None, jobs_per_device: int = 1):
    """The matrixOf* schedulers (R/countSingleBarcodes.R:117 `bplapply(files, ...)` and siblings): file i is
    counted on GPU devices[i % len(devices)] by a worker thread of this process (the C ABI releases the GIL and
    every call owns its plan, so the calls are independent).  devices=None uses every visible GPU;
    jobs_per_device > 1 keeps that many files in flight on each GPU, which is what gzip input wants: one file is
    bounded by its single inflate thread (~1.6 Mreads/s) while the device has four orders of magnitude to spare.
    Results come back in file order."""
    files = list(files)
    L = _lib.load()
    if devices is None:
        devices = list(range(max(int(L.scg_device_count()), 1)))
    devices = [d for d in devices for _ in range(max(int(jobs_per_device), 1))]
    if len(files) <= 1 or len(devices) <= 1:
        if files and devices and int(L.scg_device_count()) > 0:
            err = errbuf()
            check(L.scg_set_device(int(devices[0]), err, _lib.ERRCAP), err)
        return [fn(f) for f in files]
    import queue
    from concurrent.futures import ThreadPoolExecutor
    free = queue.Queue()
    for d in devices:
        free.put(d)

    def job(f):
        d = free.get()                      # one file at a time per GPU
        try:
            err = errbuf()
            check(L.scg_set_device(int(d), err, _lib.ERRCAP), err)
            return fn(f)
        finally:
            free.put(d)

    with ThreadPoolExecutor(max_workers=len(devices)) as pool:
        return list(pool.map(job, files))


def matrixOfSingleBarcodes(files: Sequence[str], choices: Sequence[str], withDimnames: bool = True, devices=None, jobs_per_device: int = 1,
                           flank5: str = "", flank3: str = "", template: Optional[str] = None, substitutions: int = 0, find_best: bool = False,
                           strand: str = "both", num_threads: int = 1) -> CountMatrix:
    """R/countSingleBarcodes.R:112-126.  The files are scheduled over the GPUs INSIDE one native call
    (scg_count_single_barcodes_files: library compiled once, one file at a time per device, results in file order)
    instead of BiocParallel worker processes; `devices` / `jobs_per_device` become its device list."""
    files = list(files)
    choices = list(choices)
    if template is not None:
        template = template.replace("N", "-")
    else:
        template = flank5 + "-" * (len(choices[0]) if choices else 0) + flank3
    if devices is not None:
        devices = [d for d in devices for _ in range(max(int(jobs_per_device), 1))]
    elif jobs_per_device > 1:
        devices = [d for d in range(max(int(_lib.load().scg_device_count()), 1)) for _ in range(int(jobs_per_device))]
    mat, totals = count_single_barcodes_files(files, template, _strand_code(strand), choices, substitutions, not find_best, num_threads, devices)
    se = CountMatrix(counts=mat, row_data={"choices": list(choices)},
                     col_data={"paths": list(files), "nreads": totals, "nmapped": mat.sum(axis=0).astype(np.int64).tolist()})
    if withDimnames:
        se.rownames = list(choices)
        se.colnames = [os.path.basename(f) for f in files]
    return se


def parseBarcodeTemplate(template: str):
    """R/parseBarcodeTemplate.R:29-44: positions (1-based) and lengths of the N runs, and the constant pieces."""
    pos, lens = [], []
    for m in re.finditer(r"N+", template):
        pos.append(m.start() + 1)
        lens.append(m.end() - m.start())
    constants = re.split(r"N+", template)
    return {"variable": {"pos": pos, "len": lens}, "constant": constants}


def _combo_setup(template: str, choices, strand: str):
    """Argument handling of R/countComboBarcodes.R:87-118 -> (names, pools, native template, strand code)."""
    if isinstance(choices, dict):
        names = list(choices.keys())
        pools = [list(v) for v in choices.values()]
    else:
        names = ["first", "second"]
        pools = [list(v) for v in choices]
    parsed = parseBarcodeTemplate(template)
    n_len = parsed["variable"]["len"]
    nvariables = len(n_len)
    if nvariables != 2:                                             # :105-107
        raise ScgError(_lib.SCG_ERR_INVALID, f"'length(choices)={nvariables}' is not currently supported")
    if nvariables != len(pools):                                    # :108-110
        raise ScgError(_lib.SCG_ERR_INVALID, "'length(choices)' is not equal to the number of variable regions in 'template'")
    for i in range(nvariables):                                     # :111-115
        if not all(len(s) == n_len[i] for s in pools[i]):
            raise ScgError(_lib.SCG_ERR_INVALID, "each column of 'choices' must have same width as variable region in 'template'")
    return names, pools, template.replace("N", "-"), _strand_code(strand)


def _combo_result(names, pools, indices: bool, idx, freq, total) -> ComboCounts:
    keys = idx + 1                                                  # :128
    combos: Dict[str, list] = {}
    for i, nm in enumerate(names):
        col = keys[i].tolist()
        combos[nm] = col if indices else [pools[i][k - 1] for k in col]   # :136-140
    return ComboCounts(names=names, combinations=combos, counts=freq, nreads=total)


def countComboBarcodes(fastq: str, template: str, choices, substitutions: int = 0, find_best: bool = False,
                       strand: str = "both", num_threads: int = 1, indices: bool = False) -> ComboCounts:
    """R/countComboBarcodes.R:87-124.  `choices` is a list of two pools or a dict name -> pool."""
    names, pools, native_template, strand_code = _combo_setup(template, choices, strand)
    idx, freq, total = count_combo_barcodes_single(fastq, native_template, strand_code, pools, substitutions, not find_best, num_threads)
    return _combo_result(names, pools, indices, idx, freq, total)


def combineComboCounts(*results: ComboCounts):
    """R/combineComboCounts.R:31-57: union of combinations (sorted), one count column per input."""
    names = results[0].names if results else ["first", "second"]
    keys = sorted({tuple(r.combinations[nm][j] for nm in names) for r in results for j in range(len(r.counts))})
    pos = {k: i for i, k in enumerate(keys)}
    mat = np.zeros((len(keys), len(results)), dtype=np.int32)
    for c, r in enumerate(results):
        for j in range(len(r.counts)):
            mat[pos[tuple(r.combinations[nm][j] for nm in names)], c] = r.counts[j]
    combos = {nm: [k[i] for k in keys] for i, nm in enumerate(names)}
    return combos, mat


def _device_jobs(devices, jobs_per_device: int):
    if devices is not None:
        return [d for d in devices for _ in range(max(int(jobs_per_device), 1))]
    if jobs_per_device > 1:
        return [d for d in range(max(int(_lib.load().scg_device_count()), 1)) for _ in range(int(jobs_per_device))]
    return None


def matrixOfComboBarcodes(files: Sequence[str], template: str, choices, substitutions: int = 0, find_best: bool = False, strand: str = "both",
                          num_threads: int = 1, indices: bool = False, withDimnames: bool = True, devices=None, jobs_per_device: int = 1) -> CountMatrix:
    """R/countComboBarcodes.R:149-164: all files in one native call (scg_count_combo_barcodes_single_files), then
    combineComboCounts (R/combineComboCounts.R:31-57)."""
    files = list(files)
    names, pools, native_template, strand_code = _combo_setup(template, choices, strand)
    per_file = count_combo_barcodes_single_files(files, native_template, strand_code, pools, substitutions, not find_best, num_threads,
                                                 _device_jobs(devices, jobs_per_device))
    out = [_combo_result(names, pools, indices, idx, freq, total) for idx, freq, total in per_file]
    combos, mat = combineComboCounts(*out)
    se = CountMatrix(counts=mat, row_data=combos,
                     col_data={"paths": list(files), "nreads": [o.nreads for o in out], "nmapped": mat.sum(axis=0).astype(np.int64).tolist()})
    if withDimnames:
        se.colnames = [os.path.basename(f) for f in files]
        se.rownames = [f"BARCODE_{i + 1}" for i in range(mat.shape[0])]
    return se


def _rep2(x):
    if isinstance(x, (str, bytes)) or not hasattr(x, "__len__"):
        return [x, x]
    x = list(x)
    return [x[i % len(x)] for i in range(2)]


def _dual_setup(choices, flank5, flank3, template, substitutions, strand):
    """Argument handling of R/countDualBarcodes.R:118-182 -> (names, col1, col2, template1, template2, subs[2], reverse[2])."""
    if isinstance(choices, dict):
        names = list(choices.keys())
        col1, col2 = [list(v) for v in choices.values()]
    else:
        names = ["first", "second"]
        col1, col2 = list(choices[0]), list(choices[1])
    if template is not None:                                        # :162-174
        t = _rep2(template)
        template1, template2 = re.sub("[nN]", "-", t[0]), re.sub("[nN]", "-", t[1])
    else:
        f5, f3 = _rep2(flank5), _rep2(flank3)
        template1 = f5[0] + "-" * len(col1[0]) + f3[0]
        template2 = f5[1] + "-" * len(col2[0]) + f3[1]
    subs = [int(x) for x in _rep2(substitutions)]
    strands = _rep2(strand)
    for st in strands:                                              # :176-182
        if st not in ("original", "reverse"):
            raise ValueError("'strand' should be one of 'original', 'reverse'")
    return names, col1, col2, template1, template2, subs, [st == "reverse" for st in strands]


def countDualBarcodes(fastq: Sequence[str], choices, flank5=None, flank3=None, template=None, substitutions=0,
                      find_best: bool = False, strand="original", randomized: bool = False, include_invalid: bool = False,
                      num_threads: int = 1) -> DualCounts:
    """R/countDualBarcodes.R:118-160.  `choices` is a dict / pair of two equally long columns."""
    names, col1, col2, template1, template2, subs, rev = _dual_setup(choices, flank5, flank3, template, substitutions, strand)
    out = count_dual_barcodes(fastq[0], template1, rev[0], subs[0], col1, fastq[1], template2, rev[1], subs[1], col2,
                              randomized, not find_best, include_invalid, num_threads)
    if not include_invalid:
        counts, total = out
        return DualCounts(choices={names[0]: col1, names[1]: col2}, counts=counts, npairs=total)
    counts, (idx, freq), total, b1, b2 = out                      # R/countDualBarcodes.R:152-159, :184-198
    inv1 = [col1[i] for i in idx[0]]
    inv2 = [col2[j] for j in idx[1]]
    return DualCounts(choices={names[0]: col1 + inv1, names[1]: col2 + inv2},
                      counts=np.concatenate([counts, freq]).astype(np.int32), npairs=total,
                      valid=[True] * len(col1) + [False] * len(inv1),
                      barcode1_only=b1, barcode2_only=b2, invalid_pair=int(freq.sum()))


def matrixOfDualBarcodes(files: Sequence[Sequence[str]], choices, withDimnames: bool = True, devices=None, jobs_per_device: int = 1,
                         flank5=None, flank3=None, template=None, substitutions=0, find_best: bool = False, strand="original",
                         randomized: bool = False, include_invalid: bool = False, num_threads: int = 1) -> CountMatrix:
    """R/countDualBarcodes.R:205-224: all file pairs in one native call (scg_count_dual_barcodes_files); with
    include.invalid=TRUE the rows differ from file to file (:226-254), so the pairs are counted one call each."""
    files = [list(f) for f in files]
    if include_invalid:
        out = _map_files(lambda f: countDualBarcodes(f, choices, flank5, flank3, template, substitutions, find_best, strand, randomized,
                                                     True, num_threads), files, devices, jobs_per_device)
        mat = np.stack([o.counts for o in out], axis=1) if out else np.zeros((0, 0), dtype=np.int32)
        row_data = out[0].choices if out else {}
        npairs = [o.npairs for o in out]
    else:
        names, col1, col2, template1, template2, subs, rev = _dual_setup(choices, flank5, flank3, template, substitutions, strand)
        mat, npairs = count_dual_barcodes_files([f[0] for f in files], template1, rev[0], subs[0], col1,
                                                [f[1] for f in files], template2, rev[1], subs[1], col2,
                                                randomized, not find_best, num_threads, _device_jobs(devices, jobs_per_device))
        row_data = {names[0]: col1, names[1]: col2}
    se = CountMatrix(counts=mat, row_data=row_data,
                     col_data={"paths1": [f[0] for f in files], "paths2": [f[1] for f in files], "npairs": npairs})
    if withDimnames:
        se.colnames = [os.path.basename(f[0]) for f in files]
    return se


def countDualBarcodesSingleEnd(fastq: str, choices, template: str, substitutions: int = 0, find_best: bool = False,
                               strand: str = "both", include_invalid: bool = False, num_threads: int = 1) -> DualCounts:
    """R/countDualBarcodesSingleEnd.R:85-122 (include.invalid=FALSE).  `choices`: dict / list of equally long columns,
    one per variable region of `template`."""
    if isinstance(choices, dict):
        names = list(choices.keys())
        cols = [list(v) for v in choices.values()]
    else:
        cols = [list(v) for v in choices]
        names = ["first", "second", "third", "fourth"][:len(cols)]
    out = count_dual_barcodes_single_end(fastq, re.sub("[nN]", "-", template), cols, _strand_code(strand),
                                         substitutions, not find_best, include_invalid, num_threads)
    if not include_invalid:
        counts, total = out
        return DualCounts(choices=dict(zip(names, cols)), counts=counts, npairs=total)
    counts, (idx, freq), total = out                              # R/countDualBarcodesSingleEnd.R:114-121
    inv = [[cols[r][i] for i in idx[r]] for r in range(2)]
    return DualCounts(choices={names[r]: cols[r] + inv[r] for r in range(2)},
                      counts=np.concatenate([counts, freq]).astype(np.int32), npairs=total,
                      valid=[True] * len(cols[0]) + [False] * len(inv[0]), invalid_pair=int(freq.sum()))


def matrixOfDualBarcodesSingleEnd(files: Sequence[str], choices, withDimnames: bool = True, devices=None, jobs_per_device: int = 1, **kwargs) -> CountMatrix:
    """R/countDualBarcodesSingleEnd.R:129-150 (include.invalid=FALSE)."""
    out = _map_files(lambda f: countDualBarcodesSingleEnd(f, choices, **kwargs), files, devices, jobs_per_device)
    nrow = len(out[0].counts) if out else 0
    mat = np.stack([o.counts for o in out], axis=1) if out else np.zeros((nrow, 0), dtype=np.int32)
    se = CountMatrix(counts=mat, row_data=out[0].choices if out else {},
                     col_data={"paths": list(files), "nreads": [o.npairs for o in out]})
    if withDimnames:
        se.colnames = [os.path.basename(f) for f in files]
    return se


def countRandomBarcodes(fastq: str, template: str, substitutions: int = 0, find_best: bool = False, strand: str = "both",
                        num_threads: int = 1):
    """R/countRandomBarcodes.R:61-77 -> dict(sequences sorted, counts, nreads)."""
    (seqs, freq), total = count_random_barcodes(fastq, template.replace("N", "-"), _strand_code(strand), substitutions,
                                                not find_best, num_threads)
    return {"sequences": seqs, "counts": freq, "nreads": total}


def matrixOfRandomBarcodes(files: Sequence[str], withDimnames: bool = True, devices=None, jobs_per_device: int = 1, **kwargs) -> CountMatrix:
    """R/countRandomBarcodes.R:84-108: rows = sorted union of the sequences seen in any file."""
    out = _map_files(lambda f: countRandomBarcodes(f, **kwargs), files, devices, jobs_per_device)
    keys = sorted(set().union(*[o["sequences"] for o in out])) if out else []
    pos = {k: i for i, k in enumerate(keys)}
    mat = np.zeros((len(keys), len(out)), dtype=np.int32)
    for c, o in enumerate(out):
        for sq, v in zip(o["sequences"], o["counts"].tolist()):
            mat[pos[sq], c] = v
    se = CountMatrix(counts=mat, row_data={"sequences": keys},
                     col_data={"paths": list(files), "nreads": [o["nreads"] for o in out],
                               "nmapped": mat.sum(axis=0).astype(np.int64).tolist()})
    if withDimnames:
        se.colnames = [os.path.basename(f) for f in files]
        se.rownames = list(keys)
    return se


def countPairedComboBarcodes(fastq: Sequence[str], choices, flank5=None, flank3=None, template=None, substitutions=0,
                             find_best: bool = False, strand="original", num_threads: int = 1, randomized: bool = False,
                             indices: bool = False) -> ComboCounts:
    """R/countPairedComboBarcodes.R:93-112.  `choices` is a pair (or dict) of two pools; the result carries
    npairs / barcode1_only / barcode2_only like the R function's metadata."""
    if isinstance(choices, dict):
        names = list(choices.keys())
        pools = [list(v) for v in choices.values()]
    else:
        names = ["first", "second"]
        pools = [list(choices[0]), list(choices[1])]
    if template is not None:
        t = _rep2(template)
        template1, template2 = re.sub("[nN]", "-", t[0]), re.sub("[nN]", "-", t[1])
    else:
        f5, f3 = _rep2(flank5), _rep2(flank3)
        template1 = f5[0] + "-" * len(pools[0][0]) + f3[0]
        template2 = f5[1] + "-" * len(pools[1][0]) + f3[1]
    subs = _rep2(substitutions)
    strands = _rep2(strand)
    for s in strands:
        if s not in ("original", "reverse"):
            raise ValueError("'strand' should be one of 'original', 'reverse'")
    idx, freq, total, b1, b2 = count_combo_barcodes_paired(fastq[0], template1, strands[0] == "reverse", int(subs[0]), pools[0],
                                                           fastq[1], template2, strands[1] == "reverse", int(subs[1]), pools[1],
                                                           randomized, not find_best, num_threads)
    keys = idx + 1
    combos: Dict[str, list] = {}
    for i, nm in enumerate(names):
        col = keys[i].tolist()
        combos[nm] = col if indices else [pools[i][k - 1] for k in col]
    return ComboCounts(names=names, combinations=combos, counts=freq, nreads=total, npairs=total, barcode1_only=b1, barcode2_only=b2)


def matrixOfPairedComboBarcodes(files: Sequence[Sequence[str]], withDimnames: bool = True, devices=None, jobs_per_device: int = 1, **kwargs) -> CountMatrix:
    """R/countPairedComboBarcodes.R:119-140."""
    out = _map_files(lambda f: countPairedComboBarcodes(f, **kwargs), files, devices, jobs_per_device)
    combos, mat = combineComboCounts(*out)
    se = CountMatrix(counts=mat, row_data=combos,
                     col_data={"paths1": [f[0] for f in files], "paths2": [f[1] for f in files],
                               "npairs": [o.npairs for o in out], "barcode1.only": [o.barcode1_only for o in out],
                               "barcode2.only": [o.barcode2_only for o in out]})
    if withDimnames:
        se.colnames = [os.path.basename(f[0]) for f in files]
        se.rownames = [f"BARCODE_{i + 1}" for i in range(mat.shape[0])]
    return se


def matchBarcodes(sequences: Sequence[str], choices: Sequence[str], substitutions: int = 0, reverse: bool = False):
    """R/matchBarcodes.R: 1-based index / mismatches with None for NA."""
    idx, mm = match_barcodes(sequences, choices, substitutions, reverse)
    index = [int(i) + 1 if i >= 0 else None for i in idx]
    mism = [int(m) if i >= 0 else None for i, m in zip(idx, mm)]
    return {"index": index, "mismatches": mism}
