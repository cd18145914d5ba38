"""The C-ABI library without a GPU: it loads, exports exactly what include/scg.h declares, and its
host-side half (argument checks, library compilation, FASTQ staging, combo compaction) behaves
like the reference.  No compute calls here."""
import ctypes as C
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

from tests import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "scg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(sc):
    lib = sc.load()
    declared = declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/scg.h but not exported by libscg.so"
    # and the Python binding knows every one of them
    from screencounter_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared


def test_exports_nothing_else(sc):
    from screencounter_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    # every defined dynamic symbol of any kind (T, W, D, B ...), not only the scg_-prefixed ones: the library is
    # built with -fvisibility=hidden and a version script (csrc/scg.map), so C++ helpers, kernel stubs and
    # libstdc++ instantiations must not leak
    exported = sorted(line.split()[-1] for line in out.splitlines() if len(line.split()) >= 3)
    assert exported == declared_symbols()


def test_no_ablation_switch_in_product():
    """SCG_ABLATE (phase ablation, wrong counts by design) exists only in -DSCG_ABLATE measurement builds."""
    from screencounter_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"SCG_ABLATE" not in blob


def test_no_oracle_in_product():
    """The product must never route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "screencounter_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "liboracle" not in text and "scg_oracle" not in text and "kaori_ref" not in text, f
    from screencounter_amd import _lib
    out = subprocess.check_output(["ldd", _lib.LIB_PATH], text=True)
    assert "oracle" not in out


def test_version_and_device_count(sc):
    lib = sc.load()
    assert lib.scg_version().startswith(b"scg ")
    assert lib.scg_device_count() >= 0


def expect_error(sc, code, pattern, fn, *args, **kwargs):
    with pytest.raises(sc.ScgError, match=pattern) as ei:
        fn(*args, **kwargs)
    assert ei.value.code == code, (ei.value.code, str(ei.value))


def test_plan_argument_checks_precede_device_errors(sc):
    """Every std::runtime_error the reference's constructors throw is reported as SCG_ERR_INVALID
    before any device work, so these hold with or without a GPU."""
    from screencounter_amd import _lib
    INV, UNS = _lib.SCG_ERR_INVALID, _lib.SCG_ERR_UNSUPPORTED
    P = sc.Plan
    expect_error(sc, INV, "same length \\(4\\)", P.single, "ACGT----TGCA", 2, ["AAAA", "AAA"])
    expect_error(sc, INV, "duplicate sequences detected \\(1, 2\\)", P.single, "ACGT----TGCA", 2, ["AAAA", "AAAA"])
    expect_error(sc, INV, "duplicate sequences detected \\(1, 3\\)", P.single, "ACGT----TGCA", 2, ["AAAN", "CCCC", "AAAG"])
    expect_error(sc, INV, "expected one variable region", P.single, "ACGT--A--TGCA", 2, ["AAAA"])
    expect_error(sc, INV, "expected one variable region", P.single, "ACGTTGCA", 2, ["AAAA"])
    expect_error(sc, INV, "barcode_pool sequences \\(4\\) should be the same as the barcode_pool region \\(5\\)", P.single, "ACGT-----TGCA", 2, ["AAAA"])
    expect_error(sc, INV, "unknown base 'X'", P.single, "ACXT----TGCA", 0, ["AAAA"])
    expect_error(sc, INV, "cannot complement unknown base 'X'", P.single, "ACXT----TGCA", 1, ["AAAA"])
    expect_error(sc, INV, "unknown base 'Z' detected when constructing the trie", P.single, "ACGT----TGCA", 2, ["AAZA"])
    expect_error(sc, INV, "longer than 256 bp", P.single, "A" * 250 + "----" + "C" * 10, 2, ["AAAA"])
    # (keys of up to 256 bases -- the longest template -- have an index; only matchBarcodes, which has no template, can ask for more)
    expect_error(sc, UNS, "longer than 256 bp", sc.match_barcodes, ["A" * 257], ["A" * 257], 0, False)
    expect_error(sc, INV, "length of 'barcode_pools' should equal the number of variable regions", P.dual_single_end, "ACGT----TG--CA", 2, [["AAAA"]])
    expect_error(sc, INV, "length of variable region 2 \\(2\\) should be the same as its sequences \\(3\\)", P.dual_single_end, "ACGT----TG--CA", 2, [["AAAA"], ["CCC"]])
    expect_error(sc, INV, "all entries of 'barcode_pools' should have the same length", P.dual_single_end, "ACGT----TG--CA", 2, [["AAAA", "CCCC"], ["CC"]])
    expect_error(sc, INV, "duplicate sequences detected \\(1, 2\\)", P.dual_single_end, "ACGT----TG--CA", 2, [["AAAA", "AAAA"], ["CC", "CC"]])
    nine = "AC" + "--GT" * 9
    expect_error(sc, UNS, "1 to 8 variable regions \\(got 9\\)", P.dual_single_end, nine, 2, [["AA"]] * 9)
    expect_error(sc, INV, "expected 2 variable regions", P.combo, "ACGT----TGCA", 2, ["AAAA"], ["CC"])
    expect_error(sc, INV, "length of variable region 2 \\(3\\) should be the same as its sequences \\(2\\)", P.combo, "ACGT----TG---CA", 2, ["AAAA"], ["CC"])
    expect_error(sc, INV, "both barcode pools should be of the same length", P.dual, "AC--GT", False, 0, ["AA", "CC"], "AC--GT", False, 0, ["AA"])
    expect_error(sc, INV, "expected one variable region in the second constant template", P.dual, "AC--GT", False, 0, ["AA"], "AC--G-T", False, 0, ["AA"])
    expect_error(sc, INV, "duplicate sequences detected \\(1, 2\\)", P.dual, "AC--GT", False, 0, ["AA", "AA"], "AC--GT", False, 0, ["CC", "CC"])
    expect_error(sc, INV, "length of variable sequences \\(3\\) should be the same as the variable region \\(2\\)",
                 P.dual, "AC--GT", False, 0, ["AAA"], "AC--GT", False, 0, ["CC"])


def test_valid_plan_needs_a_device_or_works(sc):
    """No CPU fallback: with no GPU a valid plan fails loudly with SCG_ERR_DEVICE."""
    from screencounter_amd import _lib
    if sc.load().scg_device_count() > 0:
        with sc.Plan.single("ACGT----TGCA", 2, ["AAAA", "CCCC"], 1, True) as p:
            assert p.num_counters == 2
    else:
        expect_error(sc, _lib.SCG_ERR_DEVICE, "no HIP device", sc.Plan.single, "ACGT----TGCA", 2, ["AAAA", "CCCC"], 1, True)
        # duplicates within one dual pool are fine (DualBarcodesPairedEnd.hpp:88-90): reaches the device stage
        expect_error(sc, _lib.SCG_ERR_DEVICE, "no HIP device", sc.Plan.dual, "AC--GT", False, 0, ["AA", "AA"], "AC--GT", False, 0, ["CC", "GG"])


def test_set_device_reports_missing_or_bad_devices(sc):
    """scg_set_device (worker threads of the matrixOf* scheduler): loud errors, never a silent default."""
    from screencounter_amd import _lib
    from screencounter_amd._lib import ERRCAP, errbuf
    L = sc.load()
    err = errbuf()
    n = L.scg_device_count()
    if n == 0:
        assert L.scg_set_device(0, err, ERRCAP) == _lib.SCG_ERR_DEVICE and b"no HIP device" in err.value
    else:
        assert L.scg_set_device(0, err, ERRCAP) == _lib.SCG_OK
        assert L.scg_set_device(n, err, ERRCAP) == _lib.SCG_ERR_DEVICE and b"out of range" in err.value
        assert L.scg_set_device(-1, err, ERRCAP) == _lib.SCG_ERR_DEVICE


def test_file_level_error_order(sc, tmp_path):
    """Missing file is reported first, like byteme::SomeFileReader in src/count_single_barcodes.cpp:30."""
    from screencounter_amd import _lib
    expect_error(sc, _lib.SCG_ERR_IO, "failed to open file", sc.count_single_barcodes, str(tmp_path / "nope.fastq"), "ACGT----TGCA", 2, ["AAAA", "AAA"], 0, True)
    fq = tmp_path / "ok.fastq"
    fq.write_bytes(b"@r\nACGTAAAATGCA\n+\nIIIIIIIIIIII\n")
    expect_error(sc, _lib.SCG_ERR_INVALID, "same length", sc.count_single_barcodes, str(fq), "ACGT----TGCA", 2, ["AAAA", "AAA"], 0, True)
    expect_error(sc, _lib.SCG_ERR_INVALID, "currently expecting only 2 variable regions", sc.count_combo_barcodes_single, str(fq), "ACGT----TGCA", 2, [["AAAA"]], 0, True)


@pytest.mark.parametrize("case", G.load("fastq_cases.json"), ids=lambda c: c["name"])
def test_fastq_stager_matches_reference(sc, tmp_path, case):
    """scg_parse_fastq (host code) against kaori::FastqReader's golden behaviour, messages included."""
    path = tmp_path / ("x.fastq.gz" if case["gz"] else "x.fastq")
    data = G.fastq_bytes(case)
    if case["gz"]:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        path.write_bytes(data)
    if "error" in case["expect"]:
        with pytest.raises(sc.ScgError) as ei:
            sc.parse_fastq(str(path))
        assert str(ei.value) == case["expect"]["error"]
        return
    seqs, offs = sc.parse_fastq(str(path))
    reads = [bytes(seqs[int(offs[i]):int(offs[i + 1])]).decode("latin1") for i in range(len(offs) - 1)]
    assert reads == case["expect"]["reads"]


def test_fastq_stager_large_random(sc, oracle, tmp_path):
    """Buffer-boundary behaviour: a multi-MB file with ragged, partly multi-line records."""
    rng = np.random.default_rng(5)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    chunks = []
    for i in range(60000):
        n = int(rng.integers(0, 300))
        s = letters[rng.integers(0, 5, n)].tobytes()
        q = b"I" * n
        if i % 97 == 0 and n > 10:   # multi-line record
            s = s[:n // 2] + b"\n" + s[n // 2:]
            q = q[:n // 3] + b"\n" + q[n // 3:]
        chunks.append(b"@r%d some text\n" % i + s + b"\n+\n" + q + b"\n")
    data = b"".join(chunks)
    for gz in (False, True):
        path = tmp_path / ("big.fastq.gz" if gz else "big.fastq")
        if gz:
            with gzip.open(path, "wb", compresslevel=1) as f:
                f.write(data)
        else:
            path.write_bytes(data)
        s1, o1 = sc.parse_fastq(str(path))
        s2, o2 = oracle.parse_fastq(str(path))
        assert np.array_equal(o1, o2) and np.array_equal(s1, s2)
        assert len(o1) - 1 == 60000


def test_combo_compact(sc):
    cells = np.zeros(12, dtype=np.int32)   # 3 x 4
    cells[[1, 6, 11]] = [5, 2, 7]
    idx, freq = sc.combo_compact(cells, 3, 4)
    assert idx.tolist() == [[0, 1, 2], [1, 2, 3]]
    assert freq.tolist() == [5, 2, 7]
    idx, freq = sc.combo_compact(np.zeros(12, dtype=np.int32), 3, 4)
    assert idx.shape == (2, 0) and freq.shape == (0,)


def test_r_level_helpers(sc):
    p = sc.parseBarcodeTemplate("ACGTNNNNAANNNTT")
    assert p["variable"] == {"pos": [5, 11], "len": [4, 3]}
    assert p["constant"] == ["ACGT", "AA", "TT"]
    from tests.rlevel import ComboCounts
    a = ComboCounts(["first", "second"], {"first": [1, 2], "second": [1, 1]}, np.array([3, 4], dtype=np.int32), 10)
    b = ComboCounts(["first", "second"], {"first": [2, 3], "second": [1, 2]}, np.array([5, 6], dtype=np.int32), 11)
    combos, mat = sc.combineComboCounts(a, b)
    assert combos == {"first": [1, 2, 3], "second": [1, 1, 2]}
    assert mat.tolist() == [[3, 0], [4, 5], [0, 6]]


@pytest.mark.parametrize("piece_kb", [1, 7, 64])
def test_parallel_fastq_reader_equals_sequential(sc, oracle, tmp_path, monkeypatch, piece_kb):
    """Plain 4-line FASTQ through the multi-threaded reader (tiny pieces => thousands of hand-overs
    between workers) equals the sequential parse; unusual files fall back and still agree."""
    rng = np.random.default_rng(11)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    recs = []
    for i in range(30000):
        n = int(rng.integers(0, 200))
        s = letters[rng.integers(0, 5, n)].tobytes()
        q = bytes(rng.integers(33, 75, n, dtype=np.uint8))       # qualities full of '@' and '+'
        recs.append(b"@r%d\n" % i + s + b"\n+\n" + q + b"\n")
    strict = tmp_path / "strict.fastq"
    strict.write_bytes(b"".join(recs))
    nofinal = tmp_path / "nofinal.fastq"
    nofinal.write_bytes(b"".join(recs)[:-1])
    multi = tmp_path / "multi.fastq"
    odd = list(recs)
    odd[17000] = b"@multi\nACGT\nACGT\n+\nIIII\nIIII\n"           # one multi-line record deep inside
    multi.write_bytes(b"".join(odd))
    broken = tmp_path / "broken.fastq"
    bad = list(recs)
    bad[23000] = b"@short\nACGT\n+\nII\n"
    broken.write_bytes(b"".join(bad))

    monkeypatch.setenv("SCG_HOST_THREADS", "5")
    monkeypatch.setenv("SCG_FASTQ_PIECE_KB", str(piece_kb))
    for path in (strict, nofinal, multi):
        s1, o1 = sc.parse_fastq(str(path))
        s2, o2 = oracle.parse_fastq(str(path))
        assert np.array_equal(o1, o2) and np.array_equal(s1, s2), path.name
    from oracle.pyoracle import OracleError
    with pytest.raises(OracleError) as e_ref:
        oracle.parse_fastq(str(broken))
    with pytest.raises(sc.ScgError) as e_got:
        sc.parse_fastq(str(broken))
    assert str(e_got.value) == str(e_ref.value)
