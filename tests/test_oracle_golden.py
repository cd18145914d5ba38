"""The C restatement (oracle/scg_oracle.c) against the committed golden vectors:
the reference's own hand-written known answers and outputs of the real kaori (CPU only)."""
import gzip

import numpy as np
import pytest

from oracle.pyoracle import OracleError
from tests import golden_util as G

CASES = G.all_count_cases()


def run_oracle(oracle, c):
    k = c["kind"]
    if k == "single":
        counts, total = oracle.count_single(c["reads"], c["template"], c["strand"], c["pool"], c["mismatches"], c["use_first"])
        return {"counts": counts.tolist(), "total": total}
    if k == "combo":
        idx, freq, total = oracle.count_combo(c["reads"], c["template"], c["strand"], c["pool0"], c["pool1"], c["mismatches"], c["use_first"])
        return {"indices": idx.tolist(), "freq": freq.tolist(), "total": total}
    if k == "dual":
        counts, total = oracle.count_dual(c["reads1"], c["reads2"], c["template1"], c["reverse1"], c["mismatches1"], c["pool1"],
                                          c["template2"], c["reverse2"], c["mismatches2"], c["pool2"], c["randomized"], c["use_first"])
        return {"counts": counts.tolist(), "total": total}
    if k == "dual_diag":
        d = oracle.count_dual_diag(c["reads1"], c["reads2"], c["template1"], c["reverse1"], c["mismatches1"], c["pool1"],
                                   c["template2"], c["reverse2"], c["mismatches2"], c["pool2"], c["randomized"], c["use_first"])
        return {"counts": d["counts"].tolist(), "indices": d["indices"].tolist(), "freq": d["freq"].tolist(), "total": d["total"],
                "barcode1_only": d["barcode1_only"], "barcode2_only": d["barcode2_only"]}
    if k == "dual_single_end_diag":
        d = oracle.count_dual_single_end_diag(c["reads"], c["template"], c["strand"], c["pools"], c["mismatches"], c["use_first"])
        return {"counts": d["counts"].tolist(), "indices": d["indices"].tolist(), "freq": d["freq"].tolist(), "total": d["total"]}
    if k == "random":
        tally, total = oracle.count_random(c["reads"], c["template"], c["strand"], c["mismatches"], c["use_first"])
        return {"sequences": sorted(tally), "freq": [tally[s] for s in sorted(tally)], "total": total}
    if k == "dual_single_end":
        counts, total = oracle.count_dual_single_end(c["reads"], c["template"], c["strand"], c["pools"], c["mismatches"], c["use_first"])
        return {"counts": counts.tolist(), "total": total}
    if k == "paired_combo":
        d = oracle.count_combo_paired(c["reads1"], c["reads2"], c["template1"], c["reverse1"], c["mismatches1"], c["pool1"],
                                      c["template2"], c["reverse2"], c["mismatches2"], c["pool2"], c["randomized"], c["use_first"])
        return {"indices": d["indices"].tolist(), "freq": d["freq"].tolist(), "total": d["total"],
                "barcode1_only": d["barcode1_only"], "barcode2_only": d["barcode2_only"]}
    idx, mm = oracle.match_barcodes(c["sequences"], c["choices"], c["substitutions"], c["reverse"])
    return {"index": idx.tolist(), "mismatches": mm.tolist()}


def normalise(expect):
    e = dict(expect)
    if "indices" in e and e["indices"] == []:
        e["indices"] = [[], []]
    return e


@pytest.mark.parametrize("i", range(len(CASES)), ids=lambda i: f"{i}-{G.case_id(CASES[i])}")
def test_oracle_matches_golden(oracle, i):
    c = CASES[i]
    if "error" in c["expect"]:
        with pytest.raises(OracleError):
            run_oracle(oracle, c)
        return
    got = run_oracle(oracle, c)
    assert normalise(got) == normalise(c["expect"])
    r = c.get("r_expect")
    if r:   # the literal expectation written in the reference's R test
        for key, val in r.items():
            if key == "sum":
                assert sum(got["counts"]) == val
            else:
                assert got[key] == val


@pytest.mark.parametrize("case", G.load("fastq_cases.json"), ids=lambda c: c["name"])
def test_oracle_fastq(oracle, tmp_path, case):
    path = tmp_path / ("x.fastq.gz" if case["gz"] else "x.fastq")
    data = G.fastq_bytes(case)
    if case["gz"]:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        path.write_bytes(data)
    if "error" in case["expect"]:
        with pytest.raises(OracleError) as ei:
            oracle.parse_fastq(str(path))
        assert str(ei.value) == case["expect"]["error"]
        return
    seqs, offs = oracle.parse_fastq(str(path))
    reads = [bytes(seqs[int(offs[i]):int(offs[i + 1])]).decode("latin1") for i in range(len(offs) - 1)]
    assert reads == case["expect"]["reads"]


def test_oracle_error_messages(oracle):
    with pytest.raises(OracleError, match="duplicate sequences detected \\(1, 2\\)"):
        oracle.count_single(["ACGT"], "AC--GT", 0, ["AA", "AA"], 0, True)
    with pytest.raises(OracleError, match="same length"):
        oracle.count_single(["ACGT"], "AC--GT", 0, ["AA", "A"], 0, True)
    with pytest.raises(OracleError, match="expected one variable region"):
        oracle.count_single(["ACGT"], "AC--G-T", 0, ["AA"], 0, True)
    with pytest.raises(OracleError, match="should be the same as the barcode_pool region"):
        oracle.count_single(["ACGT"], "AC---GT", 0, ["AA"], 0, True)
    # IUPAC overlap is a duplicate too (MismatchTrie.hpp:119-122)
    with pytest.raises(OracleError, match="duplicate sequences detected \\(1, 2\\)"):
        oracle.count_single(["ACGT"], "AC--GT", 0, ["AN", "AC"], 0, True)


def test_oracle_matches_kaori_on_a_grid_of_1_6e9_combinations(oracle):
    """2 x 40 000 barcodes: the reference sorts and run-length encodes the combinations of any number of barcodes
    (kaori/utils.hpp:173-198, src/utils.h:14-45)."""
    case, expect = G.large_grid()
    got = run_oracle(oracle, case)
    assert got == expect
