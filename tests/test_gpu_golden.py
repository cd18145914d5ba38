"""HIP path against the committed golden vectors, through the file-level C ABI entry points
(the functions the Rcpp shim binds): FASTQ on disk -> counts.  Expected values are kaori's own
outputs and the literal expectations of the reference's R tests."""
import gzip

import numpy as np
import pytest

from oracle.pyoracle import write_fastq
from tests import golden_util as G

pytestmark = pytest.mark.gpu

CASES = G.all_count_cases()


def run_file_level(sc, c, tmp_path, gz=False):
    k = c["kind"]
    ext = ".fastq.gz" if gz else ".fastq"
    if k == "single":
        fq = str(tmp_path / ("s" + ext))
        write_fastq(fq, c["reads"], gz=gz)
        counts, total = sc.count_single_barcodes(fq, c["template"], c["strand"], c["pool"], c["mismatches"], c["use_first"], 1)
        return {"counts": counts.tolist(), "total": total}
    if k == "combo":
        fq = str(tmp_path / ("c" + ext))
        write_fastq(fq, c["reads"], gz=gz)
        idx, freq, total = sc.count_combo_barcodes_single(fq, c["template"], c["strand"], [c["pool0"], c["pool1"]], c["mismatches"], c["use_first"], 1)
        return {"indices": idx.tolist(), "freq": freq.tolist(), "total": total}
    if k == "dual":
        f1, f2 = str(tmp_path / ("d1" + ext)), str(tmp_path / ("d2" + ext))
        write_fastq(f1, c["reads1"], gz=gz)
        write_fastq(f2, c["reads2"], gz=gz)
        counts, total = sc.count_dual_barcodes(f1, c["template1"], c["reverse1"], c["mismatches1"], c["pool1"],
                                               f2, c["template2"], c["reverse2"], c["mismatches2"], c["pool2"],
                                               c["randomized"], c["use_first"], False, 1)
        return {"counts": counts.tolist(), "total": total}
    if k == "dual_diag":
        f1, f2 = str(tmp_path / ("g1" + ext)), str(tmp_path / ("g2" + ext))
        write_fastq(f1, c["reads1"], gz=gz)
        write_fastq(f2, c["reads2"], gz=gz)
        counts, (idx, freq), total, b1, b2 = sc.count_dual_barcodes(f1, c["template1"], c["reverse1"], c["mismatches1"], c["pool1"],
                                                                    f2, c["template2"], c["reverse2"], c["mismatches2"], c["pool2"],
                                                                    c["randomized"], c["use_first"], True, 1)
        return {"counts": counts.tolist(), "indices": idx.tolist(), "freq": freq.tolist(), "total": total,
                "barcode1_only": b1, "barcode2_only": b2}
    if k == "dual_single_end_diag":
        fq = str(tmp_path / ("x" + ext))
        write_fastq(fq, c["reads"], gz=gz)
        counts, (idx, freq), total = sc.count_dual_barcodes_single_end(fq, c["template"], c["pools"], c["strand"], c["mismatches"], c["use_first"], True, 1)
        return {"counts": counts.tolist(), "indices": idx.tolist(), "freq": freq.tolist(), "total": total}
    if k == "random":
        fq = str(tmp_path / ("r" + ext))
        write_fastq(fq, c["reads"], gz=gz)
        (seqs, freq), total = sc.count_random_barcodes(fq, c["template"], c["strand"], c["mismatches"], c["use_first"], 1)
        return {"sequences": seqs, "freq": freq.tolist(), "total": total}
    if k == "dual_single_end":
        fq = str(tmp_path / ("e" + ext))
        write_fastq(fq, c["reads"], gz=gz)
        counts, total = sc.count_dual_barcodes_single_end(fq, c["template"], c["pools"], c["strand"], c["mismatches"], c["use_first"], False, 1)
        return {"counts": counts.tolist(), "total": total}
    if k == "paired_combo":
        f1, f2 = str(tmp_path / ("p1" + ext)), str(tmp_path / ("p2" + ext))
        write_fastq(f1, c["reads1"], gz=gz)
        write_fastq(f2, c["reads2"], gz=gz)
        idx, freq, total, b1, b2 = sc.count_combo_barcodes_paired(f1, c["template1"], c["reverse1"], c["mismatches1"], c["pool1"],
                                                                  f2, c["template2"], c["reverse2"], c["mismatches2"], c["pool2"],
                                                                  c["randomized"], c["use_first"], 1)
        return {"indices": idx.tolist(), "freq": freq.tolist(), "total": total, "barcode1_only": b1, "barcode2_only": b2}
    idx, mm = sc.match_barcodes(c["sequences"], c["choices"], c["substitutions"], c["reverse"])
    return {"index": idx.tolist(), "mismatches": mm.tolist()}


def normalise(e):
    e = dict(e)
    if "indices" in e and e["indices"] == []:
        e["indices"] = [[], []]
    return e


@pytest.mark.parametrize("i", range(len(CASES)), ids=lambda i: f"{i}-{G.case_id(CASES[i])}")
def test_gpu_matches_golden(sc, gpu, tmp_path, i):
    c = CASES[i]
    if "error" in c["expect"]:
        with pytest.raises(sc.ScgError):
            run_file_level(sc, c, tmp_path)
        return
    got = run_file_level(sc, c, tmp_path, gz=(i % 5 == 0))
    assert normalise(got) == normalise(c["expect"])
    r = c.get("r_expect")
    if r:
        for key, val in r.items():
            if key == "sum":
                assert sum(got["counts"]) == val
            else:
                assert got[key] == val


def test_r_level_wrappers(sc, gpu, tmp_path):
    """countSingleBarcodes / countComboBarcodes / countDualBarcodes / matchBarcodes with the R
    calling conventions (N in templates, strand names, 1-based indices)."""
    fq = str(tmp_path / "s.fastq")
    reads = ["ACGTGGGGGGGGGGTGCA", "ACGTGGGGCGGGGGTGCA", "ACGTGGGGCCGGGGTGCA", "ACGTGGGGGGGGGGTTCA", "CCGTGGGGGGGGGGTGCA"]
    write_fastq(fq, reads)
    choices = ["A" * 10, "C" * 10, "G" * 10, "T" * 10]
    out = sc.countSingleBarcodes(fq, choices, template="ACGTNNNNNNNNNNTGCA", substitutions=1)     # test-single.R:66-69
    assert out.counts.tolist() == [0, 0, 4, 0] and out.nreads == 5 and out.choices == choices
    out2 = sc.countSingleBarcodes(fq, choices, flank5="ACGT", flank3="TGCA", substitutions=1)
    assert out2.counts.tolist() == [0, 0, 4, 0]
    se = sc.matrixOfSingleBarcodes([fq, fq], choices, template="ACGTNNNNNNNNNNTGCA", substitutions=1)
    assert se.counts.tolist() == [[0, 0], [0, 0], [4, 4], [0, 0]]
    assert se.col_data["nreads"] == [5, 5] and se.col_data["nmapped"] == [4, 4] and se.rownames == choices

    m = sc.matchBarcodes(["AAAAAA", "AAATAA"], ["AAAAAA", "CCCCCC", "GGGGGG", "TTTTTT"], substitutions=1, reverse=True)   # test-matchBarcodes.R:19-21
    assert m == {"index": [4, 4], "mismatches": [0, 1]}
    m = sc.matchBarcodes(["AAAAAA", "AAATAA"], ["AAAAAA", "CCCCCC", "GGGGGG", "TTTTTT"])
    assert m == {"index": [1, None], "mismatches": [0, None]}

    cq = str(tmp_path / "c.fastq")
    p1, p2 = ["AAAA", "CCCC", "GGGG"], ["TT", "GG"]
    write_fastq(cq, ["ACGT" + p1[i] + "AAAA" + p2[j] + "TGCA" for i, j in [(0, 0), (2, 1), (0, 0), (1, 1)]])
    co = sc.countComboBarcodes(cq, "ACGTNNNNAAAANNTGCA", [p1, p2], indices=True)
    assert co.combinations == {"first": [1, 2, 3], "second": [1, 2, 2]} and co.counts.tolist() == [2, 1, 1] and co.nreads == 4
    co = sc.countComboBarcodes(cq, "ACGTNNNNAAAANNTGCA", {"x": p1, "y": p2})
    assert co.combinations == {"x": ["AAAA", "CCCC", "GGGG"], "y": ["TT", "GG", "GG"]}

    d = sc.countDualBarcodes([fq, fq], {"a": choices, "b": choices}, template="ACGTNNNNNNNNNNTGCA", substitutions=1)   # test-dual.R:63-66
    assert int(d.counts.sum()) == 4 and d.npairs == 5


def test_matrix_of_files_scheduled_over_devices(sc, oracle, gpu, tmp_path):
    """matrixOf* (SURVEY 8f rank 3): files are handed to one worker thread per listed device; listing GPU 0
    twice exercises the concurrent path on a one-GPU box.  Columns must equal the per-file results."""
    import random
    from tests import gen
    rng = random.Random(77)
    template = "ACGTAC" + "N" * 8 + "TGCATG"
    pool = gen.make_pool(rng, 12, 8, "ACGT")
    files, expect, nreads = [], [], []
    for i in range(5):
        reads = gen.make_reads(rng, template.replace("N", "-"), [pool], 300 + 40 * i, 2, 0.02, 0.005, 0.0, 0.1, 12)
        fq = str(tmp_path / f"m{i}.fastq")
        write_fastq(fq, reads)
        files.append(fq)
        counts, total = oracle.count_single(reads, template.replace("N", "-"), 2, pool, 1, True)
        expect.append(counts)
        nreads.append(total)
    for devices in ([0], [0, 0], None):
        se = sc.matrixOfSingleBarcodes(files, pool, template=template, substitutions=1, devices=devices)
        assert np.array_equal(se.counts, np.stack(expect, axis=1))
        assert se.col_data["nreads"] == nreads and se.col_data["paths"] == files
    pairs = [(f, f) for f in files]        # mates must have equal read counts
    seq = sc.matrixOfDualBarcodes(pairs[:3], {"a": pool, "b": pool}, template=template, substitutions=1, devices=[0])
    par = sc.matrixOfDualBarcodes(pairs[:3], {"a": pool, "b": pool}, template=template, substitutions=1, devices=[0, 0, 0])
    assert np.array_equal(seq.counts, par.counts) and seq.col_data == par.col_data
    rnd = sc.matrixOfRandomBarcodes(files[:3], template=template, substitutions=1, devices=[0], jobs_per_device=2)
    one = [sc.countRandomBarcodes(f, template, substitutions=1) for f in files[:3]]
    assert rnd.row_data["sequences"] == sorted(set().union(*[o["sequences"] for o in one])) and rnd.col_data["nreads"] == nreads[:3]
    for c, o in enumerate(one):
        assert int(rnd.counts[:, c].sum()) == int(o["counts"].sum())
    seq = sc.matrixOfPairedComboBarcodes(pairs[:3], choices=[pool, pool], template=template, substitutions=1, devices=[0])
    par = sc.matrixOfPairedComboBarcodes(pairs[:3], choices=[pool, pool], template=template, substitutions=1, devices=[0, 0])
    assert np.array_equal(seq.counts, par.counts) and seq.row_data == par.row_data and seq.col_data == par.col_data


def test_paired_files_with_different_read_counts(sc, gpu, tmp_path):
    f1, f2 = str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq")
    write_fastq(f1, ["ACGTAATGCA"] * 3)
    write_fastq(f2, ["ACGTAATGCA"] * 2)
    with pytest.raises(sc.ScgError, match="different number of reads in paired FASTQ files"):
        sc.count_dual_barcodes(f1, "ACGT--TGCA", False, 0, ["AA"], f2, "ACGT--TGCA", False, 0, ["AA"], False, True, False, 1)


def test_empty_and_short_inputs(sc, gpu, tmp_path):
    fq = str(tmp_path / "e.fastq")
    open(fq, "wb").close()
    counts, total = sc.count_single_barcodes(fq, "ACGT--TGCA", 2, ["AA", "CC"], 1, True, 1)
    assert counts.tolist() == [0, 0] and total == 0
    write_fastq(fq, ["", "A", "ACGTAATGC", "ACGTAATGCA"])     # shorter than the template, then exact fit
    counts, total = sc.count_single_barcodes(fq, "ACGT--TGCA", 2, ["AA", "CC"], 0, True, 1)
    assert counts.tolist() == [1, 0] and total == 4


def test_paired_files_parallel_stager(sc, oracle, gpu, tmp_path, monkeypatch):
    """Two plain FASTQ files with ragged reads through the multi-threaded paired stager (tiny pieces
    => many windows whose read counts differ between the files) == the oracle on the parsed reads."""
    import random
    from tests import gen
    rng = random.Random(404)
    case = gen.random_dual_case(rng, hazard_free=True, sizes=(3000,), max_mm=1)
    f1, f2 = str(tmp_path / "p1.fastq"), str(tmp_path / "p2.fastq")
    write_fastq(f1, case["reads1"])
    write_fastq(f2, case["reads2"])
    exp = oracle.count_dual(case["reads1"], case["reads2"], case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                            case["template2"], case["reverse2"], case["mismatches2"], case["pool2"], case["randomized"], case["use_first"])
    monkeypatch.setenv("SCG_HOST_THREADS", "6")
    for piece_kb in ("2", "13", "4096"):
        monkeypatch.setenv("SCG_FASTQ_PIECE_KB", piece_kb)
        got = sc.count_dual_barcodes(f1, case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                                     f2, case["template2"], case["reverse2"], case["mismatches2"], case["pool2"],
                                     case["randomized"], case["use_first"], False, 1)
        assert got[1] == exp[1] == 3000 and np.array_equal(got[0], exp[0]), piece_kb
    # one read fewer in the second file
    write_fastq(f2, case["reads2"][:-1])
    with pytest.raises(sc.ScgError, match="different number of reads in paired FASTQ files"):
        sc.count_dual_barcodes(f1, case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                               f2, case["template2"], case["reverse2"], case["mismatches2"], case["pool2"],
                               case["randomized"], case["use_first"], False, 1)
