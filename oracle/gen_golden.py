#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REAL reference (oracle/_ref/libkaori_ref.so).

Runs only in the container that has /root/reference (where `make -C oracle` can build _ref/).
The fixtures are data: inputs (templates, pools, reads as strings) and the outputs kaori v1.1.1
produced for them.  Two files:

  tests/golden/known_answers.json  -- the hand-written known-answer vectors of the reference's own
        R tests (tests/testthat/test-single.R, test-matchBarcodes.R, test-dual.R), transcribed as
        inputs + the literal expectations of those tests (`r_expect`), plus kaori's full output.
        Generation asserts that kaori reproduces every `r_expect`.
  tests/golden/kaori_random.json   -- seeded random cases (tests/gen.py) for single / combo /
        dual / matchBarcodes incl. IUPAC libraries, Ns, lower case, ties, both strands,
        first/best, randomized pairs; dual cases are hazard-free w.r.t. the reference's
        order-dependent cache and were run with nthreads=1.
  tests/golden/fastq_cases.json    -- FASTQ texts (incl. multi-line records, CRLF, missing final
        newline, malformed files) with the sequences / error kaori::FastqReader yields.

    python oracle/gen_golden.py
"""
from __future__ import annotations

import base64
import json
import os
import random
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import KaoriRef, OracleError, write_fastq  # noqa: E402
from tests import gen  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
BASES = ["A", "C", "G", "T"]


def run_case(ref: KaoriRef, case: dict, tmp: str) -> dict:
    """Adds kaori's outputs (or its error message) to a case dict."""
    k = case["kind"]
    out = dict(case)
    try:
        if k == "single":
            fq = os.path.join(tmp, "s.fastq")
            write_fastq(fq, case["reads"])
            counts, total = ref.count_single(fq, case["template"], case["strand"], case["pool"], case["mismatches"], case["use_first"], 1)
            out["expect"] = {"counts": counts.tolist(), "total": total}
        elif k == "combo":
            fq = os.path.join(tmp, "c.fastq")
            write_fastq(fq, case["reads"])
            idx, freq, total = ref.count_combo(fq, case["template"], case["strand"], case["pool0"], case["pool1"], case["mismatches"], case["use_first"], 1)
            out["expect"] = {"indices": idx.tolist(), "freq": freq.tolist(), "total": total}
        elif k == "dual":
            f1, f2 = os.path.join(tmp, "d1.fastq"), os.path.join(tmp, "d2.fastq")
            write_fastq(f1, case["reads1"])
            write_fastq(f2, case["reads2"])
            counts, total = ref.count_dual(f1, case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                                           f2, case["template2"], case["reverse2"], case["mismatches2"], case["pool2"],
                                           case["randomized"], case["use_first"], 1)
            out["expect"] = {"counts": counts.tolist(), "total": total}
        elif k == "dual_diag":
            f1, f2 = os.path.join(tmp, "g1.fastq"), os.path.join(tmp, "g2.fastq")
            write_fastq(f1, case["reads1"])
            write_fastq(f2, case["reads2"])
            d = ref.count_dual_diag(f1, case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                                    f2, case["template2"], case["reverse2"], case["mismatches2"], case["pool2"],
                                    case["randomized"], case["use_first"], 1)
            out["expect"] = {"counts": d["counts"].tolist(), "indices": d["indices"].tolist(), "freq": d["freq"].tolist(),
                             "total": d["total"], "barcode1_only": d["barcode1_only"], "barcode2_only": d["barcode2_only"]}
        elif k == "dual_single_end_diag":
            fq = os.path.join(tmp, "x.fastq")
            write_fastq(fq, case["reads"])
            d = ref.count_dual_single_end_diag(fq, case["template"], case["strand"], case["pools"], case["mismatches"], case["use_first"], 1)
            out["expect"] = {"counts": d["counts"].tolist(), "indices": d["indices"].tolist(), "freq": d["freq"].tolist(), "total": d["total"]}
        elif k == "random":
            fq = os.path.join(tmp, "r.fastq")
            write_fastq(fq, case["reads"])
            tally, total = ref.count_random(fq, case["template"], case["strand"], case["mismatches"], case["use_first"], 1)
            out["expect"] = {"sequences": sorted(tally), "freq": [tally[k2] for k2 in sorted(tally)], "total": total}
        elif k == "dual_single_end":
            fq = os.path.join(tmp, "e.fastq")
            write_fastq(fq, case["reads"])
            counts, total = ref.count_dual_single_end(fq, case["template"], case["strand"], case["pools"], case["mismatches"], case["use_first"], 1)
            out["expect"] = {"counts": counts.tolist(), "total": total}
        elif k == "paired_combo":
            f1, f2 = os.path.join(tmp, "p1.fastq"), os.path.join(tmp, "p2.fastq")
            write_fastq(f1, case["reads1"])
            write_fastq(f2, case["reads2"])
            d = ref.count_combo_paired(f1, case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                                       f2, case["template2"], case["reverse2"], case["mismatches2"], case["pool2"],
                                       case["randomized"], case["use_first"], 1)
            out["expect"] = {"indices": d["indices"].tolist(), "freq": d["freq"].tolist(),
                             "total": d["total"], "barcode1_only": d["barcode1_only"], "barcode2_only": d["barcode2_only"]}
        elif k == "match":
            idx, mm = ref.match_barcodes(case["sequences"], case["choices"], case["substitutions"], case["reverse"])
            out["expect"] = {"index": idx.tolist(), "mismatches": mm.tolist()}
        else:
            raise ValueError(k)
    except OracleError as e:
        out["expect"] = {"error": str(e)}
    return out


def known_answers() -> list:
    tmpl10 = "ACGT" + "-" * 10 + "TGCA"
    poly = ["A" * 10, "C" * 10, "G" * 10, "T" * 10]
    cases = []

    # tests/testthat/test-single.R:55-72 -- one shared mismatch budget over flanks + barcode
    cases.append(dict(kind="single", source="test-single.R:55-72", template=tmpl10, strand=2, pool=poly, mismatches=1, use_first=True,
                      reads=["ACGTGGGGGGGGGGTGCA", "ACGTGGGGCGGGGGTGCA", "ACGTGGGGCCGGGGTGCA", "ACGTGGGGGGGGGGTTCA", "CCGTGGGGGGGGGGTGCA"],
                      r_expect={"counts": [0, 0, 4, 0], "total": 5}))
    # :74-91 -- two barcodes at equal distance => discarded
    cases.append(dict(kind="single", source="test-single.R:74-91", template=tmpl10, strand=2, pool=["CCCCCCCCCC", "CCCCCCCCCA"], mismatches=1, use_first=True,
                      reads=["ACGTCCCCCCCCCCTGCA", "ACGTCCCCCCCCCATGCA", "ACGTCCCCCCCCCGTGCA", "ACGTCCCCCCCCCTTGCA"],
                      r_expect={"counts": [1, 1], "total": 4}))
    # :126-148 -- IUPAC codes in the library
    iupac_pool = ["AAAAABAAAA", "CCCCCDCCCC", "GGGGGHGGGG", "TTTTTVTTTT"]
    iupac_reads = ["ACGT" + s + "TGCA" for s in ["AAAAACAAAA", "AAAAAGAAAA", "AAAAAAAAAA", "CCCCCACCCC", "CCCCAACCCC", "GGGGGTGGGG", "TTTTTGTTTT"]]
    cases.append(dict(kind="single", source="test-single.R:126-144", template=tmpl10, strand=0, pool=iupac_pool, mismatches=0, use_first=True,
                      reads=iupac_reads, r_expect={"counts": [2, 1, 1, 1], "total": 7}))
    cases.append(dict(kind="single", source="test-single.R:146-147", template=tmpl10, strand=0, pool=iupac_pool, mismatches=1, use_first=True,
                      reads=iupac_reads, r_expect={"counts": [3, 2, 1, 1], "total": 7}))
    # same vectors under find.best=TRUE (test-single.R:50-52 expects first == best on clean data)
    cases.append(dict(kind="single", source="test-single.R:55-72 (find.best)", template=tmpl10, strand=2, pool=poly, mismatches=1, use_first=False,
                      reads=cases[0]["reads"], r_expect={"counts": [0, 0, 4, 0], "total": 5}))

    # tests/testthat/test-matchBarcodes.R:4-22 (index 0-based here, NA = -1)
    ch = ["AAAAAA", "CCCCCC", "GGGGGG", "TTTTTT"]
    q = ["AAAAAA", "AAATAA"]
    cases.append(dict(kind="match", source="test-matchBarcodes.R:7-9", sequences=q, choices=ch, substitutions=0, reverse=False,
                      r_expect={"index": [0, -1], "mismatches": [0, -1]}))
    cases.append(dict(kind="match", source="test-matchBarcodes.R:11-13", sequences=q, choices=ch, substitutions=1, reverse=False,
                      r_expect={"index": [0, 0], "mismatches": [0, 1]}))
    cases.append(dict(kind="match", source="test-matchBarcodes.R:15-17", sequences=q, choices=ch, substitutions=0, reverse=True,
                      r_expect={"index": [3, -1], "mismatches": [0, -1]}))
    cases.append(dict(kind="match", source="test-matchBarcodes.R:19-21", sequences=q, choices=ch, substitutions=1, reverse=True,
                      r_expect={"index": [3, 3], "mismatches": [0, 1]}))
    # :24-38 IUPAC
    ch2 = ["AAARAA", "CCCYCC", "GGGMGG", "TTTSTT"]
    cases.append(dict(kind="match", source="test-matchBarcodes.R:27-29", sequences=["AAAAAA", "AAAGAA"], choices=ch2, substitutions=0, reverse=False,
                      r_expect={"index": [0, 0], "mismatches": [0, 0]}))
    cases.append(dict(kind="match", source="test-matchBarcodes.R:31-33", sequences=["AAAAAA", "AAAGAA", "AAGAAA"], choices=ch2, substitutions=0, reverse=True,
                      r_expect={"index": [-1, -1, 3], "mismatches": [-1, -1, 0]}))
    cases.append(dict(kind="match", source="test-matchBarcodes.R:35-37", sequences=["AAAAAA", "AAAGAA", "AAGAAA"], choices=ch2, substitutions=2, reverse=True,
                      r_expect={"index": [3, 3, 3], "mismatches": [1, 2, 0]}))

    # tests/testthat/test-dual.R:46-93 -- per-mate substitution budgets; indel reads never match
    e1 = ["ACGTGGGGGGGGGGTGCA", "ACGTGGGGCGGGGGTGCA", "ACGTGGGGGGGGGTGCA", "ACGTGGGGGGGGGGGTGCA"]
    e2 = ["ACGTGGGGCGGGGGTGCA", "ACGTGGGGGGGGGGTGCA", "ACGTGGGGGGGGGGGTGCA", "ACGTGGGGGGGGGTGCA"]

    def dual(src, r1, r2, m1, m2, rsum, randomized=False, pool1=poly, pool2=poly, t1=tmpl10, t2=tmpl10, first=True, rexp=None):
        d = dict(kind="dual", source=src, template1=t1, reverse1=False, mismatches1=m1, pool1=pool1,
                 template2=t2, reverse2=False, mismatches2=m2, pool2=pool2, randomized=randomized, use_first=first,
                 reads1=r1, reads2=r2)
        d["r_expect"] = rexp if rexp is not None else {"sum": rsum}
        return d
    cases.append(dual("test-dual.R:59-60", e1, e1, 0, 0, 1))
    cases.append(dual("test-dual.R:65-66", e1, e1, 1, 1, 2))
    cases.append(dual("test-dual.R:83-84", e1, e2, 0, 0, 0))
    cases.append(dual("test-dual.R:86-87", e1, e2, 0, 1, 1))
    cases.append(dual("test-dual.R:89-90", e1, e2, 1, 0, 1))
    cases.append(dual("test-dual.R:92-93", e1, e2, 1, 1, 2))

    # test-dual.R:173-218 -- randomization edge cases, all-variable template
    r1 = ["AAAAAAAAA", "AAAAAAAAA", "AAAAAACAA", "AAAAAACAA"]
    r2 = ["AAAAAAAAA", "AAAAAACAA", "AAAAAAAAA", "AAAAAACAA"]
    tv = "-" * 9
    one = ["AAAAAAAAA"]
    cases.append(dual("test-dual.R:194-195", r1, r2, 0, 0, 1, pool1=one, pool2=one, t1=tv, t2=tv, rexp={"counts": [1]}))
    cases.append(dual("test-dual.R:197-198", r1, r2, 0, 0, 1, randomized=True, pool1=one, pool2=one, t1=tv, t2=tv, rexp={"counts": [1]}))
    cases.append(dual("test-dual.R:200-201", r1, r2, 1, 0, 2, pool1=one, pool2=one, t1=tv, t2=tv, rexp={"counts": [2]}))
    cases.append(dual("test-dual.R:203-204", r1, r2, 0, 1, 2, pool1=one, pool2=one, t1=tv, t2=tv, rexp={"counts": [2]}))
    cases.append(dual("test-dual.R:206-207", r1, r2, 0, 1, 3, randomized=True, pool1=one, pool2=one, t1=tv, t2=tv, rexp={"counts": [3]}))
    cases.append(dual("test-dual.R:213-216", r1, r2, 1, 1, 4, randomized=True, pool1=["AAAAAAAAA", "AAAAAACAA"], pool2=["AAAAAAAAA", "AAAAAAAAA"],
                      t1=tv, t2=tv, rexp={"counts": [2, 2]}))

    # tests/testthat/test-countDualBarcodesSingleEnd.R:38-59 -- both variable regions in one read, one shared
    # substitution budget (defaults: strand = "both", find.best = FALSE); choices2 is recycled over the 4 rows
    se_reads = ["ACGTGGGGGGGGGGTGCAAGGAAAAAAAAAAAAAAAAAGGA",
                "ACGTGGGGGGGGGGTGCAAGGAAAAAAAAAAATAAAAAGGA",
                "ACGTGGGGCGGGGGTGCAAGGAAAAAAAAAAATAAAAAGGA"]
    se_tmpl = "ACGT" + "-" * 10 + "TGCAAGGA" + "-" * 15 + "AGGA"
    for subs, line, exp in ((0, "50-51", [0, 0, 1, 0]), (1, "53-54", [0, 0, 2, 0]), (2, "56-57", [0, 0, 3, 0])):
        cases.append(dict(kind="dual_single_end", source=f"test-countDualBarcodesSingleEnd.R:{line}", template=se_tmpl, strand=2,
                          pools=[poly, ["A" * 15] * 4], mismatches=subs, use_first=True, reads=se_reads,
                          r_expect={"counts": exp}))
    return cases


def check_r_expect(case: dict) -> None:
    r, e = case.get("r_expect"), case["expect"]
    if r is None:
        return
    for key, val in r.items():
        if key == "sum":
            assert sum(e["counts"]) == val, (case["source"], e, r)
        else:
            assert e[key] == val, (case["source"], key, e, r)


def fastq_cases(ref: KaoriRef, tmp: str) -> list:
    texts = {
        "plain": b"@r1\nACGT\n+\nIIII\n@r2 desc\nGGCC\n+r2\nIIII\n",
        "no_final_newline": b"@r1\nACGT\n+\nIIII\n@r2\nTTTT\n+\nIIII",
        "multiline": b"@r1\nAC\nGT\nAA\n+\nII\nII\nII\n@r2\nT\n+\nI\n",
        "crlf": b"@r1\r\nACGT\r\n+\r\nIIIII\r\n",
        "empty_file": b"",
        "empty_read": b"@r1\n\n+\n\n@r2\nAC\n+\nII\n",
        "quality_with_at": b"@r1\nACGT\n+\n@@@@\n@r2\nGG\n+\n@I\n",
        "lowercase_and_n": b"@r1\nacgtNNRY\n+\nIIIIIIII\n",
        "bad_start": b"r1\nACGT\n+\nIIII\n",
        "short_quality": b"@r1\nACGT\n+\nIII\n@r2\nAC\n+\nII\n",
        "long_quality": b"@r1\nACGT\n+\nIIIII\n",
        "truncated_in_seq": b"@r1\nACGT",
        "truncated_in_name": b"@r1",
        "truncated_after_plus": b"@r1\nACGT\n+",
        "second_record_bad": b"@r1\nACGT\n+\nIIII\nXr2\nAC\n+\nII\n",
        "plus_in_sequence_line": b"@r1\nAC+GT\nII\n",
        "blank_line_between": b"@r1\nACGT\n+\nIIII\n\n@r2\nAC\n+\nII\n",
    }
    out = []
    for name, data in texts.items():
        for gz in (False, True):
            if gz and name not in ("plain", "multiline", "empty_read"):
                continue
            path = os.path.join(tmp, name + (".gz" if gz else ".fastq"))
            if gz:
                import gzip
                with gzip.open(path, "wb") as f:
                    f.write(data)
            else:
                with open(path, "wb") as f:
                    f.write(data)
            case = {"name": name + ("_gz" if gz else ""), "gz": gz, "text_b64": base64.b64encode(data).decode()}
            try:
                seqs, offs = ref.parse_fastq(path)
                case["expect"] = {"reads": [bytes(seqs[int(offs[i]):int(offs[i + 1])]).decode("latin1") for i in range(len(offs) - 1)]}
            except OracleError as e:
                case["expect"] = {"error": str(e)}
            out.append(case)
    return out


def config1_fixture(ref: KaoriRef, tmp: str) -> None:
    """BASELINE.json configs[0] (countSingleBarcodes, 1 M x 75 bp, 1 k-barcode library, exact,
    forward strand) run through real kaori on the benchmark generator's own stream (screencounter_amd.synth.generate_host,
    byte-identical to the device generator).  The fixture holds the outputs and a digest of the inputs."""
    import hashlib
    from screencounter_amd import synth
    w = synth.workload(1)
    reads = synth.generate_host(w, w.n_reads)
    fq = os.path.join(tmp, "config1.fastq")
    synth.reads_to_fastq(fq, reads, w.read_len)
    counts, total = ref.count_single(fq, w.template, w.strand, w.pools[0], w.mismatches, w.use_first, 1)
    counts3, total3 = ref.count_single(fq, w.template, w.strand, w.pools[0], w.mismatches, w.use_first, 3)
    assert total == total3 and (counts == counts3).all()
    with open(os.path.join(OUT, "config1_kaori.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py::config1_fixture", "reference": "kaori v1.1.1 (screenCounter 1.5.1)",
                   "workload": w.describe(), "n_reads": w.n_reads, "read_len": w.read_len, "template": w.template,
                   "strand": w.strand, "mismatches": w.mismatches, "use_first": w.use_first,
                   "reads_sha256": hashlib.sha256(reads.tobytes()).hexdigest(),
                   "pool_sha256": hashlib.sha256("\n".join(w.pools[0]).encode()).hexdigest(),
                   "expect": {"counts": counts.tolist(), "total": int(total)}}, f)
    print(f"config 1: {int(counts.sum())} of {total} reads mapped")


def big_cases(ref: KaoriRef, tmp: str) -> None:
    """tests/golden/kaori_big.json: barcodes of 65..256 bases (as long as the longest template the reference compiles,
    src/count_single_barcodes.cpp:37-47) on every entry point; tests/golden/kaori_large_grid.json: countComboBarcodes with
    2 x 40 000 barcodes, inputs by seed (tests/gen.py::large_grid_case) + digest, outputs in full."""
    rng = random.Random(20261004)
    small = (1, 12, 40)
    out = []
    for _ in range(24):
        out.append(run_case(ref, gen.random_single_case(rng, max_vlen=240, sizes=small, min_vlen=65), tmp))
    for i in range(60):
        if i % 5 == 0:
            c = gen.random_combo_case(rng, sizes=small, wide="big")
        elif i % 5 == 1:
            c = gen.random_dual_case(rng, hazard_free=True, sizes=small, max_mm=3, wide="big")
        elif i % 5 == 2:
            c = gen.random_dual_case(rng, hazard_free=True, sizes=small, max_mm=2, wide="big")
            c["kind"] = "dual_diag"
        elif i % 5 == 3:
            c = gen.random_paired_combo_case(rng, sizes=small, max_mm=2, wide="big")
        else:
            c = gen.random_big_match_case(rng)
        out.append(run_case(ref, c, tmp))
    for i in range(30):
        out.append(run_case(ref, gen.random_dual_single_end_case(rng, sizes=small, wide="big", nreg=1 + i % 5), tmp))
    for i in range(16):
        c = gen.random_dual_single_end_case(rng, sizes=small, wide="big", diag=True)
        c["kind"] = "dual_single_end_diag"
        out.append(run_case(ref, c, tmp))
    with open(os.path.join(OUT, "kaori_big.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py::big_cases", "reference": "kaori v1.1.1 (screenCounter 1.5.1)", "seed": 20261004, "cases": out}, f)
    n_err = sum(1 for c in out if "error" in c["expect"])
    print(f"big keys: {len(out)} cases ({n_err} expected errors)")

    case = gen.large_grid_case()
    done = run_case(ref, case, tmp)
    with open(os.path.join(OUT, "kaori_large_grid.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py::big_cases", "reference": "kaori v1.1.1 (screenCounter 1.5.1)",
                   "inputs": "tests/gen.py::large_grid_case()", "inputs_sha256": gen.case_digest(case), "expect": done["expect"]}, f)
    print(f"large grid: {len(done['expect']['freq'])} combinations of {done['expect']['total']} reads")


def main() -> None:
    ref = KaoriRef()
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["big"]:                     # only the round-3 files (the others stay byte for byte)
        with tempfile.TemporaryDirectory() as tmp:
            big_cases(ref, tmp)
        return
    with tempfile.TemporaryDirectory() as tmp:
        big_cases(ref, tmp)
        ka = [run_case(ref, c, tmp) for c in known_answers()]
        for c in ka:
            check_r_expect(c)
        with open(os.path.join(OUT, "known_answers.json"), "w") as f:
            json.dump({"generator": "oracle/gen_golden.py", "reference": "kaori v1.1.1 (screenCounter 1.5.1)", "cases": ka}, f, indent=1)

        rng = random.Random(20261003)
        rnd = []
        small = (1, 12, 40)
        for _ in range(60):
            rnd.append(run_case(ref, gen.random_single_case(rng, max_vlen=20, sizes=small), tmp))
        for _ in range(40):
            rnd.append(run_case(ref, gen.random_combo_case(rng, sizes=small), tmp))
        for _ in range(50):
            rnd.append(run_case(ref, gen.random_dual_case(rng, hazard_free=True, sizes=small, max_mm=3), tmp))
        for _ in range(30):
            rnd.append(run_case(ref, gen.random_match_case(rng), tmp))
        for _ in range(40):
            c = gen.random_dual_case(rng, hazard_free=True, sizes=small, max_mm=2)
            c["kind"] = "dual_diag"
            rnd.append(run_case(ref, c, tmp))
        for _ in range(40):     # appended last: the cases above keep their seeds
            rnd.append(run_case(ref, gen.random_paired_combo_case(rng, sizes=small, max_mm=2), tmp))
        for i in range(40):
            rnd.append(run_case(ref, gen.random_dual_single_end_case(rng, sizes=small, wide=(i % 2 == 0)), tmp))
        for _ in range(20):     # single barcodes of 33..64 bases (wide keys)
            rnd.append(run_case(ref, gen.random_single_case(rng, max_vlen=64, sizes=small, min_vlen=33), tmp))
        for _ in range(40):
            rnd.append(run_case(ref, gen.random_random_barcode_case(rng, sizes=small), tmp))
        for i in range(40):
            c = gen.random_dual_single_end_case(rng, sizes=small, wide=(i % 3 == 0), diag=True)
            c["kind"] = "dual_single_end_diag"
            rnd.append(run_case(ref, c, tmp))
        for i in range(30):     # templates with 3 to 5 variable regions (DualBarcodesSingleEnd.hpp:144-163 takes any number)
            rnd.append(run_case(ref, gen.random_dual_single_end_case(rng, sizes=small, nreg=3 + i % 3), tmp))
        for i in range(60):     # barcodes of 33..64 bases on the combinatorial and paired-end paths (wide keys)
            if i % 4 == 0:
                c = gen.random_combo_case(rng, sizes=small, wide=True)
            elif i % 4 == 1:
                c = gen.random_dual_case(rng, hazard_free=True, sizes=small, max_mm=3, wide=True)
            elif i % 4 == 2:
                c = gen.random_dual_case(rng, hazard_free=True, sizes=small, max_mm=2, wide=True)
                c["kind"] = "dual_diag"
            else:
                c = gen.random_paired_combo_case(rng, sizes=small, max_mm=2, wide=True)
            rnd.append(run_case(ref, c, tmp))
        with open(os.path.join(OUT, "kaori_random.json"), "w") as f:
            json.dump({"generator": "oracle/gen_golden.py", "reference": "kaori v1.1.1 (screenCounter 1.5.1)", "seed": 20261003, "cases": rnd}, f)

        fq = fastq_cases(ref, tmp)
        with open(os.path.join(OUT, "fastq_cases.json"), "w") as f:
            json.dump({"generator": "oracle/gen_golden.py", "reference": "kaori::FastqReader v1.1.1 over byteme v1.0.1", "cases": fq}, f, indent=1)
        config1_fixture(ref, tmp)
    n_err = sum(1 for c in rnd if "error" in c["expect"])
    print(f"known answers: {len(ka)}  random: {len(rnd)} ({n_err} expected errors)  fastq: {len(fq)}")
    for c in fq:
        print("  fastq", c["name"], c["expect"])


if __name__ == "__main__":
    main()
