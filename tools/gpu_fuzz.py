#!/usr/bin/env python3
"""Time-boxed random parity fuzz on the GPU: every entry point against the oracle on fresh seeds, both engines.
usage: python3 tools/gpu_fuzz.py [seconds] [first_seed]      (prints a progress line every few seconds)"""
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import screencounter_amd as sc  # noqa: E402
from oracle.pyoracle import Oracle, OracleError, write_fastq  # noqa: E402
from tests import gen  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
ora = Oracle()
dev = "cuda:0"
tally = {}


def same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


def one(kind, rng, tmp):
    if kind == "single":
        c = gen.random_single_case(rng, max_vlen=64)
        exp = ora.count_single(c["reads"], c["template"], c["strand"], c["pool"], c["mismatches"], c["use_first"])
        s, o = sc.upload_reads(c["reads"], dev)
        with sc.Plan.single(c["template"], c["strand"], c["pool"], c["mismatches"], c["use_first"]) as p:
            p.count(s, o)
            got = p.read()
        assert got[1] == exp[1] and same(got[0], exp[0]), c
    elif kind == "combo":
        c = gen.random_combo_case(rng)
        exp = ora.count_combo(c["reads"], c["template"], c["strand"], c["pool0"], c["pool1"], c["mismatches"], c["use_first"])
        s, o = sc.upload_reads(c["reads"], dev)
        with sc.Plan.combo(c["template"], c["strand"], c["pool0"], c["pool1"], c["mismatches"], c["use_first"]) as p:
            p.count(s, o)
            got = p.read_combo()
        assert got[2] == exp[2] and same(got[0], exp[0]) and same(got[1], exp[1]), c
    elif kind in ("dual", "dual_diag", "paired_combo"):
        c = gen.random_paired_combo_case(rng) if kind == "paired_combo" else gen.random_dual_case(rng, hazard_free=True, max_mm=2)
        s1, o1 = sc.upload_reads(c["reads1"], dev)
        s2, o2 = sc.upload_reads(c["reads2"], dev)
        args = (c["template1"], c["reverse1"], c["mismatches1"], c["pool1"], c["template2"], c["reverse2"], c["mismatches2"], c["pool2"],
                c["randomized"], c["use_first"])
        if kind == "dual":
            exp = ora.count_dual(c["reads1"], c["reads2"], *args)
            with sc.Plan.dual(*args) as p:
                p.count_paired(s1, s2, o1, o2)
                got = p.read()
            assert got[1] == exp[1] and same(got[0], exp[0]), c
        else:
            exp = (ora.count_dual_diag if kind == "dual_diag" else ora.count_combo_paired)(c["reads1"], c["reads2"], *args)
            with (sc.Plan.dual(*args, diagnostics=True) if kind == "dual_diag" else sc.Plan.paired_combo(*args)) as p:
                p.count_paired(s1, s2, o1, o2)
                got = p.read_diagnostics()
            for k in exp:
                assert same(exp[k], got[k]), (k, c)
    elif kind in ("dual_se", "dual_se_diag"):
        diag = kind == "dual_se_diag"
        c = gen.random_dual_single_end_case(rng, diag=diag)
        fq = os.path.join(tmp, "f.fastq")
        write_fastq(fq, c["reads"])
        if diag:
            exp = ora.count_dual_single_end_diag(c["reads"], c["template"], c["strand"], c["pools"], c["mismatches"], c["use_first"])
            counts, (idx, freq), total = sc.count_dual_barcodes_single_end(fq, c["template"], c["pools"], c["strand"], c["mismatches"], c["use_first"], True, 1)
            assert total == exp["total"] and same(counts, exp["counts"]) and same(idx, exp["indices"]) and same(freq, exp["freq"]), c
        else:
            exp = ora.count_dual_single_end(c["reads"], c["template"], c["strand"], c["pools"], c["mismatches"], c["use_first"])
            got = sc.count_dual_barcodes_single_end(fq, c["template"], c["pools"], c["strand"], c["mismatches"], c["use_first"], False, 1)
            assert got[1] == exp[1] and same(got[0], exp[0]), c
    elif kind == "random":
        c = gen.random_random_barcode_case(rng)
        fq = os.path.join(tmp, "r.fastq")
        write_fastq(fq, c["reads"])
        exp = ora.count_random(c["reads"], c["template"], c["strand"], c["mismatches"], c["use_first"])
        (seqs, freq), total = sc.count_random_barcodes(fq, c["template"], c["strand"], c["mismatches"], c["use_first"], 1)
        assert total == exp[1] and dict(zip(seqs, freq.tolist())) == exp[0], c
    else:
        c = gen.random_match_case(rng)
        exp = ora.match_barcodes(c["sequences"], c["choices"], c["substitutions"], c["reverse"])
        got = sc.match_barcodes(c["sequences"], c["choices"], c["substitutions"], c["reverse"])
        assert same(got[0], exp[0]) and same(got[1], exp[1]), c


kinds = ["single", "combo", "dual", "dual_diag", "paired_combo", "dual_se", "dual_se_diag", "random", "match"]
t0 = last = time.time()
it = 0
with tempfile.TemporaryDirectory() as tmp:
    while time.time() - t0 < budget:
        rng = random.Random(seed0 + it)
        kind = kinds[it % len(kinds)]
        for engine in ("0", "1"):
            os.environ["SCG_FORCE_GENERAL"] = engine
            os.environ["SCG_TALLY"] = "1" if (it // len(kinds)) % 2 else ""
            if not os.environ["SCG_TALLY"]:
                del os.environ["SCG_TALLY"]
            state = rng.getstate()
            try:
                one(kind, rng, tmp)
                tally[kind] = tally.get(kind, 0) + 1
            except (OracleError, sc.ScgError):
                tally[kind + ":error"] = tally.get(kind + ":error", 0) + 1
            rng.setstate(state)
        it += 1
        if time.time() - last > 5:
            last = time.time()
            print(f"{last - t0:6.0f}s  iterations {it}  {tally}", flush=True)
print("done", it, tally)
