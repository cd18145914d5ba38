#!/usr/bin/env python3
"""Randomised cross-check: C restatement (liboracle.so) vs the real kaori (_ref/libkaori_ref.so).

Runs only where oracle/_ref was built (this container).  Not a pytest file: it is the
tool used to pin the restatement before trusting it; tests/test_oracle_golden.py replays a
committed subset (tests/golden/) without needing the reference.

    python oracle/fuzz_vs_kaori.py [--iters 300] [--seed 1]
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyoracle import KaoriRef, Oracle, OracleError, write_fastq  # noqa: E402

from tests.gen import (BASES, concrete, fill_template, make_pool, make_reads, make_template, mutate,  # noqa: E402
                       rand_seq, rc)


def fuzz_single(rng, ora, ref, tmpdir, it):
    vlen = rng.choice([3, 4, 6, 8, 10, 20, 33])
    alphabet = rng.choice(["AC", "ACG", BASES, BASES])
    npool = rng.choice([1, 2, 5, 20, 100])
    pool = make_pool(rng, npool, vlen, alphabet, min_dist=1, iupac_rate=rng.choice([0, 0, 0.05]))
    template = make_template(rng, 1, [vlen], rng.choice([0, 1, 3]), rng.choice([4, 8, 12, 40]))
    strand = rng.choice([0, 1, 2])
    mm = rng.choice([0, 1, 1, 2, 3])
    first = rng.random() < 0.5
    reads = make_reads(rng, template, [pool], rng.choice([1, 30, 200]), strand, rng.choice([0, 0.02, 0.08]), rng.choice([0, 0.01, 0.05]),
                       rng.choice([0, 0.3]), 0.1, rng.choice([0, 5, 30]))
    fq = os.path.join(tmpdir, f"s{it}.fastq")
    write_fastq(fq, reads, gz=rng.random() < 0.2)
    try:
        exp = ref.count_single(fq, template, strand, pool, mm, first, nthreads=rng.choice([1, 3]))
    except OracleError as e:
        try:
            ora.count_single(reads, template, strand, pool, mm, first)
        except OracleError:
            return "both-error"
        raise AssertionError(f"kaori errored ({e}) but oracle did not: {template} {pool}")
    got = ora.count_single(reads, template, strand, pool, mm, first)
    assert got[1] == exp[1], (got[1], exp[1])
    if not np.array_equal(got[0], exp[0]):
        raise AssertionError(f"single mismatch: tmpl={template} strand={strand} mm={mm} first={first}\npool={pool}\nreads={reads}\nexp={exp[0]}\ngot={got[0]}")
    return "ok"


def fuzz_combo(rng, ora, ref, tmpdir, it):
    v0, v1 = rng.choice([3, 5, 8, 14]), rng.choice([3, 6, 14])
    alphabet = rng.choice(["AC", BASES, BASES])
    p0 = make_pool(rng, rng.choice([1, 4, 30]), v0, alphabet, iupac_rate=rng.choice([0, 0, 0.05]))
    p1 = make_pool(rng, rng.choice([1, 4, 30]), v1, alphabet, iupac_rate=rng.choice([0, 0, 0.05]))
    template = make_template(rng, 2, [v0, v1], rng.choice([0, 1, 3]), rng.choice([4, 8, 12]))
    strand = rng.choice([0, 1, 2])
    mm = rng.choice([0, 1, 2, 3])
    first = rng.random() < 0.5
    reads = make_reads(rng, template, [p0, p1], rng.choice([1, 30, 200]), strand, rng.choice([0, 0.03, 0.08]), rng.choice([0, 0.02]),
                       rng.choice([0, 0.3]), 0.1, rng.choice([0, 5, 30]))
    fq = os.path.join(tmpdir, f"c{it}.fastq")
    write_fastq(fq, reads)
    try:
        exp = ref.count_combo(fq, template, strand, p0, p1, mm, first, nthreads=rng.choice([1, 3]))
    except OracleError as e:
        try:
            ora.count_combo(reads, template, strand, p0, p1, mm, first)
        except OracleError:
            return "both-error"
        raise AssertionError(f"kaori errored ({e}) but oracle did not")
    got = ora.count_combo(reads, template, strand, p0, p1, mm, first)
    ok = got[2] == exp[2] and np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
    if not ok:
        raise AssertionError(f"combo mismatch: tmpl={template} strand={strand} mm={mm} first={first}\np0={p0}\np1={p1}\nreads={reads}\nexp={exp}\ngot={got}")
    return "ok"


def fuzz_dual(rng, ora, ref, tmpdir, it, hazard_free=True):
    l1, l2 = rng.choice([4, 6, 9, 12]), rng.choice([4, 7, 12])
    mm1, mm2 = rng.choice([0, 1, 2]), rng.choice([0, 1, 2])
    d1 = 2 * mm1 + 1 if hazard_free else 1
    d2 = 2 * mm2 + 1 if hazard_free else 1
    u1 = make_pool(rng, rng.choice([1, 3, 8]), l1, BASES, min_dist=d1)
    u2 = make_pool(rng, rng.choice([1, 3, 8]), l2, BASES, min_dist=d2)
    allpairs = [(a, b) for a in u1 for b in u2]
    rng.shuffle(allpairs)
    pairs = allpairs[: rng.randint(1, len(allpairs))]
    pool1 = [a for a, _ in pairs]
    pool2 = [b for _, b in pairs]
    t1 = make_template(rng, 1, [l1], rng.choice([0, 2, 4]), rng.choice([4, 8]))
    t2 = make_template(rng, 1, [l2], rng.choice([0, 2, 4]), rng.choice([4, 8]))
    rev1, rev2 = rng.random() < 0.3, rng.random() < 0.3
    randomized = rng.random() < 0.4
    first = rng.random() < 0.5
    n = rng.choice([1, 30, 150])
    r1s, r2s = [], []
    p_sub, p_n = rng.choice([0, 0.03, 0.08]), rng.choice([0, 0.02])
    for _ in range(n):
        u = rng.random()
        if u < 0.1:
            a, b = rand_seq(rng, rng.randint(0, 30)), rand_seq(rng, rng.randint(0, 30))
        else:
            if u < 0.8:
                x, y = rng.choice(pairs)
            else:
                x, y = rng.choice(u1), rng.choice(u2)
            a = fill_template(t1, [x])
            b = fill_template(t2, [y])
            a = mutate(rng, a, p_sub, p_n, 0.05)
            b = mutate(rng, b, p_sub, p_n, 0.05)
            pad = rng.choice([0, 4, 20])
            a = rand_seq(rng, rng.randint(0, pad)) + a + rand_seq(rng, rng.randint(0, pad))
            b = rand_seq(rng, rng.randint(0, pad)) + b + rand_seq(rng, rng.randint(0, pad))
            if rng.random() < 0.1:
                x2, _ = rng.choice(pairs)
                a += mutate(rng, fill_template(t1, [x2]), p_sub, p_n, 0.0)
            if rev1:
                a = rc(a)
            if rev2:
                b = rc(b)
            if randomized and rng.random() < 0.5:
                a, b = b, a
        r1s.append(a)
        r2s.append(b)
    fq1 = os.path.join(tmpdir, f"d{it}_1.fastq")
    fq2 = os.path.join(tmpdir, f"d{it}_2.fastq")
    write_fastq(fq1, r1s)
    write_fastq(fq2, r2s)
    exp = ref.count_dual(fq1, t1, rev1, mm1, pool1, fq2, t2, rev2, mm2, pool2, randomized, first, nthreads=1)
    got = ora.count_dual(r1s, r2s, t1, rev1, mm1, pool1, t2, rev2, mm2, pool2, randomized, first)
    ok = got[1] == exp[1] and np.array_equal(got[0], exp[0])
    if not ok:
        if not hazard_free:
            return "hazard-diff"
        raise AssertionError(f"dual mismatch: t1={t1} t2={t2} rev=({rev1},{rev2}) mm=({mm1},{mm2}) rand={randomized} first={first}\n"
                             f"pool1={pool1}\npool2={pool2}\nr1={r1s}\nr2={r2s}\nexp={exp}\ngot={got}")
    return "ok"


def fuzz_dual_diag(rng, ora, ref, tmpdir, it):
    from tests import gen
    c = gen.random_dual_case(rng, hazard_free=True, max_mm=2)
    # the diagnostics path searches each pool on its own with DuplicateAction::FIRST: feed it
    # pools with repeated barcodes (pairs stay unique) every now and then -- that is what
    # random_dual_case produces whenever a barcode takes part in several pairs.
    fq1 = os.path.join(tmpdir, f"g{it}_1.fastq")
    fq2 = os.path.join(tmpdir, f"g{it}_2.fastq")
    write_fastq(fq1, c["reads1"])
    write_fastq(fq2, c["reads2"])
    exp = ref.count_dual_diag(fq1, c["template1"], c["reverse1"], c["mismatches1"], c["pool1"],
                              fq2, c["template2"], c["reverse2"], c["mismatches2"], c["pool2"], c["randomized"], c["use_first"], 1)
    got = ora.count_dual_diag(c["reads1"], c["reads2"], c["template1"], c["reverse1"], c["mismatches1"], c["pool1"],
                              c["template2"], c["reverse2"], c["mismatches2"], c["pool2"], c["randomized"], c["use_first"])
    for key in exp:
        if not np.array_equal(np.asarray(exp[key]), np.asarray(got[key])):
            raise AssertionError(f"dual-diag mismatch in {key}: {c}\nexp={exp}\ngot={got}")
    return "ok"


def fuzz_combo_paired(rng, ora, ref, tmpdir, it):
    from tests import gen
    c = gen.random_paired_combo_case(rng)
    fq1 = os.path.join(tmpdir, f"p{it}_1.fastq")
    fq2 = os.path.join(tmpdir, f"p{it}_2.fastq")
    write_fastq(fq1, c["reads1"])
    write_fastq(fq2, c["reads2"])
    exp = ref.count_combo_paired(fq1, c["template1"], c["reverse1"], c["mismatches1"], c["pool1"],
                                 fq2, c["template2"], c["reverse2"], c["mismatches2"], c["pool2"], c["randomized"], c["use_first"], 1)
    got = ora.count_combo_paired(c["reads1"], c["reads2"], c["template1"], c["reverse1"], c["mismatches1"], c["pool1"],
                                 c["template2"], c["reverse2"], c["mismatches2"], c["pool2"], c["randomized"], c["use_first"])
    for key in exp:
        if not np.array_equal(np.asarray(exp[key]), np.asarray(got[key])):
            raise AssertionError(f"combo-paired mismatch in {key}: {c}\nexp={exp}\ngot={got}")
    return "ok"


def fuzz_dual_single_end(rng, ora, ref, tmpdir, it):
    from tests import gen
    c = gen.random_dual_single_end_case(rng)
    fq = os.path.join(tmpdir, f"e{it}.fastq")
    write_fastq(fq, c["reads"])
    try:
        exp = ref.count_dual_single_end(fq, c["template"], c["strand"], c["pools"], c["mismatches"], c["use_first"], 1)
    except OracleError:
        try:
            ora.count_dual_single_end(c["reads"], c["template"], c["strand"], c["pools"], c["mismatches"], c["use_first"])
        except OracleError:
            return "both-error"
        raise
    got = ora.count_dual_single_end(c["reads"], c["template"], c["strand"], c["pools"], c["mismatches"], c["use_first"])
    if not (got[1] == exp[1] and np.array_equal(got[0], exp[0])):
        raise AssertionError(f"dual-single-end mismatch: {c}\nexp={exp}\ngot={got}")
    return "ok"


def fuzz_dual_single_end_diag(rng, ora, ref, tmpdir, it):
    from tests import gen
    c = gen.random_dual_single_end_case(rng, wide=rng.random() < 0.3, diag=True)
    fq = os.path.join(tmpdir, f"x{it}.fastq")
    write_fastq(fq, c["reads"])
    try:
        exp = ref.count_dual_single_end_diag(fq, c["template"], c["strand"], c["pools"], c["mismatches"], c["use_first"], 1)
    except OracleError:
        try:
            ora.count_dual_single_end_diag(c["reads"], c["template"], c["strand"], c["pools"], c["mismatches"], c["use_first"])
        except OracleError:
            return "both-error"
        raise
    got = ora.count_dual_single_end_diag(c["reads"], c["template"], c["strand"], c["pools"], c["mismatches"], c["use_first"])
    for key in exp:
        if not np.array_equal(np.asarray(exp[key]), np.asarray(got[key])):
            raise AssertionError(f"dual-single-end-diag mismatch in {key}: {c}\nexp={exp}\ngot={got}")
    return "ok"


def fuzz_random(rng, ora, ref, tmpdir, it):
    from tests import gen
    c = gen.random_random_barcode_case(rng)
    fq = os.path.join(tmpdir, f"r{it}.fastq")
    write_fastq(fq, c["reads"])
    try:
        exp = ref.count_random(fq, c["template"], c["strand"], c["mismatches"], c["use_first"], 1)
    except OracleError:
        try:
            ora.count_random(c["reads"], c["template"], c["strand"], c["mismatches"], c["use_first"])
        except OracleError:
            return "both-error"
        raise
    got = ora.count_random(c["reads"], c["template"], c["strand"], c["mismatches"], c["use_first"])
    if got != exp:
        raise AssertionError(f"random mismatch: {c}\nexp={exp}\ngot={got}")
    return "ok"


def fuzz_match(rng, ora, ref):
    vlen = rng.choice([3, 5, 8, 12])
    alphabet = rng.choice(["AC", BASES])
    pool = make_pool(rng, rng.choice([1, 4, 30]), vlen, alphabet, iupac_rate=rng.choice([0, 0.1]))
    seqs = []
    for _ in range(40):
        s = concrete(rng, rng.choice(pool))
        seqs.append(mutate(rng, s, 0.15, 0.03, 0.1))
    subs = rng.choice([0, 1, 2, 3])
    rev = rng.random() < 0.5
    try:
        exp = ref.match_barcodes(seqs, pool, subs, rev)
    except OracleError:
        try:
            ora.match_barcodes(seqs, pool, subs, rev)
        except OracleError:
            return "both-error"
        raise
    got = ora.match_barcodes(seqs, pool, subs, rev)
    if not (np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])):
        raise AssertionError(f"match mismatch pool={pool} seqs={seqs} subs={subs} rev={rev}\nexp={exp}\ngot={got}")
    return "ok"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    ora, ref = Oracle(), KaoriRef()
    rng = random.Random(args.seed)
    tally: dict[str, int] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for it in range(args.iters):
            for name, fn in (("single", lambda: fuzz_single(rng, ora, ref, tmp, it)),
                             ("combo", lambda: fuzz_combo(rng, ora, ref, tmp, it)),
                             ("dual", lambda: fuzz_dual(rng, ora, ref, tmp, it, True)),
                             ("dual-hazard", lambda: fuzz_dual(rng, ora, ref, tmp, it, False)),
                             ("dual-diag", lambda: fuzz_dual_diag(rng, ora, ref, tmp, it)),
                             ("combo-paired", lambda: fuzz_combo_paired(rng, ora, ref, tmp, it)),
                             ("dual-single-end", lambda: fuzz_dual_single_end(rng, ora, ref, tmp, it)),
                             ("random", lambda: fuzz_random(rng, ora, ref, tmp, it)),
                             ("dual-single-end-diag", lambda: fuzz_dual_single_end_diag(rng, ora, ref, tmp, it)),
                             ("match", lambda: fuzz_match(rng, ora, ref))):
                res = fn()
                tally[f"{name}:{res}"] = tally.get(f"{name}:{res}", 0) + 1
            for f in os.listdir(tmp):
                os.unlink(os.path.join(tmp, f))
    for k in sorted(tally):
        print(f"{k:28s} {tally[k]}")


if __name__ == "__main__":
    main()
