"""HIP path vs the oracle on seeded random workloads, through the C ABI (batch-level plans).

Bit-exact comparison: every count, every combination, every total.
"""
import random

import numpy as np
import pytest

from tests import gen

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["staged", "general"], autouse=True)
def engine(request, monkeypatch):
    """Every parity test runs on the LDS-staged kernels and on the byte-wise general engine."""
    if request.param == "general":
        monkeypatch.setenv("SCG_FORCE_GENERAL", "1")
    else:
        monkeypatch.delenv("SCG_FORCE_GENERAL", raising=False)
    return request.param


def _unsupported(case):
    lens = [len(p[0]) for k, p in case.items() if k.startswith("pool") and p]
    return any(n > 32 for n in lens)


def run_single(sc, case, device):
    seqs, offs = sc.upload_reads(case["reads"], device)
    with sc.Plan.single(case["template"], case["strand"], case["pool"], case["mismatches"], case["use_first"]) as plan:
        plan.count(seqs, offs)
        return plan.read()


def run_combo(sc, case, device):
    seqs, offs = sc.upload_reads(case["reads"], device)
    with sc.Plan.combo(case["template"], case["strand"], case["pool0"], case["pool1"], case["mismatches"], case["use_first"]) as plan:
        plan.count(seqs, offs)
        return plan.read_combo()


def run_dual(sc, case, device):
    s1, o1 = sc.upload_reads(case["reads1"], device)
    s2, o2 = sc.upload_reads(case["reads2"], device)
    with sc.Plan.dual(case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                      case["template2"], case["reverse2"], case["mismatches2"], case["pool2"],
                      case["randomized"], case["use_first"]) as plan:
        plan.count_paired(s1, s2, o1, o2)
        return plan.read()


@pytest.mark.parametrize("seed", range(8))
def test_single_random(sc, oracle, gpu, seed):
    from oracle.pyoracle import OracleError
    rng = random.Random(1000 + seed)
    done = 0
    while done < 25:
        case = gen.random_single_case(rng, max_vlen=20)
        try:
            exp = oracle.count_single(case["reads"], case["template"], case["strand"], case["pool"], case["mismatches"], case["use_first"])
        except OracleError:
            with pytest.raises(sc.ScgError):
                run_single(sc, case, gpu)
            continue
        got = run_single(sc, case, gpu)
        assert got[1] == exp[1]
        assert np.array_equal(got[0], exp[0]), case
        done += 1


@pytest.mark.parametrize("wide", [False, True, "big"], ids=["narrow", "wide", "big"])      # wide: pools of 33..64 bases
@pytest.mark.parametrize("seed", range(6))
def test_combo_random(sc, oracle, gpu, seed, wide):
    from oracle.pyoracle import OracleError
    rng = random.Random(2000 + seed)
    done = 0
    while done < (8 if wide else 20):
        case = gen.random_combo_case(rng, wide=wide)
        try:
            exp = oracle.count_combo(case["reads"], case["template"], case["strand"], case["pool0"], case["pool1"], case["mismatches"], case["use_first"])
        except OracleError:
            with pytest.raises(sc.ScgError):
                run_combo(sc, case, gpu)
            continue
        got = run_combo(sc, case, gpu)
        assert got[2] == exp[2]
        assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]), case
        done += 1


@pytest.mark.parametrize("wide", [False, True, "big"], ids=["narrow", "wide", "big"])      # wide: barcodes of 33..64 bases on a mate
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("hazard_free", [True, False])
def test_dual_random(sc, oracle, gpu, seed, hazard_free, wide):
    # Both are compared with the cache-free oracle: hazard_free only matters when
    # comparing with the reference itself.
    rng = random.Random(3000 + seed)
    for _ in range(6 if wide else 20):
        case = gen.random_dual_case(rng, hazard_free=hazard_free, max_mm=3, wide=wide)
        exp = oracle.count_dual(case["reads1"], case["reads2"], case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                                case["template2"], case["reverse2"], case["mismatches2"], case["pool2"], case["randomized"], case["use_first"])
        got = run_dual(sc, case, gpu)
        assert got[1] == exp[1]
        assert np.array_equal(got[0], exp[0]), case


@pytest.mark.parametrize("seed", range(4))
def test_match_random(sc, oracle, gpu, seed):
    from oracle.pyoracle import OracleError
    rng = random.Random(4000 + seed)
    for _ in range(30):
        case = gen.random_match_case(rng)
        try:
            exp = oracle.match_barcodes(case["sequences"], case["choices"], case["substitutions"], case["reverse"])
        except OracleError:
            with pytest.raises(sc.ScgError):
                sc.match_barcodes(case["sequences"], case["choices"], case["substitutions"], case["reverse"])
            continue
        got = sc.match_barcodes(case["sequences"], case["choices"], case["substitutions"], case["reverse"])
        assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]), case


def test_long_and_mixed_reads(sc, oracle, gpu):
    """Reads beyond the staged tile rows (> 320 bases) route the batch to the byte-wise kernels."""
    rng = random.Random(5150)
    pool = gen.make_pool(rng, 50, 12, gen.BASES)
    template = "ACGTTGCA" + "-" * 12 + "GGATCCAA"
    reads = []
    for i in range(300):
        core = gen.mutate(rng, gen.fill_template(template, [rng.choice(pool)]), 0.02, 0.01, 0.05)
        pad = rng.choice([10, 100, 400, 900]) if i % 3 == 0 else rng.choice([0, 20, 120])
        r = gen.rand_seq(rng, rng.randint(0, pad)) + core + gen.rand_seq(rng, rng.randint(0, pad))
        reads.append(gen.rc(r) if rng.random() < 0.5 else r)
    assert max(len(r) for r in reads) > 320
    exp = oracle.count_single(reads, template, 2, pool, 1, False)
    case = dict(reads=reads, template=template, strand=2, pool=pool, mismatches=1, use_first=False)
    got = run_single(sc, case, gpu)
    assert got[1] == exp[1] and np.array_equal(got[0], exp[0])
    # wide tile (161..320 bases) on the staged path
    mid = [r for r in reads if len(r) <= 320]
    assert max(len(r) for r in mid) > 160
    exp = oracle.count_single(mid, template, 2, pool, 1, True)
    case = dict(reads=mid, template=template, strand=2, pool=pool, mismatches=1, use_first=True)
    got = run_single(sc, case, gpu)
    assert got[1] == exp[1] and np.array_equal(got[0], exp[0])


def test_wrong_max_len_hint_is_reported(sc, gpu, engine):
    if engine == "general":
        pytest.skip("the general kernels do not use the hint")
    reads = ["ACGT" + "A" * 200 + "TGCA"] * 10
    seqs, offs = sc.upload_reads(reads, gpu)
    with sc.Plan.single("ACGT----TGCA", 0, ["AAAA"], 0, True) as plan:
        plan.count(seqs, offs, max_len=100)     # a lie: the reads are 208 bases long
        with pytest.raises(sc.ScgError, match="longer than the max_len"):
            plan.read()


def test_template_longer_than_64(sc, oracle, gpu):
    """NT = 4 / 8 tiles: templates of 100 and 200 bases."""
    rng = random.Random(99)
    for flank in (40, 90):
        pool = gen.make_pool(rng, 30, 20, gen.BASES)
        template = gen.rand_seq(rng, flank) + "-" * 20 + gen.rand_seq(rng, flank)
        reads = []
        for _ in range(200):
            core = gen.mutate(rng, gen.fill_template(template, [rng.choice(pool)]), 0.01, 0.005, 0.0)
            r = gen.rand_seq(rng, rng.randint(0, 40)) + core + gen.rand_seq(rng, rng.randint(0, 40))
            reads.append(gen.rc(r) if rng.random() < 0.5 else r)
        for mm in (0, 2):
            exp = oracle.count_single(reads, template, 2, pool, mm, True)
            got = run_single(sc, dict(reads=reads, template=template, strand=2, pool=pool, mismatches=mm, use_first=True), gpu)
            assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), (flank, mm)


def test_tally_mode_matches_atomics(sc, oracle, gpu, monkeypatch):
    """The index-stream + LDS-histogram counting path (ScgCounters::unit_index, tally_kernel) against the
    atomic path and the oracle: random cases, a hot barcode that crosses the 16-bit flush threshold
    several times, and a library wide enough for two tally passes."""
    import torch
    rng = random.Random(4242)
    monkeypatch.setenv("SCG_TALLY", "1")
    for _ in range(25):
        case = gen.random_single_case(rng, max_vlen=20)
        try:
            exp = oracle.count_single(case["reads"], case["template"], case["strand"], case["pool"], case["mismatches"], case["use_first"])
        except Exception:
            continue
        got = run_single(sc, case, gpu)
        assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), (case, exp, got)
    for _ in range(15):
        case = gen.random_dual_case(rng, hazard_free=True)
        exp = oracle.count_dual(case["reads1"], case["reads2"], case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                                case["template2"], case["reverse2"], case["mismatches2"], case["pool2"], case["randomized"], case["use_first"])
        got = run_dual(sc, case, gpu)
        assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), (case, exp, got)
    for _ in range(15):
        case = gen.random_combo_case(rng)
        try:
            exp = oracle.count_combo(case["reads"], case["template"], case["strand"], case["pool0"], case["pool1"], case["mismatches"], case["use_first"])
        except Exception:
            continue
        got = run_combo(sc, case, gpu)
        assert got[2] == exp[2] and np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]), (case, exp, got)
    # hot bin: 150 000 copies of one construct (> 4 x 0x8000) plus a few others, two launches accumulate
    template = "ACGTAC" + "-" * 10 + "TGCATG"
    pool = gen.make_pool(rng, 40, 10, "ACGT")
    reads = [gen.fill_template(template, [pool[7]])] * 150000 + [gen.fill_template(template, [rng.choice(pool)]) for _ in range(5000)]
    seqs, offs = sc.upload_reads(reads, gpu)
    with sc.Plan.single(template, 0, pool, 1, True) as plan:
        plan.count(seqs, offs)
        plan.count(seqs, offs)
        counts, total = plan.read()
    exp = oracle.count_single(reads, template, 0, pool, 1, True)
    assert total == 2 * exp[1] and np.array_equal(counts, 2 * exp[0]) and counts[7] >= 300000
    # two passes: 100 000 barcodes (> 80 K bins), fixed-length batch generated on the device
    from screencounter_amd import synth
    w = synth.workload(2, n_reads=300000)
    dw = synth.DeviceWorkload(w, gpu)
    dev = dw.generate(300000)
    results = []
    for mode in ("1", "0"):
        monkeypatch.setenv("SCG_TALLY", mode)
        with dw.plan() as plan:
            plan.count(dev, fixed_len=w.read_len, n_reads=300000)
            results.append(plan.read())
    assert results[0][1] == results[1][1] == 300000 and np.array_equal(results[0][0], results[1][0]) and results[0][0].sum() > 200000


@pytest.mark.parametrize("seed", range(4))
def test_random_barcodes_random(sc, oracle, gpu, seed, tmp_path):
    """countRandomBarcodes through the file-level entry point (the tally lives on the host), plain and
    multi-threaded staging, against the oracle."""
    from oracle.pyoracle import OracleError, write_fastq
    rng = random.Random(7100 + seed)
    for it in range(15):
        case = gen.random_random_barcode_case(rng, sizes=(1, 40, 400))
        fq = str(tmp_path / f"r{it}.fastq")
        write_fastq(fq, case["reads"])
        try:
            exp = oracle.count_random(case["reads"], case["template"], case["strand"], case["mismatches"], case["use_first"])
        except OracleError:
            with pytest.raises(sc.ScgError):
                sc.count_random_barcodes(fq, case["template"], case["strand"], case["mismatches"], case["use_first"], 1)
            continue
        for threads in (1, 4):
            (seqs, freq), total = sc.count_random_barcodes(fq, case["template"], case["strand"], case["mismatches"], case["use_first"], threads)
            assert total == exp[1] and dict(zip(seqs, freq.tolist())) == exp[0] and seqs == sorted(seqs), (case, exp, seqs, freq)


@pytest.mark.parametrize("seed", range(5))
def test_dual_single_end_random(sc, oracle, gpu, seed):
    """countDualBarcodesSingleEnd: one or two regions concatenated into one (possibly > 32 bp) key."""
    rng = random.Random(6800 + seed)
    for _ in range(20):
        case = gen.random_dual_single_end_case(rng)
        exp = oracle.count_dual_single_end(case["reads"], case["template"], case["strand"], case["pools"], case["mismatches"], case["use_first"])
        seqs, offs = sc.upload_reads(case["reads"], gpu)
        with sc.Plan.dual_single_end(case["template"], case["strand"], case["pools"], case["mismatches"], case["use_first"]) as plan:
            plan.count(seqs, offs)
            got = plan.read()
        assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), (case, exp, got)


@pytest.mark.parametrize("seed", range(4))
def test_dual_single_end_diagnostics_random(sc, oracle, gpu, seed, tmp_path):
    """countDualBarcodesSingleEnd(include.invalid=TRUE): valid counts + invalid (pool1, pool2) combinations."""
    from oracle.pyoracle import OracleError, write_fastq
    rng = random.Random(7300 + seed)
    for it in range(15):
        case = gen.random_dual_single_end_case(rng, wide=rng.random() < 0.3, diag=True)
        fq = str(tmp_path / f"x{it}.fastq")
        write_fastq(fq, case["reads"])
        try:
            exp = oracle.count_dual_single_end_diag(case["reads"], case["template"], case["strand"], case["pools"], case["mismatches"], case["use_first"])
        except OracleError:
            with pytest.raises(sc.ScgError):
                sc.count_dual_barcodes_single_end(fq, case["template"], case["pools"], case["strand"], case["mismatches"], case["use_first"], True, 1)
            continue
        counts, (idx, freq), total = sc.count_dual_barcodes_single_end(fq, case["template"], case["pools"], case["strand"],
                                                                      case["mismatches"], case["use_first"], True, 1)
        assert total == exp["total"] and np.array_equal(counts, exp["counts"]), (case, exp, counts)
        assert np.array_equal(idx, exp["indices"]) and np.array_equal(freq, exp["freq"]), (case, exp, idx, freq)


@pytest.mark.parametrize("seed", range(3))
def test_big_keys_random(sc, oracle, gpu, seed):
    """Barcodes of 65..256 bases (2 x 256-bit planes; the general kernels): countSingleBarcodes, matchBarcodes and
    countDualBarcodesSingleEnd with combined keys of that size, with and without include.invalid."""
    rng = random.Random(7300 + seed)
    for _ in range(10):
        case = gen.random_single_case(rng, max_vlen=240, min_vlen=65)
        exp = oracle.count_single(case["reads"], case["template"], case["strand"], case["pool"], case["mismatches"], case["use_first"])
        got = run_single(sc, case, gpu)
        assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), (case, exp, got)
    for _ in range(8):
        c = gen.random_big_match_case(rng)
        exp = oracle.match_barcodes(c["sequences"], c["choices"], c["substitutions"], c["reverse"])
        got = sc.match_barcodes(c["sequences"], c["choices"], c["substitutions"], c["reverse"])
        assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]), c
    for i in range(10):
        case = gen.random_dual_single_end_case(rng, wide="big", nreg=1 + i % 5)
        exp = oracle.count_dual_single_end(case["reads"], case["template"], case["strand"], case["pools"], case["mismatches"], case["use_first"])
        seqs, offs = sc.upload_reads(case["reads"], gpu)
        with sc.Plan.dual_single_end(case["template"], case["strand"], case["pools"], case["mismatches"], case["use_first"]) as plan:
            plan.count(seqs, offs)
            got = plan.read()
        assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), (case, exp, got)


@pytest.mark.parametrize("seed", range(3))
def test_wide_single_and_match_random(sc, oracle, gpu, seed):
    """Barcodes of 33..64 bases through the wide-key kernels (countSingleBarcodes, matchBarcodes)."""
    rng = random.Random(6900 + seed)
    for _ in range(15):
        case = gen.random_single_case(rng, max_vlen=64, min_vlen=33)
        try:
            exp = oracle.count_single(case["reads"], case["template"], case["strand"], case["pool"], case["mismatches"], case["use_first"])
        except Exception:
            with pytest.raises(sc.ScgError):
                run_single(sc, case, gpu)
            continue
        got = run_single(sc, case, gpu)
        assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), (case, exp, got)
    for _ in range(10):
        vlen = rng.choice([33, 48, 64])
        pool = gen.make_pool(rng, rng.choice([1, 5, 30]), vlen, "ACGT")
        seqs = [gen.mutate(rng, rng.choice(pool), 0.04, 0.01, 0.1) for _ in range(40)]
        subs, rev = rng.choice([0, 1, 2, 3]), rng.random() < 0.5
        if rev:
            seqs = [gen.rc(s) if set(s.upper()) <= set("ACGT") else s for s in seqs]
        exp = oracle.match_barcodes(seqs, pool, subs, rev)
        got = sc.match_barcodes(seqs, pool, subs, rev)
        assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]), (pool, seqs, subs, rev)


@pytest.mark.parametrize("wide", [False, True, "big"], ids=["narrow", "wide", "big"])
@pytest.mark.parametrize("seed", range(5))
def test_paired_combo_random(sc, oracle, gpu, seed, wide):
    """countPairedComboBarcodes: combinations of independently matched mates, barcode1/2-only tallies."""
    rng = random.Random(6500 + seed)
    for _ in range(6 if wide else 20):
        case = gen.random_paired_combo_case(rng, wide=wide)
        exp = oracle.count_combo_paired(case["reads1"], case["reads2"], case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                                        case["template2"], case["reverse2"], case["mismatches2"], case["pool2"], case["randomized"], case["use_first"])
        s1, o1 = sc.upload_reads(case["reads1"], gpu)
        s2, o2 = sc.upload_reads(case["reads2"], gpu)
        with sc.Plan.paired_combo(case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                                  case["template2"], case["reverse2"], case["mismatches2"], case["pool2"],
                                  case["randomized"], case["use_first"]) as plan:
            plan.count_paired(s1, s2, o1, o2)
            got = plan.read_diagnostics()
        for key in exp:
            assert np.array_equal(np.asarray(exp[key]), np.asarray(got[key])), (key, case, exp, got)


@pytest.mark.parametrize("wide", [False, True, "big"], ids=["narrow", "wide", "big"])
@pytest.mark.parametrize("seed", range(5))
def test_dual_diagnostics_random(sc, oracle, gpu, seed, wide):
    """include.invalid=TRUE: valid-pair counts, invalid combinations, barcode1/2-only tallies."""
    rng = random.Random(6000 + seed)
    for _ in range(6 if wide else 20):
        case = gen.random_dual_case(rng, hazard_free=True, max_mm=2, wide=wide)
        exp = oracle.count_dual_diag(case["reads1"], case["reads2"], case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                                     case["template2"], case["reverse2"], case["mismatches2"], case["pool2"], case["randomized"], case["use_first"])
        s1, o1 = sc.upload_reads(case["reads1"], gpu)
        s2, o2 = sc.upload_reads(case["reads2"], gpu)
        with sc.Plan.dual(case["template1"], case["reverse1"], case["mismatches1"], case["pool1"],
                          case["template2"], case["reverse2"], case["mismatches2"], case["pool2"],
                          case["randomized"], case["use_first"], diagnostics=True) as plan:
            plan.count_paired(s1, s2, o1, o2)
            got = plan.read_diagnostics()
        for key in exp:
            assert np.array_equal(np.asarray(exp[key]), np.asarray(got[key])), (key, case, exp, got)
