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


@pytest.mark.parametrize("seed", range(6))
def test_combo_random(sc, oracle, gpu, seed):
    from oracle.pyoracle import OracleError
    rng = random.Random(2000 + seed)
    done = 0
    while done < 20:
        case = gen.random_combo_case(rng)
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


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("hazard_free", [True, False])
def test_dual_random(sc, oracle, gpu, seed, hazard_free):
    # Both are compared with the cache-free oracle: hazard_free only matters when
    # comparing with the reference itself.
    rng = random.Random(3000 + seed)
    for _ in range(20):
        case = gen.random_dual_case(rng, hazard_free=hazard_free, max_mm=3)
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
