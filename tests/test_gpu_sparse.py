"""Combination spaces beyond the dense histogram: the device writes one 64-bit key per read (or pair), sorts and run-length
encodes each batch (csrc/scg_sparse.hip) and the host merges the runs -- the reference's own algorithm for every size
(inst/include/kaori/utils.hpp:173-198 sort_combinations, src/utils.h:14-45 count_combinations).

Two kinds of test: the combination cases of the other GPU modules once more with the dense limit at 0 cells
($SCG_DENSE_CELLS), so that every handler with a combination grid runs through the streams; and grids that really are
beyond the limit (2 x 40 000 pools = 1.6e9 cells), against the oracle."""
import random

import numpy as np
import pytest

from tests import gen
from tests import test_gpu_parity as parity
from tests import test_gpu_golden as golden
from tests import test_gpu_ingest as ingest
from tests.test_gpu_parity import engine                           # noqa: F401  (autouse: staged and general kernels)
from tests.test_gpu_ingest import plain_scan, bgzf_inflate         # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def every_grid_is_sparse(request, monkeypatch):
    if "real_limit" not in request.keywords:
        monkeypatch.setenv("SCG_DENSE_CELLS", "0")


# the same cases, the same expectations: only the path to the counts differs
test_combo_random = parity.test_combo_random
test_paired_combo_random = parity.test_paired_combo_random
test_dual_diagnostics_random = parity.test_dual_diagnostics_random
test_dual_single_end_diagnostics_random = parity.test_dual_single_end_diagnostics_random
test_long_and_mixed_reads = parity.test_long_and_mixed_reads
test_gpu_matches_golden = golden.test_gpu_matches_golden
test_r_level_wrappers = golden.test_r_level_wrappers
test_matrix_of_files_scheduled_over_devices = golden.test_matrix_of_files_scheduled_over_devices
test_combo_and_dual_single_end_through_the_scan = ingest.test_combo_and_dual_single_end_through_the_scan
test_multi_file_entries_equal_per_file_calls = ingest.test_multi_file_entries_equal_per_file_calls
test_paired_files_through_the_scan = ingest.test_paired_files_through_the_scan
test_paired_plain_files_over_several_devices = ingest.test_paired_plain_files_over_several_devices


def big_pools(rng, n, length):
    seen = set()
    while len(seen) < n:
        seen.add(gen.rand_seq(rng, length))
    out = sorted(seen)
    rng.shuffle(out)
    return out


@pytest.mark.real_limit
def test_grid_of_1_6e9_cells(sc, gpu, tmp_path, monkeypatch):
    """countComboBarcodes with 2 x 40 000 barcodes: 1.6e9 cells, 6.4 GB as a dense histogram and beyond the 2^30 cells the
    dense mode is limited to -- against the outputs of real kaori (tests/golden/kaori_large_grid.json).  Reads, a plan reused
    over several batches, the file entry and two devices of one call."""
    from tests import golden_util as G
    case, expect = G.large_grid()
    template, pool0, pool1, reads = case["template"], case["pool0"], case["pool1"], case["reads"]
    exp = (np.asarray(expect["indices"], dtype=np.int32).reshape(2, -1), np.asarray(expect["freq"], dtype=np.int32), expect["total"])
    assert len(exp[1]) > 3000
    thirds = [reads[0:5000], reads[5000:14001], reads[14001:]]
    with sc.Plan.combo(template, 2, pool0, pool1, 1, True) as plan:
        for part in thirds:
            seqs, offs = sc.upload_reads(part, gpu)
            plan.count(seqs, offs)
        got = plan.read_combo()
        assert got[2] == exp[2] and np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
        plan.reset()
        seqs, offs = sc.upload_reads(reads, gpu)
        plan.count(seqs, offs)
        again = plan.read_combo()
    assert again[2] == exp[2] and np.array_equal(again[0], exp[0]) and np.array_equal(again[1], exp[1])
    path = str(tmp_path / "big.fastq")
    open(path, "wb").write(gen.fastq_text(reads))
    for devices in (None, "0,0"):                             # (one card twice: two plans of one call, their runs merged)
        if devices:
            monkeypatch.setenv("SCG_DEVICES", devices)
        idx, freq, total = sc.count_combo_barcodes_single(path, template, 2, [pool0, pool1], 1, True, 1)
        assert total == exp[2] and np.array_equal(idx, exp[0]) and np.array_equal(freq, exp[1])


@pytest.mark.real_limit
def test_invalid_pairs_of_a_large_dual_design(sc, oracle, gpu):
    """include.invalid=TRUE with 20 000 x 20 000 possible invalid combinations (4e8 cells > the dense limit of 2^26)."""
    rng = random.Random(78)
    pool1, pool2 = big_pools(rng, 20000, 14), big_pools(rng, 20000, 14)
    t1, t2 = "CAGT" + "-" * 14 + "GGA", "TTAC" + "-" * 14 + "CCT"
    reads1, reads2 = [], []
    for _ in range(12000):
        a = rng.randrange(20000)
        b = a if rng.random() < 0.6 else rng.randrange(20000)
        r1 = gen.rand_seq(rng, rng.randrange(0, 9)) + gen.fill_template(t1, [pool1[a]]) + gen.rand_seq(rng, rng.randrange(0, 9))
        r2 = gen.rand_seq(rng, rng.randrange(0, 9)) + gen.fill_template(t2, [pool2[b]]) + gen.rand_seq(rng, rng.randrange(0, 9))
        if rng.random() < 0.05:
            r2 = gen.rand_seq(rng, len(r2))
        reads1.append(gen.mutate(rng, r1, 0.01, 0.002, 0.0))
        reads2.append(gen.mutate(rng, r2, 0.01, 0.002, 0.0))
    exp = oracle.count_dual_diag(reads1, reads2, t1, False, 1, pool1, t2, False, 1, pool2, False, True)
    s1, o1 = sc.upload_reads(reads1, gpu)
    s2, o2 = sc.upload_reads(reads2, gpu)
    with sc.Plan.dual(t1, False, 1, pool1, t2, False, 1, pool2, False, True, diagnostics=True) as plan:
        plan.count_paired(s1, s2, o1, o2)
        got = plan.read_diagnostics()
    for key in exp:
        assert np.array_equal(np.asarray(exp[key]), np.asarray(got[key])), key
    assert len(got["freq"]) > 1000
