"""Config-scale parity: the HIP path on the ACTUAL libraries and read streams of BASELINE.json's configurations
(synth.workload(c): 100 k-barcode library with its 256 Ki-slot tables and chains, 2 x 500 pools, 50 k pairs, the six
pairs-of-quarters tables of <= 2 mismatches), against an independent checker on the same reads:

  * real kaori (oracle/_ref/libkaori_ref.so, built in the container, travels to the GPU box) on >= 2^20 reads, which
    is also large enough for the index-stream + tally path to be the one that runs; or, without it,
  * the brute-force C restatement (oracle/liboracle.so) on a 20 k-read prefix.

Both the HBM-resident batch entry (scg_count_batch*) and the file-level entry points (FASTQ -> pinned staging ->
counts) are checked.  The device generator is checked byte for byte against its numpy restatement on the way.
Reference: handlers/SingleBarcodeSingleEnd.hpp:93-104, CombinatorialBarcodesSingleEnd.hpp:149-258,
DualBarcodesPairedEnd.hpp:258-347."""
import hashlib
import os

import numpy as np
import pytest

from tests.test_config1_fixture import load_fixture

pytestmark = pytest.mark.gpu


def _dense(idx, freq, n1, size):
    cells = np.zeros(size, dtype=np.int64)
    if freq.size:
        cells[idx[0].astype(np.int64) * n1 + idx[1]] = freq
    return cells


def _reference(w, host, n, tmp_path, oracle):
    """(counts int64[num_counters], total, checker name, reads checked) from the strongest checker available."""
    from oracle.pyoracle import KaoriRef
    from screencounter_amd import synth
    L = w.read_len
    if KaoriRef.available():
        ref = KaoriRef()
        paths = []
        for m, arr in enumerate(host):
            p = str(tmp_path / f"cfg{w.config_id}_{m}.fastq")
            synth.reads_to_fastq(p, arr[: n * L], L)
            paths.append(p)
        # one thread: with more, the reference's reduce() merges its search cache while other workers read it
        # ("latent race") -- it crashed once here at 16 threads
        threads = 1
        if w.entry == "single":
            c, t = ref.count_single(paths[0], w.template, w.strand, w.pools[0], w.mismatches, w.use_first, threads)
        elif w.entry == "combo":
            idx, freq, t = ref.count_combo(paths[0], w.template, w.strand, w.pools[0], w.pools[1], w.mismatches, w.use_first, threads)
            c = _dense(idx, freq, len(w.pools[1]), len(w.pools[0]) * len(w.pools[1]))
        else:
            # one thread: the reference's segmented-search cache makes its dual path order-dependent
            c, t = ref.count_dual(paths[0], w.template, False, w.mismatches, w.pools[0], paths[1], w.template2, False, w.mismatches,
                                  w.pools[1], False, w.use_first, 1)
        return np.asarray(c, dtype=np.int64), int(t), "kaori", n, paths
    m = min(n, 20_000)
    offs = np.arange(0, (m + 1) * L, L, dtype=np.uint64)
    batches = [(arr[: m * L], offs) for arr in host]
    if w.entry == "single":
        c, t = oracle.count_single(batches[0], w.template, w.strand, w.pools[0], w.mismatches, w.use_first)
    elif w.entry == "combo":
        idx, freq, t = oracle.count_combo(batches[0], w.template, w.strand, w.pools[0], w.pools[1], w.mismatches, w.use_first)
        c = _dense(idx, freq, len(w.pools[1]), len(w.pools[0]) * len(w.pools[1]))
    else:
        c, t = oracle.count_dual(batches[0], batches[1], w.template, False, w.mismatches, w.pools[0], w.template2, False, w.mismatches,
                                 w.pools[1], False, w.use_first)
    return np.asarray(c, dtype=np.int64), int(t), "oracle", m, None


@pytest.mark.parametrize("config", [2, 3, 4, 5])
def test_config_scale_parity(sc, gpu, oracle, tmp_path, config):
    from screencounter_amd import synth
    n = (1 << 20) + 4321 if config != 4 else (1 << 20) + 999      # >= 2^20: tally mode is what runs
    w = synth.workload(config, n_reads=n)
    L = w.read_len
    dw = synth.DeviceWorkload(w, gpu)
    mates = [dw.generate(n, mate=0)]
    if w.entry == "dual":
        mates.append(dw.generate(n, mate=1))
    host = [m.cpu().numpy() for m in mates]
    # the device generator against its numpy restatement (first and a late slice)
    k = 30_000
    for m in range(len(mates)):
        assert np.array_equal(host[m][: k * L], synth.generate_host(w, k, first_read=0, mate=m))
        assert np.array_equal(host[m][(n - 1000) * L:], synth.generate_host(w, 1000, first_read=n - 1000, mate=m))
    exp, exp_total, checker, checked, paths = _reference(w, host, n, tmp_path, oracle)
    with dw.plan() as plan:
        if w.entry == "dual":
            plan.count_paired(mates[0][: checked * L], mates[1][: checked * L], fixed_len1=L, fixed_len2=L, n_pairs=checked)
        else:
            plan.count(mates[0][: checked * L], fixed_len=L, n_reads=checked)
        got, total = plan.read()
    assert total == exp_total == checked
    assert np.array_equal(got.astype(np.int64), exp), f"HIP path differs from {checker} on config {config}"
    assert int(exp.sum()) > 0.8 * checked
    if paths is None:
        return
    # the same reads through the file-level entry points (parallel FASTQ stager -> pinned -> H2D -> kernels)
    if w.entry == "single":
        c, t = sc.count_single_barcodes(paths[0], w.template, w.strand, w.pools[0], w.mismatches, w.use_first, 4)
        c = c.astype(np.int64)
    elif w.entry == "combo":
        idx, freq, t = sc.count_combo_barcodes_single(paths[0], w.template, w.strand, [w.pools[0], w.pools[1]], w.mismatches, w.use_first, 4)
        c = _dense(idx, freq, len(w.pools[1]), len(w.pools[0]) * len(w.pools[1]))
    else:
        c, t = sc.count_dual_barcodes(paths[0], w.template, False, w.mismatches, w.pools[0], paths[1], w.template2, False, w.mismatches,
                                      w.pools[1], False, w.use_first, False, 4)
        c = c.astype(np.int64)
    assert t == exp_total and np.array_equal(c, exp), f"file-level entry differs from {checker} on config {config}"


def test_config1_fixture_gpu(sc, gpu):
    """Config 1 on the HIP path against the committed kaori fixture (no checker needed on the box)."""
    from screencounter_amd import synth
    fx = load_fixture()
    w = synth.workload(1)
    dw = synth.DeviceWorkload(w, gpu)
    reads = dw.generate(w.n_reads)
    assert hashlib.sha256(reads.cpu().numpy().tobytes()).hexdigest() == fx["reads_sha256"]
    with dw.plan() as plan:
        plan.count(reads, fixed_len=w.read_len, n_reads=w.n_reads)
        got, total = plan.read()
    assert total == fx["expect"]["total"]
    assert got.tolist() == fx["expect"]["counts"]


def _option_cases():
    cases = []
    for config in (2, 3, 5):
        for use_first in (True, False):
            for strand in (0, 1, 2):
                if (use_first, strand) == (True, 2):
                    continue                                  # the configuration's own option point: test_config_scale_parity
                cases.append((config, use_first, strand, False, None))
    for use_first in (True, False):
        for randomized in (False, True):
            if (use_first, randomized) == (True, False):
                continue
            cases.append((4, use_first, 0, randomized, None))
    for use_first in (True, False):
        cases.append((3, use_first, 2, False, 2))             # combinations with a budget of 2: the six pairs-of-quarters tables per pool
        cases.append((2, use_first, 2, False, 0))             # exact matching: one table, no walk
    return cases


@pytest.mark.parametrize("config,use_first,strand,randomized,mismatches", _option_cases())
def test_config_scale_options(sc, gpu, tmp_path, config, use_first, strand, randomized, mismatches):
    """The other option points of every entry -- find.best = TRUE (search_best's tie logic over the 100 k-entry index,
    SimpleSingleMatch.hpp:259-306), strand = original / reverse, randomized = TRUE (DualBarcodesPairedEnd.hpp:353-371;
    config 4's pools are >= 3 apart, so the reference's cache-order hazard, SURVEY.md A.7, cannot occur) -- on 2^18 reads of
    the configuration's own stream and library, against real kaori."""
    from oracle.pyoracle import KaoriRef
    from screencounter_amd import synth
    from screencounter_amd.engine import Plan
    if not KaoriRef.available():
        pytest.skip("oracle/_ref/libkaori_ref.so is not here")
    n = (1 << 18) + 77
    w = synth.workload(config, n_reads=n)
    L = w.read_len
    dw = synth.DeviceWorkload(w, gpu)
    mates = [dw.generate(n, mate=0)]
    if w.entry == "dual":
        mates.append(dw.generate(n, mate=1))
    if mismatches is not None:
        w.mismatches = mismatches                            # (the stream is the configuration's; only the budget differs)
    paths = []
    for m, t in enumerate(mates):
        p = str(tmp_path / f"opt{config}_{m}.fastq")
        synth.reads_to_fastq(p, t.cpu().numpy(), L)
        paths.append(p)
    ref = KaoriRef()
    dev = gpu.index if gpu.index is not None else 0
    if w.entry == "single":
        exp, exp_total = ref.count_single(paths[0], w.template, strand, w.pools[0], w.mismatches, use_first, 1)
        plan = Plan.single(w.template, strand, w.pools[0], w.mismatches, use_first, device=dev)
    elif w.entry == "combo":
        idx, freq, exp_total = ref.count_combo(paths[0], w.template, strand, w.pools[0], w.pools[1], w.mismatches, use_first, 1)
        exp = _dense(idx, freq, len(w.pools[1]), len(w.pools[0]) * len(w.pools[1]))
        plan = Plan.combo(w.template, strand, w.pools[0], w.pools[1], w.mismatches, use_first, device=dev)
    else:
        exp, exp_total = ref.count_dual(paths[0], w.template, False, w.mismatches, w.pools[0], paths[1], w.template2, False, w.mismatches,
                                        w.pools[1], randomized, use_first, 1)
        plan = Plan.dual(w.template, False, w.mismatches, w.pools[0], w.template2, False, w.mismatches, w.pools[1],
                         randomized=randomized, use_first=use_first, device=dev)
    with plan:
        if w.entry == "dual":
            plan.count_paired(mates[0], mates[1], fixed_len1=L, fixed_len2=L, n_pairs=n)
        else:
            plan.count(mates[0], fixed_len=L, n_reads=n)
        got, total = plan.read()
    exp = np.asarray(exp, dtype=np.int64)
    assert total == exp_total == n
    assert np.array_equal(got.astype(np.int64), exp), f"HIP path differs from kaori on config {config}, use_first={use_first}, strand={strand}, randomized={randomized}"
    assert int(exp.sum()) > (0.3 if (strand != 2 and w.entry != "dual") or mismatches == 0 else 0.7) * n
