"""BASELINE.json configs[0]: countSingleBarcodes, 1 M x 75 bp, 1 k-barcode library, exact,
forward strand -- the reference's own CPU-runnable case.  tests/golden/config1_kaori.json holds real kaori's output on
that stream (oracle/gen_golden.py::config1_fixture); the C restatement must reproduce it here, the HIP path on the GPU
box (tests/test_gpu_config_parity.py::test_config1_fixture_gpu)."""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_fixture():
    with open(os.path.join(ROOT, "tests", "golden", "config1_kaori.json")) as f:
        return json.load(f)


def test_config1_oracle_matches_kaori_fixture(oracle):
    from screencounter_amd import synth
    fx = load_fixture()
    w = synth.workload(1)
    assert (w.n_reads, w.read_len, w.template, w.strand, w.mismatches, w.use_first) == \
        (fx["n_reads"], fx["read_len"], fx["template"], fx["strand"], fx["mismatches"], fx["use_first"])
    assert hashlib.sha256("\n".join(w.pools[0]).encode()).hexdigest() == fx["pool_sha256"]
    reads = synth.generate_host(w, w.n_reads)
    assert hashlib.sha256(reads.tobytes()).hexdigest() == fx["reads_sha256"]      # the generator is part of the fixture
    offs = np.arange(0, (w.n_reads + 1) * w.read_len, w.read_len, dtype=np.uint64)
    counts, total = oracle.count_single((reads, offs), w.template, w.strand, w.pools[0], w.mismatches, w.use_first)
    assert total == fx["expect"]["total"] == w.n_reads
    assert counts.tolist() == fx["expect"]["counts"]
    assert 0.6 * w.n_reads < int(counts.sum()) < 0.8 * w.n_reads
