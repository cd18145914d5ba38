"""CPU baseline leg of bench.py (TEST/BENCH INFRASTRUCTURE, never part of the product path).

Times the reference's own CPU path on the GPU box's host cores on a bounded sample of the
benchmark workload, the way screenCounter parallelises it: one process per FASTQ file
(matrixOfSingleBarcodes + BiocParallel, R/countSingleBarcodes.R:112-113), each running the kaori
driver single-threaded.  kind = "reference" when oracle/_ref/libkaori_ref.so (real kaori) is
present, else "port" (the C restatement, which is far slower by construction and labelled so).

Worker mode (no GPU, no torch):  python -m oracle.cpu_baseline <spec.json>
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_one(spec: dict):
    from oracle.pyoracle import KaoriRef, Oracle
    kind = spec["kind"]
    e = spec["entry"]
    if kind == "reference":
        ref = KaoriRef()
        if e == "single":
            counts, total = ref.count_single(spec["fastq"][0], spec["template"], spec["strand"], spec["pools"][0], spec["mismatches"], spec["use_first"], 1)
        elif e == "combo":
            idx, freq, total = ref.count_combo(spec["fastq"][0], spec["template"], spec["strand"], spec["pools"][0], spec["pools"][1],
                                               spec["mismatches"], spec["use_first"], 1)
            counts = np.zeros(len(spec["pools"][0]) * len(spec["pools"][1]), dtype=np.int32)
            counts[idx[0].astype(np.int64) * len(spec["pools"][1]) + idx[1]] = freq
        else:
            counts, total = ref.count_dual(spec["fastq"][0], spec["template"], False, spec["mismatches"], spec["pools"][0],
                                           spec["fastq"][1], spec["template2"], False, spec["mismatches"], spec["pools"][1],
                                           False, spec["use_first"], 1)
    else:
        ora = Oracle()
        r1 = ora.parse_fastq(spec["fastq"][0])
        if e == "single":
            counts, total = ora.count_single(r1, spec["template"], spec["strand"], spec["pools"][0], spec["mismatches"], spec["use_first"])
        elif e == "combo":
            idx, freq, total = ora.count_combo(r1, spec["template"], spec["strand"], spec["pools"][0], spec["pools"][1], spec["mismatches"], spec["use_first"])
            counts = np.zeros(len(spec["pools"][0]) * len(spec["pools"][1]), dtype=np.int32)
            counts[idx[0].astype(np.int64) * len(spec["pools"][1]) + idx[1]] = freq
        else:
            r2 = ora.parse_fastq(spec["fastq"][1])
            counts, total = ora.count_dual(r1, r2, spec["template"], False, spec["mismatches"], spec["pools"][0],
                                           spec["template2"], False, spec["mismatches"], spec["pools"][1], False, spec["use_first"])
    np.save(spec["out"], np.concatenate([np.asarray(counts, dtype=np.int64), [total]]))


def run(workload, sample_mates, cores: int, workdir: str):
    """workload: screencounter_amd.synth.Workload; sample_mates: list (1 or 2) of uint8 arrays of
    n*read_len bytes.  Splits the sample into `cores` files, runs one worker process per file,
    returns (dict for the JSON line, summed counts, total)."""
    from oracle.pyoracle import KaoriRef
    from screencounter_amd.synth import reads_to_fastq
    kind = "reference" if KaoriRef.available() else "port"
    L = workload.read_len
    n = sample_mates[0].size // L
    cores = max(1, min(cores, n))
    bounds = np.linspace(0, n, cores + 1).astype(np.int64)
    specs = []
    for c in range(cores):
        lo, hi = int(bounds[c]), int(bounds[c + 1])
        fq = []
        for m, arr in enumerate(sample_mates):
            path = os.path.join(workdir, f"sample_{c}_{m}.fastq")
            reads_to_fastq(path, arr[lo * L:hi * L], L, start_index=lo)
            fq.append(path)
        spec = dict(kind=kind, entry=workload.entry, fastq=fq, template=workload.template, template2=workload.template2,
                    strand=workload.strand, pools=workload.pools, mismatches=workload.mismatches, use_first=workload.use_first,
                    out=os.path.join(workdir, f"out_{c}.npy"))
        sp = os.path.join(workdir, f"spec_{c}.json")
        with open(sp, "w") as f:
            json.dump(spec, f)
        specs.append((sp, spec["out"]))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline", sp], cwd=ROOT, env=env) for sp, _ in specs]
    rcs = [p.wait() for p in procs]
    dt = time.perf_counter() - t0
    if any(rcs):
        raise RuntimeError(f"cpu baseline worker failed: {rcs}")
    acc = None
    for _, out in specs:
        a = np.load(out)
        acc = a if acc is None else acc + a
    unit = "Mpairs/s" if workload.entry == "dual" else "Mreads/s"
    info = {"value": round(n / dt / 1e6, 4), "unit": unit, "cores": cores, "kind": kind, "seconds": round(dt, 2),
            "sample": f"first {n} {'pairs' if workload.entry == 'dual' else 'reads'} of the benchmark stream as {cores} FASTQ files, "
                      f"one single-threaded {'kaori' if kind == 'reference' else 'oracle'} process per file (incl. FASTQ parse)"}
    return info, acc[:-1].astype(np.int64), int(acc[-1])


if __name__ == "__main__":
    with open(sys.argv[1]) as f:
        _run_one(json.load(f))
