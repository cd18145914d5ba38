#!/usr/bin/env python3
"""Kernel time of the dual-barcode plans on config 4's stream: plain, include.invalid=TRUE, paired combinations."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import screencounter_amd as sc
from screencounter_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
w = synth.workload(4, n_reads=n)
dw = synth.DeviceWorkload(w, "cuda:0")
m1, m2 = dw.generate(n, mate=0), dw.generate(n, mate=1)
L = w.read_len
plans = {
    "dual": lambda: sc.Plan.dual(w.template, False, w.mismatches, w.pools[0], w.template2, False, w.mismatches, w.pools[1], False, w.use_first),
    "dual+diagnostics": lambda: sc.Plan.dual(w.template, False, w.mismatches, w.pools[0], w.template2, False, w.mismatches, w.pools[1], False, w.use_first, diagnostics=True),
    "paired combinations": lambda: sc.Plan.paired_combo(w.template, False, w.mismatches, sorted(set(w.pools[0])), w.template2, False, w.mismatches, sorted(set(w.pools[1])), False, w.use_first),
}
for name, make in plans.items():
    with make() as plan:
        for rep in range(3):
            plan.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.count_paired(m1, m2, fixed_len1=L, fixed_len2=L, n_pairs=n)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{name:22s} {dt * 1e3:8.2f} ms for {n} pairs = {n / dt / 1e6:8.1f} Mpairs/s", flush=True)
