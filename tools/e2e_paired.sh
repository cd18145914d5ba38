#!/bin/bash
# GPU box: stage timings (SCG_TRACE=1) of the paired-end file-level call on config 4, plain FASTQ on tmpfs.
cd $GRAFT_REPO_ROOT
N=${N:-8000000}
SCG_TRACE=1 timeout -k 10 600 python3 - <<PY 2>&1 | grep -v "amdgpu.ids"
import os, sys, time, numpy as np, torch
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(4, n_reads=$N)
dw = synth.DeviceWorkload(w, "cuda:0")
m1, m2 = dw.generate($N, mate=0), dw.generate($N, mate=1)
L = w.read_len
p1, p2 = "/dev/shm/scg_p1.fastq", "/dev/shm/scg_p2.fastq"
synth.reads_to_fastq(p1, m1.cpu().numpy(), L)
synth.reads_to_fastq(p2, m2.cpu().numpy(), L)
pools = [sc.prepare_pool(p) for p in w.pools]
sc.count_dual_barcodes(p1, w.template, False, 0, w.pools[0][:16], p2, w.template2, False, 0, w.pools[1][:16], False, w.use_first, False, 16)
for rep in range(3):
    print("----", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    c, t = sc.count_dual_barcodes(p1, w.template, False, w.mismatches, pools[0], p2, w.template2, False, w.mismatches, pools[1], False, w.use_first, False, 16)
    dt = time.perf_counter() - t0
    print(f"N=$N rep {rep}: {t/dt/1e6:.1f} Mpairs/s, mapped {int(c.sum())}", flush=True)
os.remove(p1); os.remove(p2)
PY
