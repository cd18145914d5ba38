#!/bin/bash
# GPU box: stage timings of one file-level call on a plain FASTQ of config 2 (SCG_TRACE=1) at two sizes.
cd $GRAFT_REPO_ROOT
for N in 8000000 32000000; do
SCG_TRACE=1 timeout -k 10 600 python3 - <<PY
import os, time, numpy as np, torch
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(2, n_reads=$N)
dw = synth.DeviceWorkload(w, "cuda:0")
reads = dw.generate($N).cpu().numpy()
p = "/dev/shm/scg_trace.fastq"
synth.reads_to_fastq(p, reads, w.read_len)
sc.count_single_barcodes(p, w.template, w.strand, w.pools[0][:16], 0, True, 16)
for rep in range(2):
    t0 = time.perf_counter()
    c, t = sc.count_single_barcodes(p, w.template, w.strand, w.pools[0], w.mismatches, True, 16)
    dt = time.perf_counter() - t0
    print(f"N=$N rep {rep}: {t/dt/1e6:.1f} Mreads/s ({os.path.getsize(p)/dt/1e9:.1f} GB/s of text), mapped {int(c.sum())}", flush=True)
os.remove(p)
PY
done
