#!/bin/bash
# GPU box: paired-end file-level call on config 4 with both mates BGZF-compressed (members inflated by the host threads).
cd $GRAFT_REPO_ROOT
N=${N:-4000000}
SCG_TRACE=1 timeout -k 10 600 python3 - <<PY 2>&1 | grep -v "amdgpu.ids"
import os, sys, time, numpy as np, torch
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(4, n_reads=$N)
dw = synth.DeviceWorkload(w, "cuda:0")
L = w.read_len
paths = []
for m in (0, 1):
    p = f"/dev/shm/scg_pb{m}.fastq"
    synth.reads_to_fastq(p, dw.generate($N, mate=m).cpu().numpy(), L)
    synth.fastq_to_bgzf(p, p + ".gz", workers=16)
    os.remove(p)
    paths.append(p + ".gz")
pools = [sc.prepare_pool(p) for p in w.pools]
sc.count_dual_barcodes(paths[0], w.template, False, 0, w.pools[0][:16], paths[1], w.template2, False, 0, w.pools[1][:16], False, w.use_first, False, 16)
for rep in range(3):
    print("----", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    c, t = sc.count_dual_barcodes(paths[0], w.template, False, w.mismatches, pools[0], paths[1], w.template2, False, w.mismatches, pools[1], False, w.use_first, False, 16)
    dt = time.perf_counter() - t0
    print(f"N=$N rep {rep}: {t/dt/1e6:.1f} Mpairs/s, mapped {int(c.sum())}", flush=True)
for p in paths:
    os.remove(p)
PY
