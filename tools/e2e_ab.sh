#!/bin/bash
# GPU box: one plain FASTQ of config 2 (32 M reads), the file-level call repeated under different ingestion settings
# (each in its own process: the switches are read once).  Usage: tools/e2e_ab.sh "VAR=val VAR=val" "VAR=val" ...
cd $GRAFT_REPO_ROOT
N=${N:-32000000}
timeout -k 10 300 python3 - <<PY
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(2, n_reads=$N)
dw = synth.DeviceWorkload(w, "cuda:0")
reads = dw.generate($N).cpu().numpy()
synth.reads_to_fastq("/dev/shm/scg_ab.fastq", reads, w.read_len)
PY
for round in 1 2; do
for setting in "$@"; do
env $setting SCG_TRACE=1 timeout -k 10 300 python3 - "$setting" <<PY 2>&1 | grep -v "scan slots\|upload\|amdgpu.ids"
import os, sys, time
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(2, n_reads=$N)
p = "/dev/shm/scg_ab.fastq"
sc.count_single_barcodes(p, w.template, w.strand, w.pools[0][:16], 0, True, 16)
print("----", file=sys.stderr, flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    c, t = sc.count_single_barcodes(p, w.template, w.strand, w.pools[0], w.mismatches, True, int(os.environ.get("NT", "16")))
    dt = time.perf_counter() - t0
    print(f"[{sys.argv[1]}] rep {rep}: {t/dt/1e6:.1f} Mreads/s ({os.path.getsize(p)/dt/1e9:.1f} GB/s of text), mapped {int(c.sum())}", flush=True)
PY
done
done
rm -f /dev/shm/scg_ab.fastq
