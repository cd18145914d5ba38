#!/bin/bash
# GPU box: BGZF-compressed FASTQ of config 2 -> counts through the file-level call, members inflated on the device or
# by the host threads (SCG_DEVICE_INFLATE=0).  Usage: tools/e2e_bgzf.sh ["VAR=val ..." ...]
cd $GRAFT_REPO_ROOT
N=${N:-16000000}
timeout -k 10 600 python3 - <<PY
import os
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(2, n_reads=$N)
dw = synth.DeviceWorkload(w, "cuda:0")
reads = dw.generate($N).cpu().numpy()
synth.reads_to_fastq("/dev/shm/scg_bg.fastq", reads, w.read_len)
synth.fastq_to_bgzf("/dev/shm/scg_bg.fastq", "/dev/shm/scg_bg.fastq.gz", workers=16)
print("text", os.path.getsize("/dev/shm/scg_bg.fastq") / 1e9, "GB, compressed", os.path.getsize("/dev/shm/scg_bg.fastq.gz") / 1e9, "GB", flush=True)
os.remove("/dev/shm/scg_bg.fastq")
PY
for setting in "$@"; do
env $setting SCG_TRACE=1 timeout -k 10 300 python3 - "$setting" <<PY 2>&1 | grep -v "scan slots\|upload\|amdgpu.ids"
import os, sys, time
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(2, n_reads=$N)
p = "/dev/shm/scg_bg.fastq.gz"
sc.count_single_barcodes(p, w.template, w.strand, w.pools[0][:16], 0, True, 16)
print("----", file=sys.stderr, flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    c, t = sc.count_single_barcodes(p, w.template, w.strand, w.pools[0], w.mismatches, True, 16)
    dt = time.perf_counter() - t0
    print(f"[{sys.argv[1]}] rep {rep}: {t/dt/1e6:.1f} Mreads/s ({os.path.getsize(p)/dt/1e9:.2f} GB/s compressed), mapped {int(c.sum())}", flush=True)
PY
done
rm -f /dev/shm/scg_bg.fastq.gz
