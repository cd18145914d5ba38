#!/bin/bash
# GPU box: an ordinary (single-member) gzip FASTQ of config 2 -> counts, inflated whole by libdeflate or streamed
# through zlib (SCG_LIBDEFLATE=0).  Usage: tools/e2e_gzip.sh ["VAR=val ..." ...]
cd $GRAFT_REPO_ROOT
N=${N:-4000000}
timeout -k 10 600 python3 - <<PY
import os, zlib
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(2, n_reads=$N)
dw = synth.DeviceWorkload(w, "cuda:0")
reads = dw.generate($N).cpu().numpy()
synth.reads_to_fastq("/dev/shm/scg_gz.fastq", reads, w.read_len)
c = zlib.compressobj(4, zlib.DEFLATED, 31)
with open("/dev/shm/scg_gz.fastq", "rb") as f, open("/dev/shm/scg_gz.fastq.gz", "wb") as g:
    while True:
        b = f.read(1 << 24)
        if not b:
            break
        g.write(c.compress(b))
    g.write(c.flush())
print("text", os.path.getsize("/dev/shm/scg_gz.fastq") / 1e9, "GB, compressed", os.path.getsize("/dev/shm/scg_gz.fastq.gz") / 1e9, "GB", flush=True)
os.remove("/dev/shm/scg_gz.fastq")
PY
for setting in "$@"; do
env $setting SCG_TRACE=1 timeout -k 10 300 python3 - "$setting" <<PY 2>&1 | grep -v "scan slots\|upload\|amdgpu.ids"
import os, sys, time
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(2, n_reads=$N)
p = "/dev/shm/scg_gz.fastq.gz"
for rep in range(2):
    t0 = time.perf_counter()
    c, t = sc.count_single_barcodes(p, w.template, w.strand, w.pools[0], w.mismatches, True, 16)
    dt = time.perf_counter() - t0
    print(f"[{sys.argv[1]}] rep {rep}: {t/dt/1e6:.2f} Mreads/s, mapped {int(c.sum())}", flush=True)
PY
done
rm -f /dev/shm/scg_gz.fastq.gz
