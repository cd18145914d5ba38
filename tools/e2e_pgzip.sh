#!/bin/bash
# GPU box: an ordinary single-member gzip FASTQ of config 2 (zlib level 4) -> the parallel decoder alone (host harness,
# several thread counts) and end to end through scg_count_single_barcodes with stage timings.
# Usage: tools/e2e_pgzip.sh ["VAR=val ..." ...]        N=reads (default 8000000)
cd $GRAFT_REPO_ROOT
N=${N:-8000000}
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
g++ -O2 -std=c++17 -o /tmp/pgzip_harness tests/pgzip_harness.cpp screencounter_amd/csrc/scg_pgzip.cpp -lz -ldl -lpthread || exit 1
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for T in 8 15 16 24 32; do
  echo -n "decoder alone, $T threads: "; /tmp/pgzip_harness --time /dev/shm/scg_gz.fastq.gz $T | sort -k3 -n | head -1 | awk -v n=$N '{printf "%.3f s = %.1f Mreads/s\n", $3, n/$3/1e6}'
done
for setting in "$@"; do
env $setting SCG_TRACE=1 timeout -k 10 300 python3 - "$setting" <<PY 2>&1 | grep -v "scan slots\|upload\|amdgpu.ids"
import os, sys, time
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(2, n_reads=$N)
p = "/dev/shm/scg_gz.fastq.gz"
for rep in range(3):
    t0 = time.perf_counter()
    c, t = sc.count_single_barcodes(p, w.template, w.strand, w.pools[0], w.mismatches, True, 16)
    dt = time.perf_counter() - t0
    print(f"[{sys.argv[1]}] rep {rep}: {t/dt/1e6:.2f} Mreads/s, mapped {int(c.sum())}", flush=True)
PY
done
rm -f /dev/shm/scg_gz.fastq.gz
