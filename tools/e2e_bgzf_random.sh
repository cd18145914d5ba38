#!/bin/bash
# GPU box: BGZF FASTQ whose sequences are random bases with varied qualities (closer to real reads than the benchmark
# stream's constant flanks) -> counts, under the given settings.  Usage: tools/e2e_bgzf_random.sh "VAR=val" ...
cd $GRAFT_REPO_ROOT
N=${N:-8000000}
timeout -k 10 600 python3 - <<PY
import os, numpy as np
from screencounter_amd import synth
rng = np.random.default_rng(1)
L = 150
p = "/dev/shm/scg_rnd.fastq"
with open(p, "wb") as f:
    B = 1_000_000
    for a in range(0, $N, B):
        n = min(B, $N - a)
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
        qual = np.frombuffer(b"FFFFFFFF:F,F#", dtype=np.uint8)[rng.integers(0, 13, size=(n, L), dtype=np.uint8)]
        rec = np.empty((n, 2 * L + 24), dtype=np.uint8)
        names = np.char.encode(np.char.add("@r", np.char.zfill(np.arange(a, a + n).astype(str), 16)))
        rec[:, :18] = np.frombuffer(b"".join(names.tolist()), dtype=np.uint8).reshape(n, 18)
        rec[:, 18] = 10
        rec[:, 19:19 + L] = seq
        rec[:, 19 + L] = 10; rec[:, 20 + L] = ord("+"); rec[:, 21 + L] = 10
        rec[:, 22 + L:22 + 2 * L] = qual
        rec[:, 22 + 2 * L] = 10
        f.write(rec[:, :23 + 2 * L].tobytes())
synth.fastq_to_bgzf(p, p + ".gz", workers=16)
print("text", os.path.getsize(p) / 1e9, "GB, compressed", os.path.getsize(p + ".gz") / 1e9, "GB", flush=True)
os.remove(p)
PY
for setting in "$@"; do
env $setting SCG_TRACE=1 timeout -k 10 300 python3 - "$setting" <<PY 2>&1 | grep "rep\|device inflate"
import os, sys, time
import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(2, n_reads=1000)
p = "/dev/shm/scg_rnd.fastq.gz"
sc.count_single_barcodes(p, w.template, w.strand, w.pools[0][:16], 0, True, 16)
for rep in range(3):
    t0 = time.perf_counter()
    c, t = sc.count_single_barcodes(p, w.template, w.strand, w.pools[0], w.mismatches, True, 16)
    dt = time.perf_counter() - t0
    print(f"[{sys.argv[1][-24:]}] rep {rep}: {t/dt/1e6:.1f} Mreads/s, total {t}", flush=True)
PY
done
rm -f /dev/shm/scg_rnd.fastq.gz
