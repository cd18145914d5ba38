"""GPU box: one plain-FASTQ file-level call under rocprofv3 (kernel durations of the host-scan path).
usage: rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/prof_plain.py [n_reads]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import screencounter_amd as sc
from screencounter_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
w = synth.workload(2, n_reads=N)
dw = synth.DeviceWorkload(w, "cuda:0")
reads = dw.generate(N).cpu().numpy()
p = "/dev/shm/scg_pp.fastq"
synth.reads_to_fastq(p, reads, w.read_len)
pool = sc.prepare_pool(w.pools[0])
try:
    for rep in range(3):
        t0 = time.perf_counter()
        c, t = sc.count_single_barcodes(p, w.template, w.strand, pool, w.mismatches, True, 16)
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {t / dt / 1e6:.1f} Mreads/s", flush=True)
finally:
    os.remove(p)
