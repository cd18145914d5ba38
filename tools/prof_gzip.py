"""GPU box: one ordinary-gzip file-level call under rocprofv3 (kernel durations of the device gunzip path).
usage: rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/prof_gzip.py [n_reads]"""
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import screencounter_amd as sc
from screencounter_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
w = synth.workload(2, n_reads=N)
dw = synth.DeviceWorkload(w, "cuda:0")
reads = dw.generate(N).cpu().numpy()
plain = "/dev/shm/scg_pg.fastq"
synth.reads_to_fastq(plain, reads, w.read_len)
p = plain + ".gz"
c = zlib.compressobj(4, zlib.DEFLATED, 31)
with open(plain, "rb") as f, open(p, "wb") as g:
    while True:
        b = f.read(1 << 24)
        if not b:
            break
        g.write(c.compress(b))
    g.write(c.flush())
os.remove(plain)
pool = sc.prepare_pool(w.pools[0])
try:
    for rep in range(3):
        t0 = time.perf_counter()
        cnt, t = sc.count_single_barcodes(p, w.template, w.strand, pool, w.mismatches, True, 16)
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {t / dt / 1e6:.1f} Mreads/s", flush=True)
finally:
    os.remove(p)
