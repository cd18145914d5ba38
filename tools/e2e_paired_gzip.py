"""GPU box: config 4's mates as two ordinary gzip files (zlib level 4) -> counts through scg_count_dual_barcodes, the mates
decoded by the device (default) or by the host threads (SCG_DEVICE_GUNZIP=0).  usage: python3 tools/e2e_paired_gzip.py [n_pairs]"""
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import screencounter_amd as sc
    from screencounter_amd import synth
    w = synth.workload(4, n_reads=N)
    pools = [sc.prepare_pool(p) for p in w.pools]
    paths = ["/dev/shm/scg_pz_%d.fastq.gz" % m for m in range(2)]
    for rep in range(3):
        t0 = time.perf_counter()
        c, t = sc.count_dual_barcodes(paths[0], w.template, False, w.mismatches, pools[0], paths[1], w.template2, False, w.mismatches, pools[1],
                                      False, w.use_first, False, 16)
        dt = time.perf_counter() - t0
        print(f"[{os.environ.get('SCG_DEVICE_GUNZIP', 'default')}] rep {rep}: {t / dt / 1e6:.1f} Mpairs/s, mapped {int(c.sum())}", flush=True)
    sys.exit(0)

import screencounter_amd as sc
from screencounter_amd import synth
w = synth.workload(4, n_reads=N)
dw = synth.DeviceWorkload(w, "cuda:0")
for m in range(2):
    reads = dw.generate(N, mate=m).cpu().numpy()
    plain = "/dev/shm/scg_pz_%d.fastq" % m
    synth.reads_to_fastq(plain, reads, w.read_len)
    comp = zlib.compressobj(4, zlib.DEFLATED, 31)
    with open(plain, "rb") as f, open(plain + ".gz", "wb") as g:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            g.write(comp.compress(b))
        g.write(comp.flush())
    os.remove(plain)
try:
    for setting in (os.environ.get("SETTINGS", "1,0").split(",")):
        env = dict(os.environ, SCG_DEVICE_GUNZIP=setting)
        subprocess.run([sys.executable, __file__, str(N), "child"], env=env, check=False)
finally:
    for m in range(2):
        os.remove("/dev/shm/scg_pz_%d.fastq.gz" % m)
