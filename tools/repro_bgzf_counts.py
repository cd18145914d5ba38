"""GPU box: the BGZF file-level call over and over on one file; every call's counts against the first call's (and the
expected number of mapped reads).  usage: python3 tools/repro_bgzf_counts.py [calls] ; settings through the environment (FORM=plain: the same reads as a plain
FASTQ file)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import screencounter_amd as sc
from screencounter_amd import synth

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = 16_000_000
w = synth.workload(2, n_reads=N)
FORM = os.environ.get("FORM", "bgzf")            # bgzf | plain
p = "/dev/shm/scg_repro.fastq.gz" if FORM == "bgzf" else "/dev/shm/scg_repro.fastq"
if not os.path.exists(p):
    dw = synth.DeviceWorkload(w, "cuda:0")
    reads = dw.generate(N).cpu().numpy()
    synth.reads_to_fastq("/dev/shm/scg_repro.fastq", reads, w.read_len)
    if FORM == "bgzf":
        synth.fastq_to_bgzf("/dev/shm/scg_repro.fastq", p, workers=16)
        os.remove("/dev/shm/scg_repro.fastq")
pool = sc.prepare_pool(w.pools[0]) if hasattr(sc, "prepare_pool") else w.pools[0]
first = None
bad = 0
for k in range(CALLS):
    c, t = sc.count_single_barcodes(p, w.template, w.strand, pool, w.mismatches, True, 16)
    c = np.asarray(c).astype(np.int64)
    if first is None:
        first = c
        print("call 0: total", t, "mapped", int(c.sum()), flush=True)
        continue
    if t != N or not np.array_equal(c, first):
        bad += 1
        d = c - first
        nz = np.nonzero(d)[0]
        print(f"call {k}: total {t} mapped {int(c.sum())} differs in {len(nz)} counters, sum of differences {int(d.sum())}, "
              f"min {int(d.min())} max {int(d.max())}, first counters {nz[:12].tolist()} diffs {d[nz[:12]].tolist()}", flush=True)
print(f"{os.environ.get('TAG', '')}: {bad} of {CALLS - 1} calls differ from the first", flush=True)
