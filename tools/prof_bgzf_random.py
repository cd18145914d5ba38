"""GPU box: one call on a BGZF file of random-sequence reads under rocprofv3 (PMC of the device inflate kernel).
usage: rocprofv3 --kernel-trace --pmc ... -- python3 tools/prof_bgzf_random.py [n_reads]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import screencounter_amd as sc
from screencounter_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(1)
L = 150
p = "/dev/shm/scg_prnd.fastq"
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(N, L), dtype=np.uint8)]
qual = np.frombuffer(b"FFFFFFFF:F,F#", dtype=np.uint8)[rng.integers(0, 13, size=(N, L), dtype=np.uint8)]
rec = np.empty((N, 2 * L + 23), dtype=np.uint8)
names = np.char.encode(np.char.add("@r", np.char.zfill(np.arange(N).astype(str), 16)))
rec[:, :18] = np.frombuffer(b"".join(names.tolist()), dtype=np.uint8).reshape(N, 18)
rec[:, 18] = 10
rec[:, 19:19 + L] = seq
rec[:, 19 + L] = 10; rec[:, 20 + L] = ord("+"); rec[:, 21 + L] = 10
rec[:, 22 + L:22 + 2 * L] = qual
rec[:, 22 + 2 * L] = 10
open(p, "wb").write(rec.tobytes())
synth.fastq_to_bgzf(p, p + ".gz", workers=16)
print("text", os.path.getsize(p), "compressed", os.path.getsize(p + ".gz"), flush=True)
os.remove(p)
w = synth.workload(2, n_reads=1000)
try:
    c, t = sc.count_single_barcodes(p + ".gz", w.template, w.strand, w.pools[0], w.mismatches, True, 16)
    print("reads", t, flush=True)
finally:
    os.remove(p + ".gz")
