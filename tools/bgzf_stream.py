"""GPU box helper: writes a BGZF FASTQ of one of the two measurement streams to a path, or counts it through the file-level
call.  usage: python3 tools/bgzf_stream.py make <bench|random> <n_reads> <path.gz>
              python3 tools/bgzf_stream.py count <path.gz> [reps]          (prints Mreads/s per call)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import screencounter_amd as sc
from screencounter_amd import synth


def make(kind, n, path):
    plain = path + ".plain"
    if kind == "bench":
        w = synth.workload(2, n_reads=n)
        dw = synth.DeviceWorkload(w, "cuda:0")
        synth.reads_to_fastq(plain, dw.generate(n).cpu().numpy(), w.read_len)
    else:
        rng = np.random.default_rng(1)
        L = 150
        with open(plain, "wb") as f:
            B = 1_000_000
            for a in range(0, n, B):
                m = min(B, n - a)
                seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m, L), dtype=np.uint8)]
                qual = np.frombuffer(b"FFFFFFFF:F,F#", dtype=np.uint8)[rng.integers(0, 13, size=(m, L), dtype=np.uint8)]
                rec = np.empty((m, 2 * L + 23), dtype=np.uint8)
                names = np.char.encode(np.char.add("@r", np.char.zfill(np.arange(a, a + m).astype(str), 16)))
                rec[:, :18] = np.frombuffer(b"".join(names.tolist()), dtype=np.uint8).reshape(m, 18)
                rec[:, 18] = 10
                rec[:, 19:19 + L] = seq
                rec[:, 19 + L] = 10; rec[:, 20 + L] = ord("+"); rec[:, 21 + L] = 10
                rec[:, 22 + L:22 + 2 * L] = qual
                rec[:, 22 + 2 * L] = 10
                f.write(rec.tobytes())
    synth.fastq_to_bgzf(plain, path, workers=16)
    print(f"{kind}: text {os.path.getsize(plain)} compressed {os.path.getsize(path)}", flush=True)
    os.remove(plain)


def count(path, reps):
    w = synth.workload(2, n_reads=1000)
    sc.count_single_barcodes(path, w.template, w.strand, w.pools[0][:16], 0, True, 16)       # context, page cache
    for rep in range(reps):
        t0 = time.perf_counter()
        c, t = sc.count_single_barcodes(path, w.template, w.strand, w.pools[0], w.mismatches, True, 16)
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {t} reads, {t / dt / 1e6:.1f} Mreads/s", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "make":
        make(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    else:
        count(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
