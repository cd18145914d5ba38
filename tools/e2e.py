#!/usr/bin/env python3
"""End-to-end (FASTQ file on /dev/shm -> counts) timing of the file-level C ABI entry point, next to
the reference on the same file.  PCIe-inclusive; this is NOT bench.py's `value`."""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pinned_batches(sc, w, reads, torch):
    """Parsed batches in pinned host memory -> counts on the host (SURVEY 8d metric 1): H2D copies on two
    streams overlapped with the counting kernel, 4 M reads per batch like the file-level stager."""
    n = reads.shape[0] // w.read_len if reads.ndim == 1 else reads.shape[0]
    flat = torch.from_numpy(reads.reshape(-1)).pin_memory()
    B = 1 << 22
    with sc.Plan.single(w.template, w.strand, w.pools[0], w.mismatches, w.use_first) as plan:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        bufs = [torch.empty(B * w.read_len, dtype=torch.uint8, device="cuda:0") for _ in streams]
        for rep in range(2):                        # first pass warms up
            plan.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k, a in enumerate(range(0, n, B)):
                b = min(a + B, n)
                s = streams[k % 2]
                with torch.cuda.stream(s):
                    dst = bufs[k % 2][: (b - a) * w.read_len]
                    dst.copy_(flat[a * w.read_len: b * w.read_len], non_blocking=True)
                    plan.count(dst, None, fixed_len=w.read_len, stream=s)
            torch.cuda.synchronize()
            counts, total = plan.read()
            dt = time.perf_counter() - t0
        print(f"pinned batches -> counts: {total} reads in {dt:.3f} s = {total / dt / 1e6:.1f} Mreads/s "
              f"({total * w.read_len / dt / 1e9:.1f} GB/s over PCIe), mapped {int(counts.sum())}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--ref-threads", type=int, default=16)
    args = ap.parse_args()
    import numpy as np
    import torch
    import screencounter_amd as sc
    from screencounter_amd import synth
    w = synth.workload(2, n_reads=args.reads)
    dw = synth.DeviceWorkload(w, "cuda:0")
    reads = dw.generate(args.reads).cpu().numpy()
    pinned_batches(sc, w, reads, torch)
    d = tempfile.mkdtemp(prefix="scg_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(d, "reads.fastq")
    synth.reads_to_fastq(path, reads, w.read_len)
    if args.gz:
        os.system(f"gzip -1 {path}")
        path += ".gz"
    size = os.path.getsize(path)
    try:
        sc.count_single_barcodes(path, w.template, w.strand, w.pools[0][:10], 0, True, 1)   # warm up (context, page cache)
        t0 = time.perf_counter()
        counts, total = sc.count_single_barcodes(path, w.template, w.strand, w.pools[0], w.mismatches, w.use_first, 16)
        dt = time.perf_counter() - t0
        print(f"libscg file-level: {total} reads, {size / 1e9:.2f} GB file in {dt:.3f} s = {total / dt / 1e6:.2f} Mreads/s ({size / dt / 1e9:.2f} GB/s of FASTQ)")
        try:
            from oracle.pyoracle import KaoriRef
            if KaoriRef.available():
                ref = KaoriRef()
                t0 = time.perf_counter()
                rc, rt = ref.count_single(path, w.template, w.strand, w.pools[0], w.mismatches, w.use_first, args.ref_threads)
                dr = time.perf_counter() - t0
                print(f"kaori ({args.ref_threads} threads, one file): {rt} reads in {dr:.3f} s = {rt / dr / 1e6:.2f} Mreads/s; counts equal: {bool(np.array_equal(rc, counts) and rt == total)}")
        except Exception as e:  # the reference library is optional here
            print("kaori not run:", e)
    finally:
        os.system(f"rm -rf {d}")


if __name__ == "__main__":
    main()
