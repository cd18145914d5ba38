#!/usr/bin/env python3
"""bench.py -- headline benchmark of the barcode-counting hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2] [--reads R]

A *step* is one pass of the hot path (constant-flank scan -> variable-region extraction ->
library match -> per-barcode atomic count [-> RCCL all-reduce of the count vectors when N > 1])
over one batch of synthetic reads that is already resident in HBM.  At N = 1 the batch is
BASELINE.json configs[1]: countSingleBarcodes, 100 M x 150 bp reads against a 100 k-barcode
library, <=1 mismatch, both strands.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank holds its own shard of that size (weak
scaling, no data-path collective other than the final count reduce).

Rank 0 prints ONE JSON line: metric/value as BASELINE.json names them, plus
  roofline     -- algorithmic bytes (sum of read lengths) / average counting-kernel duration
                  measured with HIP events on the launch stream, against the 8 TB/s HBM peak
  cpu_baseline -- the reference CPU path (real kaori via oracle/_ref when present) timed on this
                  box's host cores on a bounded prefix of the same stream, N = 1 only, with the
                  GPU counts on that prefix checked bit-exact against it.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config: 2 single (default), 3 combo, 4 dual, 5 single <=2mm")
    ap.add_argument("--reads", type=int, default=None, help="reads (pairs) per GPU; default = the config's full size")
    ap.add_argument("--library", type=int, default=None, help="override the library size (tests only)")
    ap.add_argument("--cpu-sample", type=int, default=None, help="reads in the CPU-baseline sample (0 disables)")
    ap.add_argument("--cpu-cores", type=int, default=16)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # one rank per GPU; SCG_BENCH_SHARE_GPU=1 lets several ranks share a card (rehearsal on a 1-GPU box, gloo)
    share = os.environ.get("SCG_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)   # nccl == RCCL on ROCm

    import screencounter_amd as sc
    from screencounter_amd import synth

    w = synth.workload(args.config, n_reads=args.reads, n_library=args.library)
    dw = synth.DeviceWorkload(w, device)
    n = w.n_reads
    L = w.read_len
    first = rank * n    # each rank regenerates its own shard of the global stream
    mates = [dw.generate(n, first_read=first, mate=0)]
    if w.entry == "dual":
        mates.append(dw.generate(n, first_read=first, mate=1))
    torch.cuda.synchronize()

    plan = dw.plan()
    counters = torch.zeros(plan.num_counters, dtype=torch.int32, device=device)
    plan.bind_counters(counters)

    def step():
        plan.reset()
        if w.entry == "dual":
            plan.count_paired(mates[0], mates[1], fixed_len1=L, fixed_len2=L, n_pairs=n)
        else:
            plan.count(mates[0], fixed_len=L, n_reads=n)
        if world > 1:
            dist.all_reduce(counters, op=dist.ReduceOp.SUM)   # the path's one exchange step

    for _ in range(args.warmup):
        step()
    plan.set_profiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, launches = plan.kernel_stats()
    plan.set_profiling(False)

    final_counts = counters.cpu().numpy().astype(np.int64)
    mapped = int(final_counts.sum())

    if rank == 0:
        unit = "Mpairs/s" if w.entry == "dual" else "Mreads/s"
        total_units = n * world * args.steps
        value = total_units / elapsed / 1e6
        avg_kernel_s = kernel_ms / 1e3 / max(launches, 1)
        algo_bytes = n * w.bytes_per_unit   # per launch: sum of read lengths
        achieved = algo_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        # HBM-side bytes per launch from the PMC passes recorded under profiles/ (FETCH_SIZE + WRITE_SIZE,
        # separate rocprofv3 --pmc runs of this same command, gfx950 correction applied: DESIGN.md section 5)
        traffic, traffic_source = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                rec = json.load(f).get(str(args.config))
            if rec and rec.get("bytes_per_read"):
                traffic = round(rec["bytes_per_read"] * n)
                traffic_source = rec["source"]
        except (OSError, ValueError):
            pass
        out = {
            "metric": "Mreads/s (whole node) + achieved HBM GB/s, 100M x 150bp vs 100k barcodes <=1mm",
            "value": round(value, 3),
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": w.describe(), "reads_per_gpu": n, "read_len": L, "library": [len(p) for p in w.pools],
                       "max_mismatches": w.mismatches, "strand": ["forward", "reverse", "both"][w.strand] if w.entry != "dual" else "original/original",
                       "parallelism": f"read-sharded x{world}" + (" + RCCL all-reduce of counts" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": f"{w.entry}_staged_kernel", "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": launches,
                         "algorithmic_bytes_per_launch": algo_bytes, "traffic_source": traffic_source,
                         # the whole step (counting kernel + tally / fold kernel [+ all-reduce]) against the same peak
                         "step_achieved": round(algo_bytes * world / (elapsed / args.steps) / 1e9 / world, 2),
                         "step_frac": round(algo_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5)},
            "mapped_fraction": round(mapped / (n * world), 5),
        }

        sample = args.cpu_sample
        if sample is None:
            sample = min(n, 50_000_000 if w.entry != "dual" else 20_000_000)
        if world == 1 and sample > 0:
            sys.path.insert(0, ROOT)
            from oracle import cpu_baseline
            sample = min(sample, n)
            host = [m[:sample * L].cpu().numpy() for m in mates]
            # GPU counts on exactly the sample, for the same-run parity check
            plan.reset()
            if w.entry == "dual":
                plan.count_paired(mates[0], mates[1], fixed_len1=L, fixed_len2=L, n_pairs=sample)
            else:
                plan.count(mates[0], fixed_len=L, n_reads=sample)
            gpu_counts, gpu_total = plan.read()
            workdir = tempfile.mkdtemp(prefix="scg_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
            try:
                info, cpu_counts, cpu_total = cpu_baseline.run(w, host, args.cpu_cores, workdir)
            finally:
                shutil.rmtree(workdir, ignore_errors=True)
            info["parity"] = bool(gpu_total == cpu_total and np.array_equal(gpu_counts.astype(np.int64), cpu_counts))
            out["cpu_baseline"] = info
            if not info["parity"]:
                out["cpu_baseline"]["parity_note"] = "GPU counts differ from the CPU reference on the sample"
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))

    plan.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
