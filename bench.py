#!/usr/bin/env python3
"""bench.py -- headline benchmark of the barcode-counting hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2] [--reads R]

A *step* is one pass of the hot path (constant-flank scan -> variable-region extraction ->
library match -> per-barcode count [-> RCCL all-reduce of the count vectors when N > 1]) over one
batch of synthetic reads that is already resident in HBM.  At N = 1 the batch is BASELINE.json
configs[1]: countSingleBarcodes, 100 M x 150 bp reads against a 100 k-barcode library, <= 1
mismatch, both strands.  With N > 1 every rank holds its own shard of that size (weak
scaling, no data-path collective other than the final count reduce).

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own
ranks: a child `python -m torch.distributed.run --nproc-per-node N bench.py ...` is spawned BEFORE this
process imports torch or touches HIP, its output is relayed and its exit code returned (a process that
has initialised the GPU is never re-exec'ed).  Under torch.distributed.run (the driver's form) each rank
reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment as usual.

Rank 0 prints ONE JSON line: metric/value as BASELINE.json names them, plus
  roofline     -- algorithmic bytes (sum of read lengths) / average counting-kernel duration
                  measured with HIP events on the launch stream, against the 8 TB/s HBM peak
  cpu_baseline -- the reference CPU path (real kaori via oracle/_ref when present) timed on this
                  box's host cores on a bounded prefix of the same stream, N = 1 only, with the
                  GPU counts on that prefix checked bit-exact against it
  e2e          -- (N = 1) PCIe-inclusive rates on bounded samples of the same stream, never `value`:
                  parsed batches in pinned host memory -> counts on the host (SURVEY.md 8d metric 1),
                  and FASTQ file (tmpfs) -> counts through the file-level C ABI entry point.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
KERNEL_OF = {"single": "single_staged_kernel", "combo": "combo_staged_kernel", "dual": "dual_passes_kernel"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500, help="timed steps (default 500: ~2 s of back-to-back launches)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=float, default=1.5,
                    help="seconds of untimed back-to-back steps after the warm-up, so that the timed steps run at sustained clocks")
    ap.add_argument("--config", type=int, default=2, help="BASELINE.json config: 2 single (default), 3 combo, 4 dual, 5 single <=2mm")
    ap.add_argument("--reads", type=int, default=None, help="reads (pairs) per GPU; default = the config's full size")
    ap.add_argument("--library", type=int, default=None, help="override the library size (tests only)")
    ap.add_argument("--cpu-sample", type=int, default=None, help="reads in the CPU-baseline sample (0 disables)")
    ap.add_argument("--cpu-cores", type=int, default=None, help="worker processes of the CPU baseline (default: this lease's CPU share)")
    ap.add_argument("--e2e-sample", type=int, default=None, help="reads in the pinned-batch end-to-end sample (0 disables)")
    ap.add_argument("--e2e-file-sample", type=int, default=None, help="reads in the FASTQ-file end-to-end sample (0 disables)")
    return ap.parse_args()


def self_launch(args) -> None:
    """--gpus N > 1 outside torch.distributed.run: become the launcher (before any GPU / torch import)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    rc = subprocess.call(cmd, env=env, cwd=ROOT)
    sys.exit(rc)


def cpu_share():
    """(cores to use, visible cores, why): the host cores this process may really occupy.  A 1-GPU lease of a
    shared 8-GPU host sees every hardware thread but owns a share of them."""
    visible = os.cpu_count() or 1
    use, why = visible, "os.cpu_count()"
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < use:
            use, why = aff, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < use:
                use, why = q, "cgroup cpu.max"
    except (OSError, ValueError):
        pass
    cap = os.environ.get("SCG_BENCH_CPU_SHARE")
    if cap:
        return max(1, min(use, int(cap))), visible, "SCG_BENCH_CPU_SHARE"
    if why == "os.cpu_count()":
        # no quota is enforced: a lease of g of an 8-GPU host's cards owns g/8 of its hardware threads
        # (256 threads / 8 = 32 per GPU on the MI355X hosts); --cpu-cores overrides
        try:
            import torch
            g = max(1, min(8, torch.cuda.device_count()))
        except Exception:
            g = 8
        if g < 8 and visible >= 64:
            use, why = max(1, visible * g // 8), f"{g}/8 of {visible} visible hardware threads (1-GPU lease of an 8-GPU host)"
    return use, visible, why


def main() -> None:
    args = parse_args()
    self_launch(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one rank per GPU; SCG_BENCH_SHARE_GPU=1 lets several ranks share a card (rehearsal on a 1-GPU box, gloo)
    share = os.environ.get("SCG_BENCH_SHARE_GPU") == "1"
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs a GPU: libscg has no CPU fallback")
    if world > n_dev and not share:
        raise SystemExit(f"--gpus {world} but only {n_dev} visible (SCG_BENCH_SHARE_GPU=1 rehearses ranks on one card over gloo)")
    dev_index = local_rank % n_dev if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # SCG_BENCH_FORCE_DIST=1: the process group and the count reduce also at N = 1, so that the RCCL path (init with a
    # bound device, all_reduce of the bound counters, barrier) can be exercised on a one-GPU box (tests/test_bench_launcher.py)
    force_dist = world == 1 and os.environ.get("SCG_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    use_dist = world > 1 or force_dist
    if use_dist:
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
    reduce_dev = counters if not share else None     # gloo rehearsal reduces a host copy

    def step():
        plan.reset()
        if w.entry == "dual":
            plan.count_paired(mates[0], mates[1], fixed_len1=L, fixed_len2=L, n_pairs=n)
        else:
            plan.count(mates[0], fixed_len=L, n_reads=n)
        if use_dist:
            if reduce_dev is not None:
                dist.all_reduce(reduce_dev, op=dist.ReduceOp.SUM)   # the path's one exchange step
            else:
                h = counters.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                counters.copy_(h)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # clocks: a 70 ms timed region right after start-up runs at boost clocks the chip does not sustain
    settle_steps = 0
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        for _ in range(10):
            step()
        settle_steps += 10
        torch.cuda.synchronize()
    plan.set_profiling(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if not share else "cpu")
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
        # (counting kernel + the tally kernel that turns its index stream into counts)
        traffic, traffic_source, rocprof_ms, rocprof_source = None, None, None, None
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                rec = json.load(f).get(str(args.config))
            if rec and rec.get("bytes_per_read"):
                traffic = round((rec["bytes_per_read"] + rec.get("tally_bytes_per_read", 0.0)) * n)
                traffic_source = rec["source"]
                rocprof_ms = rec.get("rocprof_avg_kernel_ms")       # rocprofv3 --kernel-trace --stats of the same command
                rocprof_source = rec.get("rocprof_source")
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
                       "parallelism": f"read-sharded x{world}" + (" + RCCL all-reduce of counts" if use_dist and not share else "")
                                      + (" (gloo rehearsal, ranks share one GPU)" if share and world > 1 else ""),
                       "inputs": "resident in HBM", "settle_s": args.settle, "settle_steps": settle_steps},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": KERNEL_OF[w.entry], "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": launches,
                         # the committed rocprofv3 summary of this command (full-size runs only), next to the live HIP-event figure
                         "rocprof_avg_kernel_ms": rocprof_ms if n == synth.workload(args.config).n_reads else None,
                         "rocprof_frac": (round(algo_bytes / (rocprof_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5)
                                          if rocprof_ms and n == synth.workload(args.config).n_reads else None),
                         "rocprof_source": rocprof_source,
                         "algorithmic_bytes_per_launch": algo_bytes, "traffic_source": traffic_source,
                         # the whole step (counting kernel + tally / fold kernel [+ all-reduce]) against the same peak
                         "step_achieved": round(algo_bytes / (elapsed / args.steps) / 1e9, 2),
                         "step_frac": round(algo_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5)},
            "mapped_fraction": round(mapped / (n * world), 5),
        }

        sample = args.cpu_sample
        if sample is None:
            sample = min(n, 50_000_000 if w.entry != "dual" else 20_000_000)
        if world == 1 and sample > 0:
            from oracle import cpu_baseline
            sample = min(sample, n)
            cores, visible, why = cpu_share()
            if args.cpu_cores:
                cores, why = args.cpu_cores, "--cpu-cores"
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
                info, cpu_counts, cpu_total = cpu_baseline.run(w, host, cores, workdir)
            finally:
                shutil.rmtree(workdir, ignore_errors=True)
            info["host_cores_visible"] = visible
            info["cores_source"] = why
            info["parity"] = bool(gpu_total == cpu_total and np.array_equal(gpu_counts.astype(np.int64), cpu_counts))
            out["cpu_baseline"] = info
            if not info["parity"]:
                out["cpu_baseline"]["parity_note"] = "GPU counts differ from the CPU reference on the sample"
            del host
        else:
            out["cpu_baseline"] = None

        out["e2e"] = None
        if world == 1:
            try:
                out["e2e"] = end_to_end(sc, synth, torch, np, w, plan, mates, args)
            except Exception as e:      # the PCIe legs are informative; the headline line must still print
                out["e2e"] = {"error": f"{type(e).__name__}: {e}"}
        # north_star's ratio, spelled out: this GPU against the reference CPU path on this lease's own host cores
        # (`vs_baseline` stays null: BASELINE.md holds no published number for the metric)
        cb = out.get("cpu_baseline") or {}
        if cb.get("value"):
            ratios = {"hbm_resident": round(value / cb["value"], 1)}
            for leg, rec in (out.get("e2e") or {}).items():
                if isinstance(rec, dict) and rec.get("value"):
                    ratios[leg] = round(rec["value"] / cb["value"], 2)
            out["vs_cpu_reference"] = {"what": f"Mreads/s over cpu_baseline.value ({cb['value']} on {cb.get('cores')} cores of this box); "
                                               "pinned_batches is SURVEY 8(d) metric 1", **ratios}
        print(json.dumps(out), flush=True)

    plan.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def end_to_end(sc, synth, torch, np, w, plan, mates, args) -> dict:
    """PCIe-inclusive legs on bounded samples of the benchmark stream (never `value`)."""
    L, n = w.read_len, w.n_reads
    unit = "Mpairs/s" if w.entry == "dual" else "Mreads/s"
    res = {}
    # (1) parsed batches in pinned host memory -> counts on the host: H2D copies on two streams overlapped with
    #     the counting kernel, 4 M reads per batch like the file-level stager (SURVEY.md 8d metric 1)
    s1 = args.e2e_sample
    if s1 is None:
        s1 = min(n, 24_000_000 if w.entry != "dual" else 12_000_000)
    s1 = min(s1, n)
    if s1 > 0:
        pinned = [m[: s1 * L].cpu().pin_memory() for m in mates]
        B = 1 << 22
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        bufs = [[torch.empty(B * L, dtype=torch.uint8, device=mates[0].device) for _ in mates] for _ in streams]
        best = None
        for rep in range(3):                        # first pass warms up
            plan.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k, a in enumerate(range(0, s1, B)):
                b = min(a + B, s1)
                st = streams[k % 2]
                with torch.cuda.stream(st):
                    dst = [bufs[k % 2][m][: (b - a) * L] for m in range(len(mates))]
                    for m in range(len(mates)):
                        dst[m].copy_(pinned[m][a * L: b * L], non_blocking=True)
                    if w.entry == "dual":
                        plan.count_paired(dst[0], dst[1], fixed_len1=L, fixed_len2=L, n_pairs=b - a, stream=st)
                    else:
                        plan.count(dst[0], None, fixed_len=L, n_reads=b - a, stream=st)
            for st in streams:
                st.synchronize()
            counts, total = plan.read()
            dt = time.perf_counter() - t0
            if rep and (best is None or dt < best):
                best = dt
        res["pinned_batches"] = {"value": round(s1 / best / 1e6, 2), "unit": unit, "pcie_gbs": round(s1 * w.bytes_per_unit / best / 1e9, 2),
                                 "sample": f"first {s1} of the stream, 4 Mi-read batches, 2 streams, counts copied to the host",
                                 "total": int(total), "mapped": int(counts.astype(np.int64).sum())}
        del pinned, bufs
    # (2) FASTQ file on tmpfs -> counts through the file-level C ABI entry (parse + stage + H2D + kernels + D2H)
    s2 = args.e2e_file_sample
    if s2 is None:
        s2 = min(n, 32_000_000 if w.entry != "dual" else 8_000_000)
    s2 = min(s2, n)
    if s2 > 0:
        d = tempfile.mkdtemp(prefix="scg_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            paths = []
            for m, t in enumerate(mates):
                p = os.path.join(d, f"reads_{m}.fastq")
                synth.reads_to_fastq(p, t[: s2 * L].cpu().numpy(), L)
                paths.append(p)
            size = sum(os.path.getsize(p) for p in paths)

            def call(pools, mm):
                if w.entry == "single":
                    c, t = sc.count_single_barcodes(paths[0], w.template, w.strand, pools[0], mm, w.use_first, 16)
                    return int(c.astype(np.int64).sum()), t
                if w.entry == "combo":
                    _i, f, t = sc.count_combo_barcodes_single(paths[0], w.template, w.strand, pools, mm, w.use_first, 16)
                    return int(f.astype(np.int64).sum()), t
                c, t = sc.count_dual_barcodes(paths[0], w.template, False, mm, pools[0], paths[1], w.template2, False, mm, pools[1],
                                              False, w.use_first, False, 16)
                return int(c.astype(np.int64).sum()), t

            call([p[:16] for p in w.pools], 0)                 # warm-up: context, page cache
            pools_c = [sc.prepare_pool(p) for p in w.pools]    # C string arrays built once (this mirror's marshalling, not the library's work)

            def timed(reps=3):
                """Median of `reps` whole calls (the host side of a shared box is noisy: the spread is reported too).  Every
                call must give the same counts."""
                dts, seen = [], set()
                for _ in range(reps):
                    t0 = time.perf_counter()
                    mapped, total = call(pools_c, w.mismatches)
                    dts.append(time.perf_counter() - t0)
                    seen.add((mapped, total))
                if len(seen) != 1:
                    raise RuntimeError(f"calls on one file disagree: (mapped, total) = {sorted(seen)}")
                dts.sort()
                return dts[len(dts) // 2], dts[0], dts[-1], mapped, total

            def resident_mapped(s):
                """The same first `s` reads counted where they lie in HBM (the path `value` is measured on): what a file leg
                must arrive at.  None for the paired entry."""
                if w.entry == "single":
                    plan = sc.Plan.single(w.template, w.strand, w.pools[0], w.mismatches, w.use_first)
                elif w.entry == "combo":
                    plan = sc.Plan.combo(w.template, w.strand, w.pools[0], w.pools[1], w.mismatches, w.use_first)
                else:
                    return None
                with plan:
                    plan.count(mates[0][: s * L], fixed_len=L, n_reads=s)
                    out = plan.read() if w.entry == "single" else plan.read_combo()
                return int(np.asarray(out[0] if w.entry == "single" else out[1]).astype(np.int64).sum())

            def checked(leg, s, mapped):
                want = resident_mapped(s)
                if want is not None and want != mapped:
                    raise RuntimeError(f"{leg}: {mapped} mapped reads, {want} on the same reads resident in HBM")
                return want is not None

            dt, lo, hi, mapped, total = timed(5)
            res["fastq_file"] = {"value": round(total / dt / 1e6, 2), "unit": unit, "fastq_gbs": round(size / dt / 1e9, 2),
                                 "range": [round(total / hi / 1e6, 2), round(total / lo / 1e6, 2)],
                                 "sample": f"first {s2} of the stream as plain 4-line FASTQ on tmpfs ({size / 1e9:.2f} GB), "
                                           f"one {('scg_count_' + w.entry + '_barcodes')} call incl. library build, median of 5 calls",
                                 "total": int(total), "mapped": mapped, "equals_resident_counts": checked("fastq_file", s2, mapped)}
            if w.entry != "dual":
                # the same reads as BGZF (blocked gzip as written by bgzip): members inflated in parallel by the host threads
                s3 = min(s2, 16_000_000)
                plain = os.path.join(d, "sub.fastq")
                synth.reads_to_fastq(plain, mates[0][: s3 * L].cpu().numpy(), L)
                gz = os.path.join(d, "sub.fastq.gz")
                synth.fastq_to_bgzf(plain, gz, workers=min(16, os.cpu_count() or 1))
                os.remove(plain)
                paths[0] = gz
                gsize = os.path.getsize(gz)
                call([p[:16] for p in w.pools], 0)
                dt, lo, hi, mapped, total = timed()
                res["fastq_bgzf"] = {"value": round(total / dt / 1e6, 2), "unit": unit, "compressed_gbs": round(gsize / dt / 1e9, 2),
                                     "range": [round(total / hi / 1e6, 2), round(total / lo / 1e6, 2)],
                                     "sample": f"first {s3} of the stream as BGZF-compressed FASTQ on tmpfs ({gsize / 1e9:.2f} GB compressed), median of 3 calls",
                                     "total": int(total), "mapped": mapped, "equals_resident_counts": checked("fastq_bgzf", s3, mapped)}
                # ... and as an ordinary single-member gzip file (one zlib stream, level 4, as `gzip` or a sequencer's software
                # writes it): decoded by all host threads at once (csrc/scg_pgzip.h)
                s4 = min(s2, 8_000_000)
                plain = os.path.join(d, "sub4.fastq")
                synth.reads_to_fastq(plain, mates[0][: s4 * L].cpu().numpy(), L)
                gz1 = os.path.join(d, "sub4.fastq.gz")
                import zlib
                comp = zlib.compressobj(4, zlib.DEFLATED, 31)
                with open(plain, "rb") as fin, open(gz1, "wb") as fout:
                    while True:
                        chunk = fin.read(1 << 24)
                        if not chunk:
                            break
                        fout.write(comp.compress(chunk))
                    fout.write(comp.flush())
                os.remove(plain)
                paths[0] = gz1
                call([p[:16] for p in w.pools], 0)
                dt, lo, hi, mapped, total = timed()
                res["fastq_gzip"] = {"value": round(total / dt / 1e6, 2), "unit": unit, "range": [round(total / hi / 1e6, 2), round(total / lo / 1e6, 2)],
                                     "sample": f"first {s4} of the stream as one gzip member on tmpfs ({os.path.getsize(gz1) / 1e9:.2f} GB compressed), "
                                               "decoded in chunks on the device (csrc/scg_dgzip.cpp), median of 3 calls",
                                     "total": int(total), "mapped": mapped, "equals_resident_counts": checked("fastq_gzip", s4, mapped)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return res


if __name__ == "__main__":
    main()
