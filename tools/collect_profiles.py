#!/usr/bin/env python3
"""Assemble profiles/<tag>_config<N>_* from gpurun_out/profiles_<tag>_config<N> (made by tools/make_profiles.sh).

HBM-side traffic: FETCH_SIZE / WRITE_SIZE (KB) from their own --pmc passes.  MI355X_MICROARCH.md notes that on gfx950
FETCH_SIZE can tally the 128-B requests of a wide coalesced stream at 64 B; whether it does is MEASURED here, not assumed:
the -DSCG_ABLATE measurement build run with SCG_ABLATE=2 executes phase A alone, i.e. reads exactly the algorithmic bytes
with the very same 16-byte non-temporal loads, and its FETCH_SIZE / algorithmic-bytes ratio is the calibration factor
applied to the streaming share of the full kernel (index lookups are scattered 16-byte requests and taken as tallied).
TCC_EA0_RDREQ (and its 32-B subset) is recorded next to it so that the two views can be compared."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kernel_sub = {2: "single_staged", 5: "single_staged", 3: "combo_staged", 4: "dual_passes"}[cfg]
root = os.path.join("gpurun_out", f"profiles_{tag}_config{cfg}")
os.makedirs("profiles", exist_ok=True)

stats = glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)
rocprof_avg_ms = None
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_config{cfg}_kernel_stats.csv")
    for row in csv.DictReader(open(stats[0])):
        if kernel_sub in row["Name"]:
            rocprof_avg_ms = float(row["AverageNs"]) / 1e6          # (warm-up launches included: 23 launches of the bench command)
if os.path.exists(os.path.join(root, "pmc_summary.txt")):
    shutil.copy(os.path.join(root, "pmc_summary.txt"), f"profiles/{tag}_config{cfg}_pmc_summary.txt")
bench = open(os.path.join(root, "bench_default.log")).read().strip().splitlines()[-1]
open(f"profiles/{tag}_config{cfg}_bench.json", "w").write(bench + "\n")
line = json.loads(bench)
n = line["config"]["reads_per_gpu"]
algo = line["roofline"]["algorithmic_bytes_per_launch"]


def counters(a, name, sub=None):
    sub = sub or kernel_sub
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, f"pmc_a{a}_{name}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if sub in row["Kernel_Name"]:
                acc[(row["Counter_Name"], row["Dispatch_Id"])].append(float(row["Counter_Value"]))
    per = defaultdict(list)
    for (c, d), v in acc.items():
        per[c].append(sum(v))
    return {c: sum(v) / len(v) for c, v in per.items()}


out = {"tag": tag, "config": cfg, "reads_per_launch": n, "algorithmic_bytes_per_launch": algo, "kernel": kernel_sub}
names = ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum_TCC_EA0_RDREQ_32B_sum_TCC_HIT_sum_TCC_MISS_sum")
for a, label in ((2, "stream_only"), (0, "full")):
    d = {}
    for name in names:
        d.update(counters(a, name))
    if d:
        out[label] = d
        if "TCC_EA0_RDREQ_sum" in d:
            r32 = d.get("TCC_EA0_RDREQ_32B_sum", 0.0)
            d["rdreq_bytes"] = (d["TCC_EA0_RDREQ_sum"] - r32) * 64 + r32 * 32
cal = None
if "stream_only" in out and out["stream_only"].get("FETCH_SIZE"):
    cal = out["stream_only"]["FETCH_SIZE"] * 1024 / algo
else:                                   # combination / dual kernels share the staging loop: take config 2's calibration of this tag
    try:
        cal = json.load(open(f"profiles/{tag}_config2_traffic.json"))["calibration"]["FETCH_SIZE_bytes_over_true_bytes"]
    except (OSError, KeyError, ValueError):
        cal = None
out["calibration"] = {
    "what": "SCG_ABLATE=2 (measurement build) runs phase A alone: exactly the algorithmic bytes through the same 16-B non-temporal loads",
    "FETCH_SIZE_bytes_over_true_bytes": None if cal is None else round(cal, 4),
    "stream_only_rdreq_bytes_over_true_bytes": (round(out["stream_only"]["rdreq_bytes"] / algo, 4)
                                                if "stream_only" in out and "rdreq_bytes" in out["stream_only"] else None),
}
full = out.get("full", {})
fetch_bytes = full.get("FETCH_SIZE", 0) * 1024
read_bytes = fetch_bytes + algo * (1 - cal) if cal else None
write_bytes = full.get("WRITE_SIZE", 0) * 1024
out["traffic"] = {"read_bytes_per_launch": read_bytes, "write_bytes_per_launch": write_bytes,
                  "total_bytes_per_launch": (read_bytes + write_bytes) if read_bytes is not None else None,
                  "bytes_per_read": ((read_bytes + write_bytes) / n) if read_bytes is not None else None,
                  "vs_algorithmic": ((read_bytes + write_bytes) / algo) if read_bytes is not None else None,
                  "rdreq_view_read_bytes": full.get("rdreq_bytes")}
tally = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    tally.update(counters(0, name, "tally_kernel"))
tally_bytes_per_read = 0.0
if tally:
    out["tally_kernel"] = {"FETCH_SIZE_KB": tally.get("FETCH_SIZE"), "WRITE_SIZE_KB": tally.get("WRITE_SIZE")}
    # the tally kernel's one pass over the 4-byte index stream (16-byte loads like the staging loop: same calibration) and its counter updates
    fetch = (tally.get("FETCH_SIZE") or 0.0) * 1024
    tally_bytes_per_read = ((fetch / cal if cal else fetch) + (tally.get("WRITE_SIZE") or 0.0) * 1024) / n
    out["tally_kernel"]["bytes_per_read"] = tally_bytes_per_read
out["rocprof_avg_kernel_ms"] = rocprof_avg_ms
json.dump(out, open(f"profiles/{tag}_config{cfg}_traffic.json", "w"), indent=1)
tj = "profiles/traffic.json"
cur = json.load(open(tj)) if os.path.exists(tj) else {}
if out["traffic"]["bytes_per_read"]:
    cur[str(cfg)] = {"bytes_per_read": out["traffic"]["bytes_per_read"], "tally_bytes_per_read": tally_bytes_per_read,
                     "rocprof_avg_kernel_ms": rocprof_avg_ms, "rocprof_source": f"profiles/{tag}_config{cfg}_kernel_stats.csv",
                     "source": f"profiles/{tag}_config{cfg}_traffic.json",
                     "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE corrected with the measured streaming-only calibration"}
    json.dump(cur, open(tj, "w"), indent=1)
print(json.dumps(out["calibration"], indent=1))
print(json.dumps(out["traffic"], indent=1))
print(open(f"profiles/{tag}_config{cfg}_kernel_stats.csv").read()[:1500] if stats else "no stats")
