#!/usr/bin/env python3
"""Assemble profiles/<tag>_* from gpurun_out/profiles_<tag> (run after tools/make_profiles.sh)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kernel_sub = sys.argv[3] if len(sys.argv) > 3 else "staged_kernel"
root = os.path.join("gpurun_out", f"profiles_{tag}")
os.makedirs("profiles", exist_ok=True)

stats = glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_config{cfg}_kernel_stats.csv")
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
for a, label in ((2, "stream_only"), (0, "full")):
    d = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum_TCC_HIT_sum_TCC_MISS_sum"):
        d.update(counters(a, name))
    out[label] = d
cal = out["stream_only"].get("FETCH_SIZE", 0) * 1024 / algo if algo else 0
out["calibration"] = {
    "what": "SCG_ABLATE=2 runs phase A alone: the kernel then reads exactly the algorithmic bytes with the same 16-B coalesced loads",
    "FETCH_SIZE_bytes_over_true_bytes": round(cal, 4),
    "note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies the 128-B requests of a wide coalesced stream at 64 B; confirmed here (ratio ~0.5)",
}
full = out["full"]
fetch_bytes = full.get("FETCH_SIZE", 0) * 1024
# the streaming requests are under-counted by half of the algorithmic bytes; everything else (index lookups) is tallied as is
read_bytes = fetch_bytes + algo * (1 - cal) if cal else None
write_bytes = full.get("WRITE_SIZE", 0) * 1024
out["traffic"] = {"read_bytes_per_launch": read_bytes, "write_bytes_per_launch": write_bytes,
                  "total_bytes_per_launch": (read_bytes + write_bytes) if read_bytes is not None else None,
                  "bytes_per_read": ((read_bytes + write_bytes) / n) if read_bytes is not None else None,
                  "vs_algorithmic": ((read_bytes + write_bytes) / algo) if read_bytes is not None else None}
# the tally kernel that follows the counting kernel in tally mode (ScgCounters::unit_index): its own HBM traffic
tally = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    tally.update(counters(0, name, "tally_kernel"))
if tally:
    out["tally_kernel"] = {"FETCH_SIZE_KB": tally.get("FETCH_SIZE"), "WRITE_SIZE_KB": tally.get("WRITE_SIZE"),
                           "note": "index stream read (4 B per read and pass, coalesced: FETCH_SIZE tallies it at x0.5) + flush atomics"}
json.dump(out, open(f"profiles/{tag}_config{cfg}_traffic.json", "w"), indent=1)
# the figure bench.py reports as roofline.traffic
tj = "profiles/traffic.json"
cur = json.load(open(tj)) if os.path.exists(tj) else {}
cur[str(cfg)] = {"bytes_per_read": out["traffic"]["bytes_per_read"], "source": f"profiles/{tag}_config{cfg}_traffic.json",
                 "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE corrected with the streaming-only calibration run"}
json.dump(cur, open(tj, "w"), indent=1)
print(json.dumps(out["traffic"], indent=1))
print(open(f"profiles/{tag}_config{cfg}_kernel_stats.csv").read() if stats else "no stats")
