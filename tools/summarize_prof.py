#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/prof_<tag>) into small summaries under profiles/.
  profiles/<tag>_kernel_stats.csv   : the --stats table (per-kernel calls / total / average)
  profiles/<tag>_pmc.json           : per-kernel FETCH_SIZE / WRITE_SIZE sums and per-launch HBM bytes
  profiles/pmc_traffic.json         : {pooled kernel name: HBM bytes per launch} read by bench.py
FETCH_SIZE is doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md "HBM");
units are KiB per the counter definition."""
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
traffic_name = sys.argv[2] if len(sys.argv) > 2 else "pmc_traffic.json"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def pooled(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name


stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows:
            w.writerow([r.get("Name"), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")])
    pool = {}
    for r in rows:
        p = pool.setdefault(pooled(r["Name"]), [0, 0])
        p[0] += int(r["Calls"])
        p[1] += int(r["TotalDurationNs"])
    with open(os.path.join(dst, tag + "_kernel_stats_pooled.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel(all template instances)", "Calls", "TotalMs", "AverageMs"])
        for k, (c, t) in sorted(pool.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, c, round(t / 1e6, 3), round(t / 1e6 / c, 4)])
    print("wrote kernel stats:", len(rows), "kernels")

pmc = {}
for which, scale in (("fetch", 2.0), ("write", 1.0)):
    files = glob.glob(os.path.join(src, "pmc_" + which, "**", "*counter_collection.csv"), recursive=True)
    for fn in files:
        for r in csv.DictReader(open(fn)):
            k = pooled(r["Kernel_Name"])
            d = pmc.setdefault(k, {"launches_fetch": 0, "launches_write": 0, "FETCH_SIZE_KiB_x2": 0.0, "WRITE_SIZE_KiB": 0.0})
            v = float(r["Counter_Value"])
            if r["Counter_Name"] == "FETCH_SIZE":
                d["FETCH_SIZE_KiB_x2"] += v * scale
                d["launches_fetch"] += 1
            elif r["Counter_Name"] == "WRITE_SIZE":
                d["WRITE_SIZE_KiB"] += v
                d["launches_write"] += 1
if pmc:
    traffic = {}
    for k, d in pmc.items():
        lf, lw = max(d["launches_fetch"], 1), max(d["launches_write"], 1)
        d["hbm_bytes_per_launch"] = int(d["FETCH_SIZE_KiB_x2"] * 1024 / lf + d["WRITE_SIZE_KiB"] * 1024 / lw)
        traffic[k] = d["hbm_bytes_per_launch"]
    # whole-step totals (the PMC passes run ONE step): what bench.py reports as HBM bytes per input byte
    total = sum(d["FETCH_SIZE_KiB_x2"] + d["WRITE_SIZE_KiB"] for k, d in pmc.items() if k.startswith("k_")) * 1024
    traffic["__total_bytes_per_step"] = int(total)
    traffic["__source"] = tag + "_pmc.json"
    # the kernel sources the counters belong to (run this right behind the profile, on the tree that was profiled):
    # bench.py compares it with the tree it runs on and says in its line whether the committed traffic is that of its kernels
    import hashlib
    hh = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(root, "rust-compression_amd", "csrc", "k_*.hip")) + [os.path.join(root, "rust-compression_amd", "csrc", "bzgpu.h")]):
        hh.update(open(fn, "rb").read())
    traffic["__kernel_sources_sha16"] = hh.hexdigest()[:16]
    for f in ("bench_fetch.json",):
        pth = os.path.join(src, f)
        try:
            line = [x for x in open(pth).read().splitlines() if x.startswith("{")][-1]
            b = json.loads(line)
            cfg = b.get("config", {})
            if "input_bytes" in cfg:
                traffic["__input_bytes"] = cfg["input_bytes"]
            elif "mib" in cfg:
                traffic["__input_bytes"] = cfg["mib"] << 20
            else:
                m = re.match(r"(\d+) MiB", str(cfg.get("workload", "")))
                traffic["__input_bytes"] = (int(m.group(1)) << 20) if m else None
        except Exception:
            pass
    json.dump(pmc, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    json.dump(traffic, open(os.path.join(dst, traffic_name), "w"), indent=1, sort_keys=True)
    print("wrote pmc for", len(pmc), "kernels")
for f in ("bench_trace.json", "bench_fetch.json", "bench_write.json"):
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p):
        open(os.path.join(dst, tag + "_" + f), "w").write(open(p).read())
