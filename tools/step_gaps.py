#!/usr/bin/env python3
"""Idle GPU time inside one step, from a rocprofv3 --kernel-trace CSV: tools/step_gaps.py <kernel_trace.csv> [top]
The LAST step of the run (from its first k_rle_tile_scan to its last k_frame); a gap = time during which no kernel
of the process ran; listed with the kernels on either side (host round trips show as gaps behind a copy)."""
import csv
import re
import sys


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    def nm(r):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        return m.group(1) if m else r["Kernel_Name"][:28]
    names = [nm(r) for r in rows]
    i1 = max(i for i, n in enumerate(names) if n == "k_frame")
    i0 = max(i for i, n in enumerate(names[:i1]) if n == "k_rle_tile_scan")
    t0, t1 = int(rows[i0]["Start_Timestamp"]), int(rows[i1]["End_Timestamp"])
    cur, gaps, busy = t0, [], {}
    for i in range(i0, i1 + 1):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        if s > cur:
            gaps.append((s - cur, names[i - 1], names[i]))
        cur = max(cur, e)
        busy[names[i]] = busy.get(names[i], 0) + (e - s)
    idle = sum(g[0] for g in gaps)
    print("step: %.3f ms from the first k_rle_tile_scan to the end of k_frame, %d kernels, idle %.3f ms in %d gaps"
          % ((t1 - t0) / 1e6, i1 - i0 + 1, idle / 1e6, len(gaps)))
    for g in sorted(gaps, reverse=True)[:top]:
        print("  %7.1f us  after %-28s before %s" % (g[0] / 1e3, g[1], g[2]))
    print("kernel time by name (ms):")
    for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:30]:
        print("  %-30s %8.3f" % (k, v / 1e6))


if __name__ == "__main__":
    main()
