#!/usr/bin/env python3
"""The launches of the LAST bz_encode_buffer call of a rocprofv3 --kernel-trace CSV of tools/small_run.py: when each kernel
began (µs from the call's first kernel), how long it ran, the idle gap in front of it; totals at the end.
usage: tools/small_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_rle_tile_scan' in r['Kernel_Name']]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
prev_end = t0
busy = 0
gaps = []
for r in rows[start:]:
    m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name'])
    name = m.group(1) if m else r['Kernel_Name'][:40]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3
    busy += (e - s) / 1e3
    gaps.append((gap, name))
    print("%9.1f us  run %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name))
    prev_end = max(prev_end, e)
n = len(rows) - start
print("launches %d, span %.1f us, kernels busy %.1f us, idle %.1f us" % (n, (prev_end - t0) / 1e3, busy, (prev_end - t0) / 1e3 - busy))
print("largest gaps:", sorted(gaps, reverse=True)[:12])
