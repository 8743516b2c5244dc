#!/usr/bin/env python3
"""Prints the per-launch timeline of the last step from a rocprofv3 kernel trace CSV."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_rle_tile_scan' in r['Kernel_Name']]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    m = re.search(r'(k_[a-z0-9_]+(<[^>]*>)?)', r['Kernel_Name'])
    name = m.group(1) if m else r['Kernel_Name'][:40]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d > 0.05:
        print("%8.2f  %7.3f  %s" % ((int(r['Start_Timestamp']) - t0) / 1e6, d, name))
