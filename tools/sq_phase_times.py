#!/usr/bin/env python3
"""Per-phase durations of k_sq_forward from a rocprofv3 kernel trace taken with VF_SQ_PHASES=1 (one launch per phase).
Usage: sq_phase_times.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "k_sq_forward" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
per = collections.defaultdict(list)
gaps = []
for i, (s, e) in enumerate(rows):
    per[i % 4].append(e - s)
    if i and i % 48:
        gaps.append(s - rows[i - 1][1])
names = ["P1 ln+qkv+attention", "P2 o-proj", "P3 ln+ffn-up", "P4 ffn-down"]
for k in range(4):
    v = sorted(per[k][len(per[k]) // 2:])
    print(names[k], "median ns", v[len(v) // 2], "min", v[0], "n", len(v))
gaps.sort()
print("gap between phase launches: median ns", gaps[len(gaps) // 2] if gaps else None)
