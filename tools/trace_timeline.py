#!/usr/bin/env python3
"""Timeline of the dispatches of a few consecutive batches from a rocprofv3 kernel trace (CSV): start / end of every kernel relative
to the first one shown, so that what a launch interval is made of can be read off.  usage: trace_timeline.py <kernel_trace.csv> [first dispatch] [count]"""
import csv, sys

def short(n):
    for key, s in (("k_scan2", "scan2"), ("k_scan_wide8", "wide8"), ("k_scan_wide", "wide"), ("k_scan<", "scan(sample)"), ("k_scan", "scan"), ("k_sel0", "sel0"), ("k_final", "final"),
                   ("k_prep_queries", "prepq"), ("k_prep_wide8", "prep8"), ("k_merge", "merge"), ("copyBuffer", "copy"), ("fillBuffer", "fill")):
        if key in n:
            return s
    return n[:24]

def main():
    path = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[first:first + count]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"{short(r['Kernel_Name']):14s} q{r.get('Queue_Id', '?'):>3s}  start {s / 1e3:9.1f}  end {e / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f} us")

if __name__ == "__main__":
    main()
