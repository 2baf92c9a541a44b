#!/usr/bin/env python3
"""Launch interval of a kernel from a rocprofv3 kernel trace (CSV): (last end - first start) / dispatches, next to the
average dispatch duration.  With overlapping dispatches (small shards: CU split + overlapping main scans) the interval is
what a launch costs; bench.py's roofline uses the same definition.  usage: trace_span.py <kernel_trace.csv> <name substring> [skip [count]]
(skip = the warm-up dispatches, count = the timed ones: what follows the timed loop in a bench run -- verification, the host-buffer entry,
one call at a time -- is not part of the interval)"""
import csv
import sys


def main():
    path, needle = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = [r for r in csv.DictReader(open(path)) if needle in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[skip:]
    if len(sys.argv) > 4:
        rows = rows[:int(sys.argv[4])]
    if not rows:
        print("no dispatch of", needle)
        return
    st = [int(r["Start_Timestamp"]) for r in rows]
    en = [int(r["End_Timestamp"]) for r in rows]
    dur = [e - s for s, e in zip(st, en)]
    overl = sum(1 for i in range(1, len(rows)) if st[i] < max(en[:i]))
    print(f"{needle}: dispatches {len(rows)} (first {skip} skipped)  avg duration {sum(dur) / len(dur) / 1e3:.1f} us  "
          f"launch interval (makespan / n) {(max(en) - st[0]) / len(rows) / 1e3:.1f} us  dispatches that start before an earlier one ended: {overl}")


if __name__ == "__main__":
    main()
