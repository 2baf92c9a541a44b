#!/usr/bin/env python3
"""Per-layer kernel sequence of an encoder forward from a rocprofv3 kernel trace (CSV): the dispatches of the LAST forward in
the trace, grouped by position inside the layer (the two residual products share a kernel name; their order tells them apart).
usage: trace_layer.py <kernel_trace.csv> <layers>"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void vft::", "").replace("vft::", "")
    return n[:70]


def main():
    path, layers = sys.argv[1], int(sys.argv[2])
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    # the last forward: from the last k_embed_ln on
    start = max(i for i, k in enumerate(ks) if "k_embed_ln" in k[0])
    fwd = ks[start:]
    body = [k for k in fwd if "k_embed_ln" not in k[0] and "k_position_ids" not in k[0] and "k_pool" not in k[0] and "copy" not in k[0].lower()]
    per = len(body) // layers
    print(f"{len(fwd)} dispatches in the last forward, {per} per layer; wall {(fwd[-1][3] - fwd[0][2]) / 1e3:.0f} us")
    acc = defaultdict(list)
    for i, k in enumerate(body[:per * layers]):
        acc[(i % per, k[0])].append(k[1])
    tot = 0.0
    for (pos, name), v in sorted(acc.items()):
        m = sum(v) / len(v)
        tot += m
        print(f"  {pos:2d} {name:72s} {m:8.1f} us  (min {min(v):.1f} max {max(v):.1f})")
    print(f"  sum of per-layer kernel times {tot:.1f} us  x {layers} = {tot * layers / 1e3:.2f} ms")
    gaps = [(body[i + 1][2] - body[i][3]) / 1e3 for i in range(len(body) - 1)]
    print(f"  gaps between consecutive kernels: mean {sum(gaps) / len(gaps):.2f} us, total {sum(gaps):.0f} us")


if __name__ == "__main__":
    main()
