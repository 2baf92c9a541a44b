#!/usr/bin/env python3
"""HBM bytes per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE counter_collection CSVs, taken in
separate runs with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) -> the JSON record bench.py reads.
usage: pmc_traffic.py FETCH.csv WRITE.csv KERNEL_SUBSTRING rows dim batch k algorithmic_bytes out.json [note]"""
import csv, json, sys


def mean_of(path, ctr, sub):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r.get("Counter_Name") == ctr and sub in r["Kernel_Name"]]
    if not v:
        raise SystemExit(f"no {ctr} rows for a kernel containing {sub!r} in {path}")
    name = next(r["Kernel_Name"] for r in csv.DictReader(open(path)) if sub in r["Kernel_Name"])
    return sum(v) / len(v), len(v), name


def main(a):
    fetch, write, sub = a[0], a[1], a[2]
    rows, dim, batch, k, alg = int(a[3]), int(a[4]), int(a[5]), int(a[6]), int(a[7])
    f, nf, name = mean_of(fetch, "FETCH_SIZE", sub)
    w, nw, _ = mean_of(write, "WRITE_SIZE", sub)
    hbm = f * 1024 * 2 + w * 1024
    rec = {"kernel": name[:120], "workload": {"rows": rows, "dim": dim, "batch": batch, "k": k, "n_gpus": 1},
           "FETCH_SIZE_mean_KB": f, "WRITE_SIZE_mean_KB": w, "launches": min(nf, nw),
           "correction": "hbm_bytes = FETCH_SIZE*1024*2 (gfx950 reports half of a wide coalesced streaming read, LDS-DMA loads included) + "
                         "WRITE_SIZE*1024; separate --pmc passes with --kernel-trace only",
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": hbm / alg, "measured": a[9] if len(a) > 9 else "round 5"}
    json.dump(rec, open(a[8], "w"), indent=1)
    print(a[8], "ratio", round(hbm / alg, 4), "launches", rec["launches"], name[:60])


if __name__ == "__main__":
    main(sys.argv[1:])
