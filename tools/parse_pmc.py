#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<tag>_pmc_summary.json.

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, section HBM): the counters are in KiB; FETCH_SIZE
reports exactly 1/2 of the bytes of a wide coalesced streaming read, so reads are doubled; WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import sys


def collect(path):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        d[row["Kernel_Name"]].append((float(row["Counter_Value"]),
                                      (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
    return d


def main(fetch_dir, write_dir, out, note=""):
    f = collect(glob.glob(fetch_dir + "/**/*counter_collection.csv", recursive=True)[0])
    w = collect(glob.glob(write_dir + "/**/*counter_collection.csv", recursive=True)[0])
    res = {"note": note, "kernels": {}}
    for k in f:
        if not k.startswith(("spike::", "void spike::", "k_")):
            continue
        fv = sorted(x[0] for x in f[k])
        wv = sorted(x[0] for x in w.get(k, [(0.0, 0.0)]))
        med = lambda v: v[len(v) // 2]
        # per launch = the MEDIAN over the launches of the run: bench.py's warm-up setup (64 K rows) also launches the apply
        # kernels once, which would drag a mean down (the timed launches all have the same size)
        res["kernels"][k] = {
            "launches": len(fv),
            "FETCH_SIZE_KiB_median": med(fv),
            "WRITE_SIZE_KiB_median": med(wv),
            "FETCH_SIZE_KiB_mean": sum(fv) / len(fv),
            "hbm_read_bytes_corrected": 2.0 * 1024.0 * med(fv),
            "hbm_write_bytes": 1024.0 * med(wv),
            "median_duration_us_under_pmc": med(sorted(x[1] for x in f[k])),
        }
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main(*sys.argv[1:])
