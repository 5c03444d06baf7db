#!/usr/bin/env python3
"""one line per bench.py JSON file given on the command line"""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c, r = d["config"], d["roofline"]
        print("%-34s N=%-8d K=%-3d P=%-5d rows/part=%-6d m=%-4d p=%d  ms/step=%.4f med=%.4f min=%.4f  GB/s=%-5.0f pass_ms=%.4f frac=%.3f ceil=%.0f setup=%.3f err=%.1e"
              % (f.split("/")[-1], c["N"], c["K"], c["partitions"], c["rows_per_partition"], c["stored_spike_rows"], c["passes_over_factors"],
                 d["ms_per_step"], d["apply_ms_median_device"], d["apply_ms_min_device"], d["value"], r["pass_ms"], r["frac"],
                 r["measured_read_ceiling_GBps"], d["setup_s"], d["max_abs_error_vs_exact_solution"]))
    except Exception as e:
        print(f, "failed:", e)
