#!/usr/bin/env python3
"""Copy the judged summaries of one tools/r2_profile.sh run from gpurun_out/prof_<tag>/ (scratch) into profiles/ (tracked).
usage: tools/collect_profiles.py <tag>"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
dst = "profiles"
for name in ["bench", "n2097152", "n1048576", "n524288", "c2", "fiedler"]:
    # the flat copy made ON the box first: the local stats_* directories accumulate the runs of earlier captures
    f = glob.glob(os.path.join(src, name + "_kernel_stats.csv")) or glob.glob(os.path.join(src, "stats_" + name, "*", "*_kernel_stats.csv"))
    if f:
        shutil.copy(f[0], os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
for name in ["bench_line.json", "bench_line_under_rocprof.json", "pmc_summary.json", "rank_n2097152.json", "rank_n1048576.json",
             "rank_n524288.json", "c2.json", "rank_n524288_rccl_overlap.json", "fiedler_trace.log", "config4_timing.log",
             "setup_trace_k128.log", "setup_trace_k256.log", "c3.json", "headline_p8.json", "headline_p64.json",
             "rank_n524288_rccl_serial.json", "ab_rank_n524288.log", "ab_headline.log", "summary.txt", "capture_commit.txt"]:
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name)))
pm = os.path.join(src, "pmc_summary.json")
if os.path.exists(pm):
    d = json.load(open(pm))
    tot = 0.0
    ks = []
    # the PCApply sweeps: tag 0; of the shapes that appear (the setup-time measurement launches every candidate a few times) the
    # forward and the backward kernel with the most launches
    for rev in ("false, 0>(spike::SweepArgs)", "true, 0>(spike::SweepArgs)"):
        c = [(v["launches"], k) for k, v in d["kernels"].items() if "k_sweep<" in k and k.rstrip().endswith(rev)]
        if c:
            k = max(c)[1]
            v = d["kernels"][k]
            tot += v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
            ks.append(k)
    line = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
    commit = open(os.path.join(src, "capture_commit.txt")).read().strip() if os.path.exists(os.path.join(src, "capture_commit.txt")) else "unknown"
    json.dump({"N": line["config"]["N"], "K": line["config"]["K"], "P": line["config"]["partitions"], "traffic_per_pass_bytes": tot,
               "commit": commit,
               "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; reads x2 per "
                         "MI355X_MICROARCH.md HBM section); kernels: %s" % (tag, " + ".join(ks))},
              open(os.path.join(dst, "pmc_current.json"), "w"), indent=1)
print(sorted(os.listdir(dst)))
