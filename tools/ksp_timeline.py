#!/usr/bin/env python3
"""Timeline of GMRES iterations from a rocprofv3 --kernel-trace CSV: kernels, durations and idle gaps between two
band mat-vecs in the middle of the run.  usage: ksp_timeline.py <kernel_trace.csv> [first_matvec_index] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
idx = [i for i, e in enumerate(ev) if 'band_matvec' in e[2]]
a0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 2
a, b = idx[a0], idx[a0 + cnt]
t0, prev = ev[a][0], None
busy = 0
for s, e, n in ev[a:b]:
    gap = (s - prev) / 1e3 if prev else 0.0
    busy += e - s
    print("%9.1f us  dur %8.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n[:70]))
    prev = e
tot = ev[b][0] - t0
print("per iteration: %.1f us, busy %.1f us, idle %.1f us" % (tot / 1e3 / cnt, busy / 1e3 / cnt, (tot - busy) / 1e3 / cnt))
