#!/usr/bin/env python3
"""Per-kernel time of the setup phase (from k_gen_band to the first PCApply sweep) in a rocprofv3 --kernel-trace CSV."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
i0 = [i for i, e in enumerate(ev) if 'k_gen_band' in e[2]][0]
i1 = [i for i, e in enumerate(ev) if ('k_sweep<' in e[2] and ', 0>' in e[2]) or 'k_scan_sweep<' in e[2] and ', 0>' in e[2]][0]
tot, cnt = collections.Counter(), collections.Counter()
for s, e, n in ev[i0:i1]:
    tot[n[:70]] += e - s
    cnt[n[:70]] += 1
print("setup span %.2f ms, kernels busy %.2f ms" % ((ev[i1][0] - ev[i0][0]) / 1e6, sum(tot.values()) / 1e6))
for k, v in tot.most_common(16):
    print("%8.2f ms %5d  %s" % (v / 1e6, cnt[k], k))
