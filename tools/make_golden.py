#!/usr/bin/env python3
"""Writes the small fixtures under tests/golden/.

Provenance (read this before trusting them):
  * mc64_wbm_3x3.json   -- input = the 3x3 matrix hard-coded in /root/reference/src/wbm.c:485-497; expected output =
                           what the reference's own HSLmc64AD(job 5) returned for it as recorded in SURVEY.md section 4
                           (perm = [3,1,2], num = 3, u = [0,0,ln 2], v = [-ln 8,-ln 2,-ln 4]).  The only reference-run
                           vector that exists for this repo.
  * spike_*.npz         -- produced by THIS repo's CPU oracle (oracle/spike_oracle.c), itself pinned against LAPACK in
                           tests/test_oracle.py.  They are regression vectors for the oracle and a second, file-based
                           check for the GPU path; they are NOT reference outputs (the reference has none: SURVEY 8c).
Run from the repo root:  python tools/make_golden.py
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)

json.dump({
    "source_input": "/root/reference/src/wbm.c:485-497 (rows r0={(1,8),(2,3)}, r1={(1,2),(2,1)}, r2={(0,4)})",
    "source_output": "SURVEY.md section 4 item 1: reference HSLmc64AD(job=5) run on these arrays (CSR handed to the CSC interface)",
    "n": 3, "ia": [0, 2, 4, 5], "ja": [1, 2, 1, 2, 0], "a": [8.0, 3.0, 2.0, 1.0, 4.0],
    "perm_1based": [3, 1, 2], "num": 3,
    "u": [0.0, 0.0, math.log(2.0)], "v": [-math.log(8.0), -math.log(2.0), -math.log(4.0)],
}, open(os.path.join(G, "mc64_wbm_3x3.json"), "w"), indent=1)

for name, (N, K, P, delta) in {"spike_n4096_k8_p4": (4096, 8, 4, 0.8), "spike_n8192_k40_p8": (8192, 40, 8, 1.2),
                               "spike_n16384_k1_p4": (16384, 1, 4, 1.2)}.items():
    band = O.gen_band(N, K, seed=12345, delta=delta)
    f = O.gen_vec(N, seed=54321)
    sp = O.Spike(band, P)
    idx = np.unique(np.concatenate([np.arange(0, N, 97), sp.starts()[1:-1] - 1, sp.starts()[1:-1]]))
    np.savez_compressed(os.path.join(G, name + ".npz"), N=N, K=K, P=P, delta=delta, seed=12345, rhs_seed=54321, idx=idx,
                        x_coupled=sp.apply(f, 1)[idx], x_decoupled=sp.apply(f, 0)[idx])
print("wrote", sorted(os.listdir(G)))
