"""per-level trace of the Fiedler ordering at ASIC_320k scale (SPIKE_FIEDLER_TRACE=1): level size, backend, iterations, time"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SPIKE_FIEDLER_TRACE"] = "1"
import numpy as np, scipy.sparse as sp
import spike_petsc_amd.host as H
from matrices import circuit_like
n = 321821
A = circuit_like(n, seed=7, band=24)
perm, *_ = H.mc64_job5(n, A.indptr, A.indices, A.data)
B = A[perm].tocsr(); B.sort_indices()
for dev in (True, False):
    t = time.time(); o, v = H.fiedler_order(n, B.indptr, B.indices, B.data, use_device=dev); print("use_device=%s total %.3f s" % (dev, time.time() - t), flush=True)
