"""Deterministic test matrices (no files: every matrix named by the reference's docs is absent offline)."""
import numpy as np
import scipy.sparse as sp


def circuit_like(n, seed=0, band=12, scramble=True, unsym_rows=True):
    """ASIC-like stand-in: ~6 entries/row inside a hidden band, heavy value ties (+-1, 0.5), a few
    dense-ish rows/columns removed on purpose so that the band PC is meaningful, diagonally dominant in the hidden
    ordering; then the band is hidden by a symmetric random permutation and the diagonal is destroyed by a row
    permutation (so that a matching is needed to bring large entries back)."""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    tie = np.array([1.0, -1.0, 0.5, -0.5, 0.25])
    for i in range(n):
        k = rng.integers(3, 7)
        offs = rng.choice(np.arange(-band, band + 1), size=k, replace=False)
        s = 0.0
        for o in offs:
            j = i + int(o)
            if j < 0 or j >= n or j == i:
                continue
            v = float(tie[rng.integers(0, len(tie))]) * (1.0 + 1e-3 * rng.standard_normal())
            rows.append(i); cols.append(j); vals.append(v)
            s += abs(v)
        rows.append(i); cols.append(i); vals.append(1.5 * s + 1.0 + 0.01 * rng.random())
    A = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    A.sum_duplicates()
    if scramble:
        q = rng.permutation(n)
        A = A[q][:, q]
    pr = None
    if unsym_rows:
        pr = rng.permutation(n)
        A = A[pr]
    A = A.tocsr()
    A.sort_indices()
    return A
