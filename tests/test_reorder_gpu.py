"""The reordering front-end on the device (csrc/spike_reorder.hip): integer / byte work, so the bar is
IDENTITY with the host loops -- MatPermute / VecPermute of the host mirror (csrc/host/sp_host.c) and the sequential AWBM
(csrc/host/awbm.c, itself pinned to an independent restatement in tests/test_host_cpu.py)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from matrices import circuit_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from conftest import _ensure_built
    _ensure_built()
    import spike_petsc_amd.host as H
    H.chk(H.lib().SpikePetscRegisterAll())
    return H


def _host_permute(A, rowp, colp):
    """what sp_host.c:MatPermute does: row i <- row rowp[i], column c -> position of c in colp, rows sorted by column"""
    icol = np.empty(len(colp), dtype=np.int64)
    icol[colp] = np.arange(len(colp))
    B = A[rowp]
    B = sp.csr_matrix((B.data, icol[B.indices], B.indptr), shape=A.shape)
    B.sort_indices()
    return B


@pytest.mark.parametrize("n,seed", [(5000, 1), (321821, 7)])
def test_device_matpermute_is_the_host_matpermute(spike, H, n, seed):
    A = circuit_like(n, seed=seed)
    rng = np.random.default_rng(seed)
    rowp, colp = rng.permutation(n), rng.permutation(n)
    ib, jb, b = spike.permute_csr(A, rowp, colp)
    B = _host_permute(A, rowp, colp)
    assert np.array_equal(ib, B.indptr) and np.array_equal(jb, B.indices) and np.array_equal(b.view(np.uint64), B.data.view(np.uint64))
    # through the mirror's MatPermute with the option on and off: the same Mat
    out = {}
    for dev in (1, 0):
        H.options(mat_permute_device=dev)
        M = H.Mat.from_scipy(A)
        r, c, PM = C.c_void_p(), C.c_void_p(), C.c_void_p()
        i64p = C.POINTER(C.c_int64)
        H.chk(H.lib().ISCreateGeneral(n, rowp.astype(np.int64).ctypes.data_as(i64p), C.byref(r)))
        H.chk(H.lib().ISCreateGeneral(n, colp.astype(np.int64).ctypes.data_as(i64p), C.byref(c)))
        H.chk(H.lib().MatPermute(M.h, r, c, C.byref(PM)))
        out[dev] = H.Mat(handle=PM).to_scipy()
        H.chk(H.lib().MatDestroy(C.byref(PM))); H.chk(H.lib().ISDestroy(C.byref(r))); H.chk(H.lib().ISDestroy(C.byref(c)))
    for dev in (1, 0):
        assert np.array_equal(out[dev].indptr, B.indptr) and np.array_equal(out[dev].indices, B.indices) and np.array_equal(out[dev].data, B.data)
    # not a permutation: refused
    bad = colp.copy(); bad[0] = bad[1]
    with pytest.raises(spike.SpikeError):
        spike.permute_csr(A, rowp, bad)


def test_device_vecpermute_both_directions(spike):
    import torch
    n = 1 << 20
    rng = np.random.default_rng(0)
    idx = rng.permutation(n)
    x = rng.standard_normal(n)
    y = spike.permute_vec(x, idx, inverse=False)
    assert np.array_equal(y, x[idx])                          # VecPermute(x, is, PETSC_FALSE), kspreorder.c:122-123
    z = spike.permute_vec(y, idx, inverse=True)
    assert np.array_equal(z, x)                               # ... and PETSC_TRUE undoes it (:126-127)
    xd, idd = torch.from_numpy(x).cuda(), torch.from_numpy(idx).cuda()
    yd = spike.permute_vec(xd, idd, inverse=False)
    assert np.array_equal(yd.cpu().numpy(), x[idx])
    assert np.array_equal(spike.permute_vec(yd, idd, inverse=True).cpu().numpy(), x)


@pytest.mark.parametrize("n,seed,kind", [(3000, 3, "circuit"), (30000, 11, "circuit"), (321821, 7, "circuit"), (4000, 5, "ties"), (6000, 2, "dense_rows")])
def test_device_awbm_is_the_sequential_awbm(spike, H, n, seed, kind):
    """phases 1 and 3 as a parallel fixed point == the sequential greedy: the SAME permutation as csrc/host/awbm.c on the
    circuit-like family (config-4 size included), on a tie-heavy matrix (values from a 4-element set: many tight edges compete for
    the same rows, long displacement chains) and with a few dense rows"""
    rng = np.random.default_rng(seed)
    if kind == "circuit":
        A = circuit_like(n, seed=seed)
    elif kind == "ties":
        R = sp.random(n, n, density=6.0 / n, random_state=seed, data_rvs=lambda k: rng.choice([1.0, -1.0, 0.5, 2.0], k))
        A = sp.csr_matrix(R + sp.eye(n, format="csr")[rng.permutation(n)])
    else:
        R = sp.random(n, n, density=4.0 / n, random_state=seed, data_rvs=lambda k: rng.choice([1.0, 0.5, 0.25], k)).tolil()
        for r in rng.choice(n, 5, replace=False):
            R[r, rng.choice(n, n // 3, replace=False)] = 1.0
        A = sp.csr_matrix(sp.csr_matrix(R) + sp.eye(n, format="csr")[rng.permutation(n)])
    A = sp.csr_matrix(A)
    A.sort_indices()
    ph = H.awbm(n, A.indptr, A.indices, A.data)
    pd, rounds = spike.awbm_device(n, A.indptr, A.indices, A.data)
    assert np.array_equal(pd, ph)
    assert sorted(pd.tolist()) == list(range(n))
    assert 1 <= rounds[0] < n and 1 <= rounds[1] < n
    print("awbm n=%d %s: fixed-point rounds phase 1 %d, phase 3 %d" % (n, kind, rounds[0], rounds[1]))
    # through the mirror's ordering with the device option on / off
    for dev in (1, 0):
        H.options(mat_awbm_device=dev)
        M = H.Mat.from_scipy(A)
        r, c = C.c_void_p(), C.c_void_p()
        H.chk(H.lib().MatGetOrdering(M.h, b"awbm", C.byref(r), C.byref(c)))
        assert np.array_equal(H.is_indices(r), ph) and np.array_equal(H.is_indices(c), np.arange(n))
        H.chk(H.lib().ISDestroy(C.byref(r))); H.chk(H.lib().ISDestroy(C.byref(c)))
