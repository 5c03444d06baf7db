"""CPU tests of the oracle itself: it is pinned against LAPACK (scipy.linalg.solve_banded) because the
reference holds no golden vector for the factor/solve arithmetic ('parity unpinned' at the
reference boundary), and against hand-worked band-extraction cases that follow
/root/reference/src/matbanded.c:38-56,104-105 with the reference's defaults kmax=50, frac=0.95 (:261-262)."""
import numpy as np
import pytest
from scipy.linalg import solve_banded


@pytest.mark.parametrize("N,K", [(257, 1), (1000, 3), (4096, 8), (3000, 17), (2048, 32), (1500, 50)])
def test_lu_matches_lapack(oracle, N, K):
    band = oracle.gen_band(N, K, seed=7)
    u = np.ones(N)
    b = oracle.band_matvec(band, u)
    xe = solve_banded((K, K), oracle.to_lapack_ab(band), b)
    sp = oracle.Spike(band, 1)
    x = sp.apply(b, 1)
    assert np.abs(x - xe).max() <= 1e-12 * np.abs(xe).max()
    assert np.abs(x - u).max() <= 1e-11  # manufactured solution, reference src/testbed2.c:120-132


@pytest.mark.parametrize("N,K,P", [(16384, 1, 4), (8192, 8, 8), (8192, 32, 16), (16384, 64, 8), (9000, 20, 5)])
@pytest.mark.parametrize("delta", [1.2, 0.8])
def test_truncated_spike_close_to_exact(oracle, N, K, P, delta):
    band = oracle.gen_band(N, K, delta=delta)
    u = oracle.gen_vec(N)
    b = oracle.band_matvec(band, u)
    xe = solve_banded((K, K), oracle.to_lapack_ab(band), b)
    sp = oracle.Spike(band, P)
    xc = sp.apply(b, 1)
    xd = sp.apply(b, 0)
    # coupled truncated SPIKE reproduces the exact band solve when the spikes decay inside a partition
    assert np.linalg.norm(xc - xe) <= 1e-9 * np.linalg.norm(xe)
    # the decoupled variant is only a block-Jacobi approximation
    assert np.linalg.norm(xd - xe) <= 0.5 * np.linalg.norm(xe)
    assert sp.nboost == 0


def test_partition_rule(oracle):
    s = oracle.partition(16384, 4)
    assert list(s) == [0, 4096, 8192, 12288, 16384]
    s = oracle.partition(1000, 3)  # 16 blocks of 64 -> 5,5,6 blocks; last one is cut at N
    assert list(s) == [0, 320, 640, 1000]
    with pytest.raises(ValueError):
        oracle.partition(100, 3)


def test_pivot_boost(oracle):
    N, K = 512, 2
    band = oracle.gen_band(N, K)
    band[K, 100] = 0.0  # zero pivot candidate
    band[K - 1, 100] = 0.0
    band[K - 2, 100] = 0.0  # row 100 has no lower entries: its pivot stays exactly 0 without boosting
    sp = oracle.Spike(band, 2, boost_rel=1e-8)
    assert sp.nboost >= 1
    x = sp.apply(np.ones(N), 1)
    assert np.all(np.isfinite(x))


def _csr_from_dense(A):
    n = A.shape[0]
    ia = [0]
    ja = []
    a = []
    for r in range(n):
        for c in range(n):
            if A[r, c] != 0.0:
                ja.append(c)
                a.append(A[r, c])
        ia.append(len(ja))
    return n, np.array(ia), np.array(ja), np.array(a)


def test_band_extract_rule_handworked(oracle):
    # weights per offset: w0 = 4*10 = 40, w1 = 6*1 = 6, w2 = 4*0.5 = 2, w3 = 2*0.25 = .5 ; normA = 48.5
    n = 4
    A = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            A[i, j] = {0: 10.0, 1: 1.0, 2: 0.5, 3: 0.25}[abs(i - j)]
    n, ia, ja, a = _csr_from_dense(A)
    # frac .8: 40/48.5 = .8247 >= .8 at k=0
    k, f, ib, jb, b = oracle.band_extract(n, ia, ja, a, kmax=50, frac=0.8)
    assert k == 0 and abs(f - 40 / 48.5) < 1e-15 and len(b) == 4
    # frac .95: 46/48.5 = .948 < .95 ; 48/48.5 = .9897 at k=2
    k, f, ib, jb, b = oracle.band_extract(n, ia, ja, a, kmax=50, frac=0.95)
    assert k == 2 and abs(f - 48 / 48.5) < 1e-15 and len(b) == 14
    # kmax fall-through (matbanded.c:53-56): loop ends with k = kmax = 1, normB holds w0 only,
    # but the copy keeps |c-r| <= 1
    k, f, ib, jb, b = oracle.band_extract(n, ia, ja, a, kmax=1, frac=0.95)
    assert k == 1 and abs(f - 40 / 48.5) < 1e-15 and len(b) == 10
    # defaults of the reference (matbanded.c:261-262)
    k, f, *_ = oracle.band_extract(n, ia, ja, a)
    assert k == 2


def test_band_extract_then_band_layout(oracle):
    rng = np.random.default_rng(3)
    n = 200
    A = np.zeros((n, n))
    for i in range(n):
        for j in range(max(0, i - 6), min(n, i + 7)):
            A[i, j] = rng.uniform(-1, 1) * (0.3 ** abs(i - j))
        A[i, i] = 4.0
    n, ia, ja, a = _csr_from_dense(A)
    k, f, ib, jb, b = oracle.band_extract(n, ia, ja, a, kmax=50, frac=0.95)
    band = oracle.csr_to_band(n, ib, jb, b, k)
    B = np.zeros((n, n))
    for d in range(2 * k + 1):
        for i in range(n):
            c = i + d - k
            if 0 <= c < n:
                B[i, c] = band[d, i]
    mask = np.abs(np.subtract.outer(np.arange(n), np.arange(n))) <= k
    assert np.array_equal(B, A * mask)
    assert f >= 0.95 and k < 6


def test_gmres_manufactured_solution(oracle):
    # reference src/testbed2.c:120-132 (u = 1, b = A u, report ||x-u||) with src/makefile:18 options
    N, K, P = 8192, 16, 8
    band = oracle.gen_band(N, K, delta=0.8)
    u = np.ones(N)
    b = oracle.band_matvec(band, u)
    sp = oracle.Spike(band, P)
    x, it, rn, hist, ok = oracle.gmres(band, b, sp, variant=1, restart=30, rtol=1e-5, maxit=500)
    assert ok and it <= 2 and np.linalg.norm(x - u) <= 1e-6 * np.sqrt(N)
    x0, it0, rn0, hist0, ok0 = oracle.gmres(band, b, sp, variant=0)
    assert ok0 and it0 > it and np.linalg.norm(x0 - u) <= 1e-2 * np.sqrt(N)


def test_distributed_band_rule_pieces_equal_the_sequential_rule(oracle):
    """spike_csr_band_weights (one rank's part of the weights, row-block layout of matbanded.c:36) + spike_band_rule: the
    parts of any row split, added in rank order, reproduce the k of the sequential rule (and its fraction to rounding)."""
    import ctypes as C
    import scipy.sparse as sps
    from conftest import _ensure_built
    _ensure_built()
    import spike_petsc_amd as S
    L = S.lib()
    rng = np.random.default_rng(3)
    n = 3000
    A = sps.random(n, n, density=0.004, random_state=5, format="csr") + sps.diags([rng.uniform(1, 2, n)], [0])
    A = sps.csr_matrix(A)
    A.sort_indices()
    ia, ja, a = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)
    for kmax, frac in ((50, 0.95), (7, 0.999), (300, 0.5)):
        k0, f0 = S.csr_band_k(n, ia, ja, a, kmax, frac)
        ko, fo, *_ = oracle.band_extract(n, ia, ja, a, kmax, frac)
        assert (k0, f0) == (ko, fo)
        for cuts in ([0, n], [0, 1024, n], [0, 64, 1984, 2048, n]):
            w = np.zeros(kmax)
            na = 0.0
            for r0, r1 in zip(cuts[:-1], cuts[1:]):
                wp = np.zeros(kmax)
                nap = C.c_double(0)
                lia = np.ascontiguousarray(ia[r0:r1 + 1] - ia[r0])
                lja = np.ascontiguousarray(ja[ia[r0]:ia[r1]])
                la = np.ascontiguousarray(a[ia[r0]:ia[r1]])
                assert L.spike_csr_band_weights(n, r0, r1 - r0, lia.ctypes.data_as(S.iptr), lja.ctypes.data_as(S.iptr),
                                                la.ctypes.data_as(S.dptr), kmax, wp.ctypes.data_as(S.dptr), C.byref(nap)) == 0
                w += wp
                na += nap.value
            k = C.c_int(0)
            f = C.c_double(0)
            assert L.spike_band_rule(n, w.ctypes.data_as(S.dptr), na, kmax, frac, C.byref(k), C.byref(f)) == 0
            assert k.value == k0 and abs(f.value - f0) <= 1e-14
