"""Product (csrc/host/fiedler.c: spike_fiedler_order) against oracle/fiedler_oracle.py, BIT FOR BIT: the Fiedler vector's
64-bit patterns and the permutation.  The oracle is a second, separately written statement (numpy, whole-vector) of the
deterministic ordering spec this repository publishes for the slot MatGetOrdering_Fiedler
(/root/reference/src/petsc_mat_fiedler.c:11-58).  Against the reference's own HSL_MC73 the path stays PARITY UNPINNED
(library absent, no expected output anywhere in the reference); what is pinned here is product == published spec.

hostbox: runs under -m "not gpu" here and is added to -m gpu on the GPU box (tests/conftest.py)."""
import numpy as np
import pytest
import scipy.sparse as sp

from matrices import circuit_like

pytestmark = pytest.mark.hostbox


@pytest.fixture(scope="module")
def H():
    from conftest import _ensure_built
    _ensure_built()
    import spike_petsc_amd.host as H
    return H


def _same(H, n, ia, ja, a):
    from oracle import fiedler_oracle as FO
    po, pv = H.fiedler_order(n, ia, ja, a, use_device=False)
    oo, ov = FO.fiedler_order(n, ia, ja, a)
    assert sorted(po.tolist()) == list(range(n))
    assert np.array_equal(pv.view(np.uint64), ov.view(np.uint64)), np.abs(pv - ov).max()
    assert np.array_equal(po, oo)
    return po, pv


def test_reference_dead_8x8_pattern(H):
    """the only MC73 input the reference holds (dead code, petsc_mat_fiedler.c:34-36; it records NO expected output):
    1-based, rows unsorted, 17 entries, all values 1.0 (=> weighted mode, hslmc73f.F90:19)"""
    exia = np.array([1, 6, 8, 10, 12, 14, 15, 17, 18]) - 1
    exja = np.array([3, 1, 5, 6, 7, 2, 8, 3, 7, 4, 5, 6, 5, 8, 7, 8, 8]) - 1
    exa = np.ones(17)
    order, vec = _same(H, 8, exia, exja, exa)
    assert abs(vec.sum()) < 1e-12          # every component's vector is orthogonal to the constant


def test_path_graph_is_ordered_end_to_end(H):
    n = 300
    A = sp.diags([np.ones(n - 1), 2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1]).tocsr()
    order, _ = _same(H, n, A.indptr, A.indices, A.data)
    assert order.tolist() in (list(range(n)), list(range(n - 1, -1, -1)))


@pytest.mark.parametrize("n,seed", [(70, 0), (300, 1), (1500, 2)])
def test_random_symmetric_weighted(H, n, seed):
    rng = np.random.default_rng(seed)
    R = sp.random(n, n, density=4.0 / n, random_state=seed, data_rvs=lambda k: rng.uniform(0.1, 2.0, k))
    A = sp.csr_matrix(R + R.T + sp.eye(n))
    A.sort_indices()
    _same(H, n, A.indptr, A.indices, A.data)


def test_unsorted_rows_duplicates_and_one_sided_entries(H):
    """the general graph-build path: rows in arbitrary column order, repeated (i, j) pairs, entries stored on one side
    only, entries below the 1e-12 drop tolerance"""
    rng = np.random.default_rng(5)
    n = 200
    ia, ja, a = [0], [], []
    for i in range(n):
        cols = rng.integers(0, n, size=rng.integers(2, 7)).tolist() + [i]
        if i + 1 < n:
            cols.append(i + 1)           # keeps the graph connected
        if i % 7 == 0:
            cols.append(cols[0])         # a duplicate
        rng.shuffle(cols)
        for j in cols:
            ja.append(int(j))
            a.append(float(rng.choice([1.0, -0.5, 0.25, 1e-13, 2.0])))
        ia.append(len(ja))
    a[0] = 1.0                            # weighted mode
    _same(H, n, np.array(ia), np.array(ja), np.array(a))


def test_unweighted_mode_components_and_tiny_components(H):
    """a[0] <= 0 => every edge weighs 1; several components, among them single vertices and a pair"""
    blocks = []
    for m, seed in ((90, 1), (1, 0), (130, 2), (2, 0), (70, 3)):
        rng = np.random.default_rng(seed)
        B = sp.random(m, m, density=min(1.0, 5.0 / m), random_state=seed, data_rvs=lambda k: rng.uniform(0.5, 1.5, k))
        P = sp.diags([np.ones(m - 1)], [1]) if m > 1 else sp.csr_matrix((1, 1))
        blocks.append(sp.csr_matrix(B + B.T + P + P.T + sp.eye(m)))
    A = sp.block_diag(blocks).tocsr()
    n = A.shape[0]
    q = np.random.default_rng(9).permutation(n)
    A = A[q][:, q].tocsr()
    A.sort_indices()
    A.data[0] = -abs(A.data[0])
    _same(H, n, A.indptr, A.indices, A.data)


def test_star_graph_stalls_the_matching(H):
    """a star: heavy-edge matching pairs the hub with one leaf and stalls (coarse > 9/10 fine) => 1000-iteration level"""
    n = 400
    rows = [0] * (n - 1) + list(range(1, n))
    cols = list(range(1, n)) + [0] * (n - 1)
    A = sp.csr_matrix((np.ones(2 * (n - 1)), (rows, cols)), shape=(n, n)) + sp.eye(n)
    A = sp.csr_matrix(A)
    A.sort_indices()
    _same(H, n, A.indptr, A.indices, A.data)


@pytest.mark.parametrize("n,seed", [(2000, 3), (5000, 11)])
def test_circuit_like_standin(H, n, seed):
    """the config-4 stand-in family (tests/matrices.py) after symmetrisation, at sizes the numpy oracle finishes in seconds"""
    A = circuit_like(n, seed=seed, unsym_rows=False)
    S = sp.csr_matrix(A + A.T)
    S.sort_indices()
    order, _ = _same(H, n, S.indptr, S.indices, S.data)
    # and it does what the ordering is for: the hidden band comes back
    from spike_petsc_amd.host import profile_bandwidth
    assert profile_bandwidth(n, S.indptr, S.indices, order)[1] < profile_bandwidth(n, S.indptr, S.indices)[1] // 10
