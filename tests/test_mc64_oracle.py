"""MC64 job 5: the product's C matching (spike_mc64_job5) against the separately written Python oracle
(oracle/mc64_oracle.py), BIT FOR BIT, on inputs whose optimum is not unique (ties) -- the cases where the answer
depends on the reference's traversal order (/root/reference/src/hslmc64.c:1917-2380) -- and both against the one known
answer the reference's files hold (tests/golden/mc64_wbm_3x3.json) and against the optimal objective (scipy)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linear_sum_assignment

from matrices import circuit_like
from oracle import mc64_oracle as MO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

pytestmark = pytest.mark.hostbox   # runs in BOTH suites (tests/conftest.py)


@pytest.fixture(scope="module")
def H():
    from conftest import _ensure_built
    _ensure_built()
    import spike_petsc_amd.host as H
    H.lib()
    return H


def _both(H, A):
    A = sp.csc_matrix(A)
    A.sort_indices()
    n = A.shape[0]
    perm, u, v, num = H.mc64_job5(n, A.indptr, A.indices, A.data)
    po, uo, vo, no = MO.mc64_job5(n, A.indptr, A.indices, A.data)
    return n, A, (perm, u, v, num), (np.array(po), np.array(uo), np.array(vo), no)


def _assert_identical(prod, orc):
    assert prod[3] == orc[3]
    assert np.array_equal(prod[0], orc[0])
    assert np.array_equal(prod[1], orc[1])          # scalings bit for bit (same operations in the same order)
    assert np.array_equal(prod[2], orc[2])


def test_oracle_reproduces_the_reference_known_answer():
    d = json.load(open(os.path.join(G, "mc64_wbm_3x3.json")))
    perm, u, v, num = MO.mc64_job5(d["n"], d["ia"], d["ja"], d["a"])
    assert [p + 1 for p in perm] == d["perm_1based"] and num == d["num"]
    assert u == d["u"] and v == d["v"]
    rows, cols = MO.wbm_ordering(d["n"], d["ia"], d["ja"], d["a"])   # petsc_mat_wbm.c:57-58
    assert rows == [0, 1, 2] and [c + 1 for c in cols] == d["perm_1based"]


@pytest.mark.parametrize("n,seed", [(5, 0), (12, 1), (40, 2), (64, 3), (130, 4), (400, 5)])
def test_ties_everywhere_product_equals_oracle(H, n, seed):
    # values from a 4-element set: massive ties in every cost comparison
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=min(1.0, 5.0 / n), random_state=rng, format="lil",
                  data_rvs=lambda k: rng.choice([1.0, -1.0, 0.5, 2.0], k))
    for i in range(n):
        A[i, (i * 3 + 1) % n] = rng.choice([1.0, 0.5])     # a transversal exists
    n, A, prod, orc = _both(H, A)
    _assert_identical(prod, orc)
    assert prod[3] == n and sorted(prod[0]) == list(range(n))
    Ad = np.abs(A.toarray())
    D = np.where(Ad > 0, np.log(Ad.max(axis=0))[None, :] - np.log(np.where(Ad > 0, Ad, 1.0)), 1e30)
    r, c = linear_sum_assignment(D)
    assert abs(D[np.arange(n), prod[0]].sum() - D[r, c].sum()) <= 1e-9


@pytest.mark.parametrize("seed", [3, 11])
def test_circuit_like_product_equals_oracle(H, seed):
    A = circuit_like(700, seed=seed)
    _, _, prod, orc = _both(H, A)
    _assert_identical(prod, orc)


def test_all_equal_dense_block_product_equals_oracle(H):
    # every entry 1: any permutation is optimal, the result is pure traversal order
    for n in (3, 8, 60):
        _, _, prod, orc = _both(H, np.ones((n, n)))
        _assert_identical(prod, orc)
        assert prod[3] == n


def test_dense_column_rule_product_equals_oracle(H):
    n = 80                                              # n > 50 and a column with more than n/10 entries
    rng = np.random.default_rng(7)
    A = sp.lil_matrix((n, n))
    for i in range(n):
        A[i, i] = rng.choice([1.0, 2.0])
        A[i, (i * 7 + 3) % n] = rng.choice([0.5, 1.0])
    A[:, 5] = rng.choice([1.0, 2.0, 4.0], (n, 1))
    A[17, :] = rng.choice([1.0, 2.0], (1, n))
    _, _, prod, orc = _both(H, A)
    _assert_identical(prod, orc)
    assert prod[3] == n


def test_structurally_singular_and_explicit_zeros_product_equals_oracle(H):
    B = sp.lil_matrix((6, 6))
    B[0, 0] = 2; B[1, 1] = 3; B[2, 1] = 1; B[3, 3] = 5; B[4, 3] = 5; B[5, 5] = 1    # columns 2 and 4 empty
    _, _, prod, orc = _both(H, B)
    _assert_identical(prod, orc)
    assert prod[3] == 4 and (prod[0] < 0).sum() == 2
    # explicitly stored zeros take the "infinite" cost RINF/n (hslmc64.c:731-735)
    A = sp.csc_matrix(np.array([[0.0, 2.0, 1.0], [3.0, 0.0, 1.0], [1.0, 1.0, 0.0]]))
    Z = sp.csc_matrix((np.array([0.0, 3.0, 1.0, 2.0, 0.0, 1.0, 1.0, 1.0, 0.0]),
                       np.array([0, 1, 2, 0, 1, 2, 0, 1, 2]), np.array([0, 3, 6, 9])), shape=(3, 3))
    n = 3
    perm, u, v, num = H.mc64_job5(n, Z.indptr, Z.indices, Z.data)
    po, uo, vo, no = MO.mc64_job5(n, Z.indptr, Z.indices, Z.data)
    assert np.array_equal(perm, po) and num == no and np.array_equal(u, uo) and np.array_equal(v, vo)
    assert all(A[i, perm[i]] != 0 for i in range(3))


def test_hypothesis_small_tie_matrices(H):
    # exhaustive-ish sweep of tiny matrices with tied values (the regime where order decides)
    rng = np.random.default_rng(2024)
    for _ in range(300):
        n = int(rng.integers(2, 9))
        M = rng.choice([0.0, 0.0, 1.0, 1.0, 0.5, 2.0], (n, n))
        if not M.any():
            continue
        A = sp.csc_matrix(M)
        A.eliminate_zeros()
        if A.nnz == 0:
            continue
        _, _, prod, orc = _both(H, A)
        _assert_identical(prod, orc)
