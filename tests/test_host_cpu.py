"""CPU tests of the host mirror (libspike_petsc_host.so): MC64 job 5, Fiedler, permutation conventions, options.

MC64 pinning: (1) the known answer SURVEY.md section 4 records for the 3x3 matrix of /root/reference/src/wbm.c:485-497
run through the reference's own HSLmc64AD(job 5) (perm = [3,1,2], num = 3, u = [0,0,ln2], v = [-ln8,-ln2,-ln4]);
(2) on generic values the optimum is unique, so ANY correct maximum-product matching must give the same permutation:
checked against scipy.optimize.linear_sum_assignment.  Tie-heavy inputs: 'parity unpinned' (the reference cannot be
built here without stand-in headers), only optimality and the scaling property are asserted."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linear_sum_assignment

from matrices import circuit_like


@pytest.fixture(scope="module")
def H():
    from conftest import _ensure_built
    _ensure_built()
    import spike_petsc_amd.host as H
    H.lib()
    return H


def test_host_symbols_exported(H):
    L = H.lib()
    assert not [s for s in H.HOST_SYMBOLS if not hasattr(L, s)]
    import os, re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "spike_petsc_host.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decl = set(re.findall(r"\b([A-Za-z_][A-Za-z_0-9]*)\s*\(", src)) - {"defined"}
    decl = {d for d in decl if not d.endswith("Fn") and d[0].isupper() or d.startswith("spike_")}
    decl -= {"PetscErrorCode"}
    missing = [d for d in decl if not hasattr(L, d)]
    assert not missing, missing


@pytest.mark.hostbox
def test_mc64_known_answer_wbm_3x3(H):
    # rows of the matrix at src/wbm.c:485-497: r0={(1,8),(2,3)}, r1={(1,2),(2,1)}, r2={(0,4)}; the wrapper hands the
    # CSR arrays to the CSC interface (src/petsc_mat_wbm.c:29,52)
    ia, ja, a = [0, 2, 4, 5], [1, 2, 1, 2, 0], [8.0, 3.0, 2.0, 1.0, 4.0]
    perm, u, v, num = H.mc64_job5(3, ia, ja, a)
    assert list(perm + 1) == [3, 1, 2] and num == 3
    assert np.array_equal(u, [0.0, 0.0, 0.6931471805599453])
    assert np.array_equal(v, [-2.0794415416798357, -0.6931471805599453, -1.3862943611198906])


def _cost(A):
    Ad = np.abs(A.toarray())
    m = Ad > 0
    cmax = Ad.max(axis=0)
    D = np.full(Ad.shape, 1e30)
    D[m] = (np.log(cmax)[None, :] - np.log(np.where(m, Ad, 1.0)))[m]
    return Ad, D


@pytest.mark.parametrize("n", [8, 64, 512, 2000])
def test_mc64_unique_optimum_matches_scipy(H, n):
    rng = np.random.default_rng(n)
    A = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=rng, format="csc",
                  data_rvs=lambda k: rng.uniform(0.1, 1, k) * rng.choice([-1, 1], k))
    A = (A + sp.diags(rng.uniform(0.01, 0.2, n))).tocsc()
    A.sort_indices()
    perm, u, v, num = H.mc64_job5(n, A.indptr, A.indices, A.data)
    assert num == n and sorted(perm) == list(range(n))
    Ad, D = _cost(A)
    r, c = linear_sum_assignment(D)
    assert np.array_equal(perm, c)                      # unique optimum -> identical permutation
    S = Ad * np.exp(u)[:, None] * np.exp(v)[None, :]     # |a_ij| e^{u_i+v_j} <= 1, = 1 on the matching
    assert S.max() <= 1 + 1e-12 and S[np.arange(n), perm].min() >= 1 - 1e-12


def test_mc64_tie_heavy_is_optimal(H):
    A = circuit_like(600, seed=3).tocsc()
    A.sort_indices()
    n = A.shape[0]
    perm, u, v, num = H.mc64_job5(n, A.indptr, A.indices, A.data)
    assert num == n and sorted(perm) == list(range(n))
    Ad, D = _cost(A)
    r, c = linear_sum_assignment(D)
    assert abs(D[np.arange(n), perm].sum() - D[r, c].sum()) <= 1e-9
    S = Ad * np.exp(u)[:, None] * np.exp(v)[None, :]
    assert S.max() <= 1 + 1e-10 and S[np.arange(n), perm].min() >= 1 - 1e-10
    p2, *_ = H.mc64_job5(n, A.indptr, A.indices, A.data)  # re-entrant and deterministic (no static state)
    assert np.array_equal(perm, p2)


def test_mc64_dense_column_rule_and_singular(H):
    # n > 50 with one column holding more than n/10 entries: the cheap-assignment pass must skip it
    # (src/hslmc64.c:1999-2001) and the main loop must still complete the matching
    n = 80
    rng = np.random.default_rng(7)
    A = sp.lil_matrix((n, n))
    for i in range(n):
        A[i, i] = 1.0 + rng.random()
        A[i, (i * 7 + 3) % n] = 0.3 * rng.random() + 0.1
    A[:, 5] = rng.uniform(0.5, 3.0, (n, 1))
    A = A.tocsc(); A.sort_indices()
    perm, u, v, num = H.mc64_job5(n, A.indptr, A.indices, A.data)
    Ad, D = _cost(A)
    r, c = linear_sum_assignment(D)
    assert num == n and abs(D[np.arange(n), perm].sum() - D[r, c].sum()) <= 1e-9
    # structurally singular: an empty column -> num < n and one negative (completed) entry
    B = sp.lil_matrix((4, 4))
    B[0, 0] = 2; B[1, 1] = 3; B[2, 1] = 1; B[3, 3] = 5
    B = B.tocsc(); B.sort_indices()
    perm, u, v, num = H.mc64_job5(4, B.indptr, B.indices, B.data)
    assert num == 3 and (perm < 0).sum() == 1 and sorted(np.where(perm < 0, -perm - 1, perm)) == [0, 1, 2, 3]


@pytest.mark.hostbox
def test_wbm_ordering_conventions(H):
    L = H.lib()
    H.chk(L.SpikePetscRegisterAll())
    H.options()
    A = circuit_like(300, seed=1)
    M = H.Mat.from_scipy(A)
    import ctypes as C
    r, c = C.c_void_p(), C.c_void_p()
    H.chk(L.MatGetOrdering(M.h, b"wbm", C.byref(r), C.byref(c)))
    row, col = H.is_indices(r), H.is_indices(c)
    perm, *_ = H.mc64_job5(300, A.indptr, A.indices, A.data)
    assert np.array_equal(row, np.arange(300)) and np.array_equal(col, perm)   # petsc_mat_wbm.c:57-58
    # the non-reference option applies it as a row permutation: matched entries land on the diagonal
    H.options(mat_wbm_rows=1)
    r2, c2 = C.c_void_p(), C.c_void_p()
    H.chk(L.MatGetOrdering(M.h, b"wbm", C.byref(r2), C.byref(c2)))
    PM = C.c_void_p()
    H.chk(L.MatPermute(M.h, r2, c2, C.byref(PM)))
    B = H.Mat(handle=PM).to_scipy()
    d = np.abs(B.diagonal())
    assert d.min() > 0 and np.log(d).sum() >= np.log(np.abs(A.diagonal()) + 1e-300).sum()
    Bref = A[H.is_indices(r2)][:, H.is_indices(c2)]
    assert abs(B - Bref).max() == 0                                            # B[i][j] = A[rowp[i]][colp[j]]
    H.options()


@pytest.mark.hostbox
def test_vecpermute_matches_matpermute(H):
    # (PA) VecPermute(x, col) = VecPermute(b, row) when A x = b   (the identity KSPSolve_Reorder relies on)
    L = H.lib()
    import ctypes as C
    rng = np.random.default_rng(2)
    n = 50
    A = sp.random(n, n, density=0.2, random_state=rng, format="csr") + sp.eye(n)
    x = rng.standard_normal(n)
    b = A @ x
    rp, cp = rng.permutation(n), rng.permutation(n)
    ir, ic = C.c_void_p(), C.c_void_p()
    H.chk(L.ISCreateGeneral(n, rp.astype(np.int64).ctypes.data_as(H.i64p), C.byref(ir)))
    H.chk(L.ISCreateGeneral(n, cp.astype(np.int64).ctypes.data_as(H.i64p), C.byref(ic)))
    M = H.Mat.from_scipy(A)
    PM = C.c_void_p()
    H.chk(L.MatPermute(M.h, ir, ic, C.byref(PM)))
    vx, vb, vy = H.Vec(values=x), H.Vec(values=b), H.Vec(n)
    H.chk(L.VecPermute(vx.h, ic, 0)); H.chk(L.VecPermute(vb.h, ir, 0))
    H.chk(L.MatMult(PM, vx.h, vy.h))
    assert np.abs(vy.array - vb.array).max() <= 1e-12
    H.chk(L.VecPermute(vx.h, ic, 1))
    assert np.array_equal(vx.array, x)


@pytest.mark.hostbox
def test_fiedler_recovers_hidden_band_and_is_deterministic(H):
    n, K = 4000, 6
    rng = np.random.default_rng(0)
    B = sp.diags([rng.uniform(0.2, 1, n - abs(d)) for d in range(-K, K + 1)], list(range(-K, K + 1))).tocsr()
    q = rng.permutation(n)
    A = B[q][:, q].tocsr(); A.sort_indices()
    o1, v1 = H.fiedler_order(n, A.indptr, A.indices, A.data)
    o2, v2 = H.fiedler_order(n, A.indptr, A.indices, A.data)
    assert np.array_equal(o1, o2) and sorted(o1) == list(range(n))
    p0, b0 = H.profile_bandwidth(n, A.indptr, A.indices)
    p1, b1 = H.profile_bandwidth(n, A.indptr, A.indices, o1)
    assert b1 <= 4 * K and b0 > 100 * K and p1 < p0 / 50
    # the vector approximates the Fiedler vector: orthogonal to the constants, Rayleigh quotient within a few per cent of
    # lambda_2 (the spec fixes the number of refinement steps; the ORDER is what must be good, see the bandwidth above)
    W = abs(A - sp.diags(A.diagonal())); W = W + W.T
    Lp = (sp.diags(np.asarray(W.sum(axis=1)).ravel()) - W).tocsc()
    rho = v1 @ (Lp @ v1) / (v1 @ v1)
    import scipy.sparse.linalg as spl
    lam = np.sort(spl.eigsh(Lp, k=2, sigma=-1e-3, which="LM", return_eigenvectors=False))[1]
    assert abs(v1.sum()) <= 1e-8 and lam * (1 - 1e-9) <= rho <= 1.25 * lam


@pytest.mark.hostbox
def test_fiedler_halves_is_the_reference_prototype(H):
    """Per-half reordering (src/spectralPartition.c:326-417): Fiedler cut where the vector changes sign, RCM on each half's
    diagonal block, the two permutations composed.  Checked against a direct restatement of those steps."""
    n, K = 3000, 5
    rng = np.random.default_rng(2)
    B = sp.diags([rng.uniform(0.2, 1, n - abs(d)) for d in range(-K, K + 1)], list(range(-K, K + 1))).tocsr()
    q = rng.permutation(n)
    A = B[q][:, q].tocsr(); A.sort_indices()
    order, npos, bw = H.fiedler_halves_order(n, A.indptr, A.indices, A.data)
    assert sorted(order.tolist()) == list(range(n))
    of, vec = H.fiedler_order(n, A.indptr, A.indices, A.data)
    assert npos == int((vec > 0).sum()) and 0 < npos < n
    seq = np.concatenate([[v for v in of if vec[v] > 0], [v for v in of if not vec[v] > 0]]).astype(np.int64)   # :340-343
    assert set(order[:npos].tolist()) == set(seq[:npos].tolist())                                     # halves keep their members
    for off, m, k in ((0, npos, 0), (npos, n - npos, 2)):
        S = A[seq[off:off + m]][:, seq[off:off + m]].tocsr(); S.sort_indices()                         # MatGetSubMatrix, :373-374
        so = H.rcm_order(m, S.indptr, S.indices)                                                       # MatGetOrdering, :377-378
        assert np.array_equal(order[off:off + m], seq[off + so])                                       # composition, :388-404
        assert bw[k] == H.profile_bandwidth(m, S.indptr, S.indices)[1] and bw[k + 1] == H.profile_bandwidth(m, S.indptr, S.indices, so)[1]
        assert bw[k + 1] <= bw[k]
    # (the scheme reorders the halves independently: it narrows each diagonal block, not the coupling between the halves)
    assert bw[1] <= 4 * K and bw[3] <= 4 * K


@pytest.mark.hostbox
def test_fiedler_sorted_and_unsorted_rows_and_the_tie_rule(H):
    """Rows with ascending columns take the transpose-and-merge graph build, anything else the general counting-sort build:
    same graph, same bits.  Large components are ordered by a stable radix sort: descending value, ties by index."""
    n, K = 6000, 4
    rng = np.random.default_rng(5)
    B = sp.diags([rng.uniform(0.2, 1, n - abs(d)) for d in range(-K, K + 1)], list(range(-K, K + 1))).tocsr()
    q = rng.permutation(n)
    A = B[q][:, q].tocoo()
    keep = (A.row <= A.col) | (rng.random(A.nnz) < 0.5)          # unsymmetric pattern: half the edges stored on one side only
    A = sp.csr_matrix((A.data[keep], (A.row[keep], A.col[keep])), shape=(n, n)); A.sort_indices()
    ia, ja, a = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.copy()
    o1, v1 = H.fiedler_order(n, ia, ja, a)
    ja2, a2 = ja.copy(), a.copy()
    for i in range(n):                                           # the same rows, entries in descending column order
        ja2[ia[i]:ia[i + 1]] = ja[ia[i]:ia[i + 1]][::-1]; a2[ia[i]:ia[i + 1]] = a[ia[i]:ia[i + 1]][::-1]
    assert a[0] > 0 and a2[0] > 0                                # weighted mode (decided by the sign of a[0]) on both
    o2, v2 = H.fiedler_order(n, ia, ja2, a2)
    assert np.array_equal(o1, o2) and np.array_equal(v1, v2)
    assert np.array_equal(o1, np.argsort(-(v1 + 0.0), kind="stable"))
    # ties: two identical, disconnected halves of a graph give the same vector twice per component; inside a component of
    # >= 4096 vertices with mirror symmetry pairs of equal entries exist -> the smaller index comes first
    m = 5001
    P = sp.diags([np.ones(m - 1), np.ones(m - 1)], [-1, 1]).tocsr()   # a path: v_i = -v_{m-1-i}; the middle entry is 0
    o, v = H.fiedler_order(m, P.indptr, P.indices, P.data)
    assert np.array_equal(o, np.argsort(-(v + 0.0), kind="stable")) and sorted(o.tolist()) == list(range(m))


@pytest.mark.hostbox
def test_fiedler_small_exact_and_components(H):
    # path graph on 9 vertices: Fiedler vector is monotone -> the order is the path (up to direction)
    n = 9
    A = sp.diags([np.ones(n - 1), 2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1]).tocsr()
    o, v = H.fiedler_order(n, A.indptr, A.indices, A.data)
    assert list(o) in (list(range(n)), list(range(n))[::-1])
    ev = np.linalg.eigvalsh((sp.diags(np.asarray((A - sp.diags(A.diagonal())).sum(axis=1)).ravel()) - (A - sp.diags(A.diagonal()))).toarray() * 2)
    # two components + an isolated vertex: components in order of smallest vertex
    C2 = sp.block_diag([A[:5, :5], A[:3, :3], sp.eye(1)]).tocsr()
    o, v = H.fiedler_order(9, C2.indptr, C2.indices, C2.data)
    assert sorted(o[:5]) == [0, 1, 2, 3, 4] and sorted(o[5:8]) == [5, 6, 7] and o[8] == 8


@pytest.mark.hostbox
def test_matcreatesubmatrixbanded_is_reference_rule(H, ):
    import ctypes as C
    import oracle as O
    L = H.lib()
    A = circuit_like(500, seed=5, scramble=False, unsym_rows=False)
    M = H.Mat.from_scipy(A)
    for kmax, frac in [(50, 0.95), (3, 0.99), (50, 0.6)]:
        k, f, B = C.c_int64(kmax), C.c_double(frac), C.c_void_p()
        H.chk(L.MatCreateSubMatrixBanded(M.h, C.byref(k), C.byref(f), C.byref(B)))
        ko, fo, ib, jb, bb = O.band_extract(500, A.indptr, A.indices, A.data, kmax, frac)
        n, ia, ja, a = H.Mat(handle=B).csr()
        assert k.value == ko and f.value == fo and np.array_equal(ia, ib) and np.array_equal(ja, jb) and np.array_equal(a, bb)


def _awbm_py(n, ia, ja, a):
    """independent restatement of /root/reference/src/petsc_mat_awbm.c:42-205 in plain Python (small n only)"""
    import math
    eps = math.sqrt(np.finfo(float).eps)
    amax = [max(abs(a[r]) for r in range(ia[c], ia[c + 1])) for c in range(n)]
    w = [0.0] * len(a)
    for c in range(n):
        for r in range(ia[c], ia[c + 1]):
            w[r] = float("inf") if a[r] == 0 else math.log(amax[c] / abs(a[r]))
    u = [float("inf")] * n
    for c in range(n):
        for r in range(ia[c], ia[c + 1]):
            u[ja[r]] = min(u[ja[r]], w[r])
    v = [min(w[r] - u[ja[r]] for r in range(ia[c], ia[c + 1])) for c in range(n)]
    match, matchR = [-1] * n, [-1] * n
    for c in range(n):
        for r in range(ia[c], ia[c + 1]):
            if w[r] - u[ja[r]] - v[c] <= eps and matchR[ja[r]] < 0:
                match[c], matchR[ja[r]] = ja[r], c
                break
    for tight in (True, False):
        if not tight:
            for c in range(n):
                if match[c] >= 0:
                    continue
                for r in range(ia[c], ia[c + 1]):
                    if matchR[ja[r]] < 0:
                        match[c], matchR[ja[r]] = ja[r], c
                        break
        for c in range(n):
            if match[c] >= 0:
                continue
            for r in range(ia[c], ia[c + 1]):
                if tight and w[r] - u[ja[r]] - v[c] > eps:
                    continue
                c1 = matchR[ja[r]]
                for r1 in range(ia[c1], ia[c1 + 1]):
                    if matchR[ja[r1]] < 0 and (not tight or w[r1] - u[ja[r1]] - v[c1] <= eps):
                        match[c], matchR[ja[r]] = ja[r], c
                        match[c1], matchR[ja[r1]] = ja[r1], c1
                        break
                if match[c] >= 0:
                    break
    r = 0
    for c in range(n):
        if match[c] < 0:
            while r < n:
                if matchR[r] < 0:
                    match[c], matchR[r] = r, c
                    break
                r += 1
    p = [0] * n
    for c in range(n):
        p[match[c]] = c
    return np.array(p)


@pytest.mark.hostbox
def test_awbm_matches_independent_restatement(H):
    for seed, n in [(0, 40), (1, 150), (2, 400)]:
        A = circuit_like(n, seed=seed).tocsr()
        A.sort_indices()
        p = H.awbm(n, A.indptr, A.indices, A.data)
        assert sorted(p) == list(range(n))
        assert np.array_equal(p, _awbm_py(n, list(A.indptr), list(A.indices), list(A.data)))
        # B[i][j] = A[p[i]][j] has a zero-free diagonal (the matching is perfect on these matrices)
        assert np.abs(A[p].diagonal()).min() > 0
    import ctypes as C
    L = H.lib()
    H.chk(L.SpikePetscRegisterAll())
    M = H.Mat.from_scipy(A)
    r, c = C.c_void_p(), C.c_void_p()
    H.chk(L.MatGetOrdering(M.h, b"awbm", C.byref(r), C.byref(c)))          # registered as in testbed2.c:67
    assert np.array_equal(H.is_indices(r), p) and np.array_equal(H.is_indices(c), np.arange(n))


@pytest.mark.hostbox
def test_file_formats_roundtrip(H, tmp_path):
    import ctypes as C
    import struct
    import scipy.io
    L = H.lib()
    A = circuit_like(200, seed=4).tocsr()
    A.sort_indices()
    # MatrixMarket written by scipy, read by the host library; and back
    mm = str(tmp_path / "a.mtx")
    scipy.io.mmwrite(mm, A, precision=17)
    h = C.c_void_p()
    H.chk(L.MatLoadMatrixMarket(mm.encode(), C.byref(h)))
    assert abs(H.Mat(handle=h).to_scipy() - A).max() == 0
    mm2 = str(tmp_path / "b.mtx")
    H.chk(L.MatViewMatrixMarket(h, mm2.encode()))
    assert abs(scipy.io.mmread(mm2).tocsr() - A).max() == 0
    S = (A + A.T).tocsr()
    scipy.io.mmwrite(str(tmp_path / "s.mtx"), S, symmetry="symmetric", precision=17)
    hs = C.c_void_p()
    H.chk(L.MatLoadMatrixMarket(str(tmp_path / "s.mtx").encode(), C.byref(hs)))
    assert abs(H.Mat(handle=hs).to_scipy() - S).max() <= 1e-15
    # PETSc binary AIJ written here from the format's definition (big-endian), read by MatLoad; and back
    pb = str(tmp_path / "a.bin")
    with open(pb, "wb") as f:
        f.write(struct.pack(">4i", 1211216, 200, 200, A.nnz))
        f.write(struct.pack(">%di" % 200, *np.diff(A.indptr)))
        f.write(struct.pack(">%di" % A.nnz, *A.indices))
        f.write(struct.pack(">%dd" % A.nnz, *A.data))
    hb = C.c_void_p()
    H.chk(L.MatLoad(pb.encode(), C.byref(hb)))
    assert abs(H.Mat(handle=hb).to_scipy() - A).max() == 0
    pb2 = str(tmp_path / "b.bin")
    H.chk(L.MatViewBinary(hb, pb2.encode()))
    assert open(pb, "rb").read() == open(pb2, "rb").read()
    assert L.MatLoad(mm.encode(), C.byref(C.c_void_p())) != 0     # wrong format is an error, not a crash


@pytest.mark.hostbox
def test_rcm_reduces_bandwidth_like_scipy(H):
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    n, K = 3000, 5
    rng = np.random.default_rng(3)
    B = sp.diags([rng.uniform(0.2, 1, n - abs(d)) for d in range(-K, K + 1)], list(range(-K, K + 1))).tocsr()
    q = rng.permutation(n)
    A = B[q][:, q].tocsr(); A.sort_indices()
    o = H.rcm_order(n, A.indptr, A.indices)
    assert sorted(o) == list(range(n)) and np.array_equal(o, H.rcm_order(n, A.indptr, A.indices))
    p0, b0 = H.profile_bandwidth(n, A.indptr, A.indices)
    p1, b1 = H.profile_bandwidth(n, A.indptr, A.indices, o)
    ps, bs = H.profile_bandwidth(n, A.indptr, A.indices, reverse_cuthill_mckee(A, symmetric_mode=False).astype(np.int64))
    assert b1 <= 3 * K and b1 <= bs + K and b0 > 100 * K          # as good as scipy's RCM on a hidden band
    # disconnected pattern: every vertex appears once, components stay contiguous
    C2 = sp.block_diag([B[:50, :50], B[:30, :30], sp.eye(1)]).tocsr()
    o = H.rcm_order(81, C2.indptr, C2.indices)
    assert sorted(o) == list(range(81))


@pytest.mark.hostbox
def test_32bit_index_entry_points_equal_the_64bit_ones(H):
    """PETSc's default build has a 32-bit PetscInt: the _i32 entry points (csrc/host/idx32.c) must give the 64-bit kernels'
    results on the same matrix (they widen, call, narrow).  Guard words around the 4-byte output arrays catch an 8-byte
    write (what casting a PetscInt* to int64_t* did in round 2's glue)."""
    import ctypes as C
    L = H.lib()
    i32p = C.POINTER(C.c_int32)
    dpp = C.POINTER(C.c_double)
    n = 3000
    A = circuit_like(n, seed=4)
    ia32, ja32 = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    a = np.ascontiguousarray(A.data, dtype=np.float64)

    def guarded():
        buf = np.full(n + 2, -77, dtype=np.int32)
        return buf, buf[1:n + 1]

    # MC64 job 5
    p64, u64, v64, num64 = H.mc64_job5(n, A.indptr, A.indices, A.data)
    buf, perm = guarded()
    u, v, num = np.zeros(n), np.zeros(n), C.c_int32(0)
    assert L.spike_mc64_job5_i32(n, ia32.ctypes.data_as(i32p), ja32.ctypes.data_as(i32p), a.ctypes.data_as(dpp),
                                 perm.ctypes.data_as(i32p), u.ctypes.data_as(dpp), v.ctypes.data_as(dpp), C.byref(num)) == 0
    assert buf[0] == -77 and buf[-1] == -77
    assert np.array_equal(perm.astype(np.int64), p64) and num.value == num64 and np.array_equal(u, u64) and np.array_equal(v, v64)
    # AWBM
    buf, perm = guarded()
    assert L.spike_awbm_i32(n, ia32.ctypes.data_as(i32p), ja32.ctypes.data_as(i32p), a.ctypes.data_as(dpp),
                            perm.ctypes.data_as(i32p), None, None) == 0
    assert buf[0] == -77 and buf[-1] == -77 and np.array_equal(perm.astype(np.int64), H.awbm(n, A.indptr, A.indices, A.data))
    # Fiedler, RCM on the symmetrised pattern
    S = sp.csr_matrix(A + A.T)
    S.sort_indices()
    sa32, sj32 = S.indptr.astype(np.int32), S.indices.astype(np.int32)
    sd = np.ascontiguousarray(S.data, dtype=np.float64)
    buf, order = guarded()
    vec = np.zeros(n)
    assert L.spike_fiedler_order_i32(n, sa32.ctypes.data_as(i32p), sj32.ctypes.data_as(i32p), sd.ctypes.data_as(dpp),
                                     order.ctypes.data_as(i32p), vec.ctypes.data_as(dpp), 0) == 0
    o64, v64 = H.fiedler_order(n, S.indptr, S.indices, S.data)
    assert buf[0] == -77 and buf[-1] == -77 and np.array_equal(order.astype(np.int64), o64) and np.array_equal(vec, v64)
    buf, order = guarded()
    assert L.spike_rcm_order_i32(n, sa32.ctypes.data_as(i32p), sj32.ctypes.data_as(i32p), order.ctypes.data_as(i32p)) == 0
    assert buf[0] == -77 and buf[-1] == -77 and np.array_equal(order.astype(np.int64), H.rcm_order(n, S.indptr, S.indices))
    # bad input is refused, not read
    assert L.spike_rcm_order_i32(0, sa32.ctypes.data_as(i32p), sj32.ctypes.data_as(i32p), order.ctypes.data_as(i32p)) != 0
