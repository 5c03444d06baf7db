"""The approximate matching for a row-distributed matrix (csrc/host/awbm_dist.c; reference: MatComputeMatching_MPIAIJ,
/root/reference/src/wbm.c:201-440, and its one-rank form MatComputeMatching_SeqAIJ, :44-183).  The reference holds no
expected output for it (parity unpinned): one rank is compared with an independent restatement of :44-183 in plain Python,
several gloo ranks with a restatement of the distributed steps, with numpy for the reduced weights, and -- block-diagonal
input -- with the one-rank result of every block."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

EPS = math.sqrt(np.finfo(float).eps)
BIG = np.finfo(float).max


def _weights(ia, ja, a, n):
    w = [0.0] * len(ja)
    for c in range(n):
        amax = max([abs(a[r]) for r in range(ia[c], ia[c + 1])] or [0.0])
        for r in range(ia[c], ia[c + 1]):
            w[r] = BIG if a[r] == 0.0 else math.log(amax / abs(a[r]))
    return w


def _match_py(n, row0, N, ia, ja, a, u):
    """phases of wbm.c (:291-318 tight greedy, :320-395 tight one-step augmentation, :398-410 fill) on the rows
    [row0, row0 + n) with partners inside the diagonal block; one rank (row0 = 0, n = N): wbm.c:90-150"""
    w = _weights(ia, ja, a, n)
    v = [min([w[r] - u[ja[r]] for r in range(ia[c], ia[c + 1])] or [BIG]) for c in range(n)]
    match, matchR = [-1] * n, [-1] * n
    loc = lambda r: row0 <= ja[r] < row0 + n
    for c in range(n):
        for r in range(ia[c], ia[c + 1]):
            if loc(r) and w[r] - u[ja[r]] - v[c] <= EPS and matchR[ja[r] - row0] < 0:
                match[c] = ja[r] - row0; matchR[ja[r] - row0] = c
                break
    for c in range(n):
        if match[c] >= 0:
            continue
        for r in range(ia[c], ia[c + 1]):
            if not loc(r) or w[r] - u[ja[r]] - v[c] > EPS:
                continue
            l = ja[r] - row0
            c1 = matchR[l]
            if c1 < 0:
                continue
            for r1 in range(ia[c1], ia[c1 + 1]):
                if loc(r1) and matchR[ja[r1] - row0] < 0 and w[r1] - u[ja[r1]] - v[c1] <= EPS:
                    match[c] = l; matchR[l] = c
                    match[c1] = ja[r1] - row0; matchR[ja[r1] - row0] = c1
                    break
            if match[c] >= 0:
                break
    r = 0
    for c in range(n):
        if match[c] >= 0:
            continue
        while r < n:
            if matchR[r] < 0:
                match[c] = r; matchR[r] = c
                break
            r += 1
    p = [0] * n
    for c in range(n):
        p[match[c]] = c
    return np.array(p)


def _u_numpy(A):
    """per column: the minimum over ALL entries of log(rowmax / |a|)"""
    A = A.tocsr()
    amax = np.abs(A).max(axis=1).toarray().ravel()
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    # libm's log, as the C code uses (numpy's vectorised log may differ in the last bit)
    w = np.array([BIG if x == 0.0 else math.log(m / abs(x)) for x, m in zip(A.data, amax[rows])])
    u = np.full(A.shape[1], BIG)
    np.minimum.at(u, A.indices, w)
    return u


def _matrix(n, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "block":       # two (or three) diagonal blocks cut where the ranks cut
        parts = [sp.random(n // 2, n // 2, density=0.03, random_state=seed + 1, format="csr"),
                 sp.random(n - n // 2, n - n // 2, density=0.03, random_state=seed + 2, format="csr")]
        A = sp.block_diag(parts, format="csr")
    else:
        A = sp.random(n, n, density=0.02, random_state=seed, format="csr")
    A = A + sp.diags(rng.uniform(0.1, 2.0, n))          # a nonzero diagonal of mixed weight
    if kind == "ties":
        A.data[:] = rng.choice([1.0, 0.5, 0.25], size=A.nnz)
    A = A.tocsr(); A.sort_indices()
    return A


@pytest.mark.parametrize("kind", ["general", "ties", "block"])
def test_one_rank_equals_the_restatement_of_the_sequential_routine(kind):
    import spike_petsc_amd.host as H
    for n, seed in [(60, 1), (257, 2), (1000, 3)]:
        A = _matrix(n, seed, kind)
        p, u = H.awbm_dist(0, n, A.indptr, A.indices, A.data)
        assert np.array_equal(u, _u_numpy(A))
        assert sorted(p) == list(range(n))
        assert np.array_equal(p, _match_py(n, 0, n, list(A.indptr), list(A.indices), list(A.data), list(u)))
    # scalings as in wbm.c:429-432
    A = _matrix(120, 9, "general")
    p, u, sr, sc = H.awbm_dist(0, 120, A.indptr, A.indices, A.data, scalings=True)
    assert np.allclose(sc, np.exp(u)) and np.all(sr > 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, kind, q):
    import torch.distributed as dist
    import spike_petsc_amd.host as H
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A = _matrix(n, seed, kind)
    cuts = [(n * r) // world for r in range(world + 1)]
    if kind == "block" and world == 2:
        cuts = [0, n // 2, n]
    r0, r1 = cuts[rank], cuts[rank + 1]
    loc = A[r0:r1].tocsr()                     # this rank's rows, global column indices
    p, u = H.awbm_dist(r0, n, loc.indptr, loc.indices, loc.data)
    q.put((rank, r0, r1, p, u))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "general"), (2, "ties"), (2, "block"), (3, "general")])
def test_gloo_ranks(world, kind):
    import torch.multiprocessing as mp
    import spike_petsc_amd.host as H
    n, seed = 301, 17
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, seed, kind, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    A = _matrix(n, seed, kind)
    ug = _u_numpy(A)
    for rank, r0, r1, p, u in res:
        assert np.array_equal(u, ug), "the reduced weights are the minimum over all ranks' entries"
        assert sorted(p) == list(range(r1 - r0))
        loc = A[r0:r1].tocsr()
        assert np.array_equal(p, _match_py(r1 - r0, r0, n, list(loc.indptr), list(loc.indices), list(loc.data), list(ug)))
        if kind == "block":      # nothing couples the blocks: every rank's result is the one-rank result of its block
            blk = A[r0:r1, r0:r1].tocsr()
            p1, _ = H.awbm_dist(0, r1 - r0, blk.indptr, blk.indices, blk.data)
            assert np.array_equal(p, p1)
