"""N>1 algorithm on ONE GPU: ranks are host threads joined by the library's loopback transport
(spike_comm_init_local) -- same buffers, same call order, same interface/halo logic as the RCCL path.
The sharded result must equal the single-handle result (and the oracle) for the same global partitioning."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_group = [100]


def _run_ranks(spike, G, fn):
    """fn(rank, sp) in G threads; returns list of results; re-raises the first exception."""
    _group[0] += 1
    grp = _group[0]
    out, err = [None] * G, [None] * G

    def work(r):
        try:
            sp = spike.Spike(partitions=0)
            sp.comm_init_local(G, r, grp)
            out[r] = fn(r, sp)
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    for e in err:
        if e is not None:
            raise e
    assert all(not t.is_alive() for t in th), "a rank hung"
    return out


def _split(N, G):
    nblk = (N + 63) // 64
    return [(nblk * r // G) * 64 for r in range(G)] + [N]


@pytest.mark.parametrize("G,N,K,Pl", [(2, 16384, 16, 4), (2, 32768, 128, 2), (4, 32768, 40, 2), (3, 24576, 8, 4), (2, 8192, 1, 2),
                                      (2, 16384, 2, 4), (3, 24576 + 37, 3, 2)])   # (K <= 3: the wavefront scan across ranks)
@pytest.mark.parametrize("variant", ["coupled", "decoupled"])
def test_sharded_apply_equals_single_and_oracle(spike, oracle, G, N, K, Pl, variant):
    import torch
    band = oracle.gen_band(N, K, delta=0.8)
    f = oracle.gen_vec(N)
    cuts = _split(N, G)

    def fn(r, sp):
        sp.set_option("partitions", Pl)
        sp.set_option("variant", variant)
        r0, r1 = cuts[r], cuts[r + 1]
        sp.setup_band(np.ascontiguousarray(band[:, r0:r1]), n_global=N, row0=r0)
        x = sp.apply(torch.from_numpy(f[r0:r1].copy()).cuda())
        torch.cuda.synchronize()
        i = sp.info()
        assert i.nranks == G and i.rank == r and i.n_global == N and i.row0 == r0
        return x.cpu().numpy()

    x = np.concatenate(_run_ranks(spike, G, fn))
    ref = oracle.Spike(band, G * Pl).apply(f, 1 if variant == "coupled" else 0)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)
    single = spike.Spike(partitions=G * Pl, variant=variant).setup_band(band).apply(f)
    assert np.linalg.norm(x - single) <= 1e-12 * np.linalg.norm(single)


@pytest.mark.parametrize("delta", [1.2, 0.8])
def test_unequal_row_blocks_take_the_same_collective_branches(spike, oracle, delta):
    """Ranks that own different row counts (4032 vs 4096 rows) would pick different sub-split factors on their own; the
    sub-split decision, its probe and the redo (delta = 0.8: the spikes do not die inside a chain) are collective, so
    every rank must walk the same branches (mismatched collectives).  Result = the 8-partition oracle."""
    import torch
    G, N, K, Pl = 2, 8128, 16, 4
    band = oracle.gen_band(N, K, delta=delta)
    f = oracle.gen_vec(N)
    cuts = _split(N, G)
    assert cuts[1] - cuts[0] != cuts[2] - cuts[1]

    def fn(r, sp):
        sp.set_option("partitions", Pl)
        r0, r1 = cuts[r], cuts[r + 1]
        sp.setup_band(np.ascontiguousarray(band[:, r0:r1]), n_global=N, row0=r0)
        x = sp.apply(torch.from_numpy(f[r0:r1].copy()).cuda())
        torch.cuda.synchronize()
        return x.cpu().numpy(), sp.info().chains_local

    res = _run_ranks(spike, G, fn)
    x = np.concatenate([r[0] for r in res])
    assert res[0][1] == res[1][1]                      # the same number of chains per partition on both ranks
    ref = oracle.Spike(band, G * Pl).apply(f, 1)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("G,N,K", [(2, 65536, 16), (3, 98304, 40), (2, 131072, 128), (2, 32768, 1), (2, 65536, 4), (2, 65536, 2), (3, 98304, 3)])
def test_overlapped_exchange_is_bit_identical_and_exact(spike, oracle, G, N, K):
    """Several ranks: the tip exchange (all-gather), the rank-boundary interface solves and the corrections they drive
    run on a second stream while the local interface solves and corrections run on the main one (the
    collective overlaps local work).  Same kernel code on the same operands in another launch order: bit-identical to
    the serial order; and on this dominant system truncated SPIKE equals the exact band solve."""
    import torch
    band = oracle.gen_band(N, K, delta=1.2)
    f = oracle.gen_vec(N)
    cuts = _split(N, G)
    def fn(r, sp):
        r0, r1 = cuts[r], cuts[r + 1]
        sp.setup_band(np.ascontiguousarray(band[:, r0:r1]), n_global=N, row0=r0)   # overlap_exchange is on by default
        fr = torch.from_numpy(f[r0:r1].copy()).cuda()
        xs = [sp.apply(fr) for _ in range(3)]                  # repeated: the second stream and its events are reused
        sp.set_option("overlap_exchange", "off")               # same chains, serial launch order
        xo = sp.apply(fr)
        torch.cuda.synchronize()
        assert all(torch.equal(xs[0], q) for q in xs[1:]) and torch.equal(xs[0], xo)
        return xs[0].cpu().numpy(), sp.info().chains_local

    res = _run_ranks(spike, G, fn)
    out = {"on": np.concatenate([r[0] for r in res])}
    assert all(r[1] >= 3 for r in res)
    exact = oracle.Spike(band, 1).apply(f, 0)
    assert np.linalg.norm(out["on"] - exact) <= 1e-10 * np.linalg.norm(exact)


def test_sharded_gmres_and_matvec(spike, oracle):
    import torch
    G, N, K, Pl = 2, 32768, 32, 8
    band = oracle.gen_band(N, K, delta=0.8)
    u = oracle.gen_vec(N, seed=9)
    b = oracle.band_matvec(band, u)
    cuts = _split(N, G)

    def fn(r, sp):
        sp.set_option("partitions", Pl)
        sp.set_option("variant", "decoupled")
        r0, r1 = cuts[r], cuts[r + 1]
        sp.setup_band(np.ascontiguousarray(band[:, r0:r1]), n_global=N, row0=r0)
        y = sp.matvec(torch.from_numpy(u[r0:r1].copy()).cuda())
        x = torch.zeros(r1 - r0, dtype=torch.float64, device="cuda")
        it, rn, ms, ok = sp.gmres(torch.from_numpy(b[r0:r1].copy()).cuda(), x, restart=30, rtol=1e-8, maxit=200)
        return y.cpu().numpy(), x.cpu().numpy(), it, ok

    res = _run_ranks(spike, G, fn)
    y = np.concatenate([r[0] for r in res])
    x = np.concatenate([r[1] for r in res])
    assert np.linalg.norm(y - b) <= 1e-14 * np.linalg.norm(b)          # halo exchange of the mat-vec
    assert res[0][2] == res[1][2] and res[0][3] and res[1][3]          # both ranks agree on the iteration count
    xo, ito, rno, hist, oko = oracle.gmres(band, b, oracle.Spike(band, G * Pl), variant=0, rtol=1e-8, maxit=200)
    assert abs(res[0][2] - ito) <= 1
    assert np.linalg.norm(x - u) <= 1e-6 * np.linalg.norm(u)


@pytest.mark.parametrize("G", [2, 3])
def test_row_block_distributed_csr_entry_equals_single_rank(spike, oracle, G):
    """spike_setup_csr_dist: every rank passes its row block (global columns) -- the layout MatCreateSubMatrixBanded is
    written for (/root/reference/src/matbanded.c:36, 74-75).  All ranks choose the single-rank k, the fraction agrees to
    rounding (per-rank sums combined in rank order), and the sharded apply equals the single-rank CSR entry and the oracle."""
    import torch
    import scipy.sparse as sps
    rng = np.random.default_rng(7)
    n, Pl = 6016 * 2, 4
    offs = list(range(-30, 31)) + [-200, 333]                   # a band plus two far diagonals the rule must cut off
    diags = [rng.uniform(-1, 1, n - abs(o)) * (0.6 ** min(abs(o), 40)) for o in offs]
    A = sps.diags(diags, offs, shape=(n, n), format="csr")
    A.setdiag(3.0)
    A = sps.csr_matrix(A)
    A.sort_indices()
    ia, ja, a = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data
    rhs = oracle.gen_vec(n)
    cuts = _split(n, G)
    single = spike.Spike(partitions=G * Pl)
    k1, f1 = single.setup_csr(n, ia, ja, a, kmax=50, frac=0.95)
    x1 = single.apply(rhs)
    ko, fo, ib, jb, bb = oracle.band_extract(n, ia, ja, a, 50, 0.95)
    assert (k1, f1) == (ko, fo)

    def fn(r, sp):
        sp.set_option("partitions", Pl)
        r0, r1 = cuts[r], cuts[r + 1]
        k, f = sp.setup_csr_dist(n, r0, ia[r0:r1 + 1] - ia[r0], ja[ia[r0]:ia[r1]], a[ia[r0]:ia[r1]], kmax=50, frac=0.95)
        x = sp.apply(torch.from_numpy(rhs[r0:r1].copy()).cuda())
        torch.cuda.synchronize()
        return k, f, x.cpu().numpy()

    res = _run_ranks(spike, G, fn)
    assert all(r[0] == k1 for r in res) and len({r[1] for r in res}) == 1      # same k, bitwise the same fraction on all ranks
    assert abs(res[0][1] - f1) <= 1e-12                  # another summation order of ~7e5 terms than the sequential rule
    x = np.concatenate([r[2] for r in res])
    assert np.linalg.norm(x - x1) <= 1e-12 * np.linalg.norm(x1)
    band = oracle.csr_to_band(n, ib, jb, bb, ko)
    ref = oracle.Spike(band, G * Pl).apply(rhs, 1)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)
