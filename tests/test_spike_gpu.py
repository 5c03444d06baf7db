"""GPU parity tests: the HIP path, called through the C-ABI (include/spike_mi355.h), against the CPU
oracle on the same seeded inputs.  Tolerance for the fp64 factor/solve: relative 2-norm 1e-10
(the north star's 'stated fp64 tolerance'); integer/bit results (generator, half-bandwidth) exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch


def test_generator_bit_exact(spike, oracle, torch_cuda):
    for (N, K, row0, nrows) in [(1000, 3, 0, 1000), (5000, 32, 1000, 2500), (777, 128, 0, 777)]:
        d = spike.gen_band_device(N, K, seed=12345, delta=1.2, row0=row0, nrows=nrows).cpu().numpy()
        h = oracle.gen_band(N, K, seed=12345, delta=1.2, row0=row0, nrows=nrows)
        assert np.array_equal(d, h)


CASES = [
    # N, K, P  (K picks the kernel configuration: R=8/16/32 multi-chain waves, R=64 with 2..8 waves)
    (16384, 1, 4),      # BASELINE config 1 shape (tridiagonal, 4 partitions)
    (4096, 1, 16),
    (3000, 5, 5),
    (8192, 8, 8),
    (5000, 13, 7),
    (8192, 16, 16),
    (10000, 32, 9),
    (65536, 32, 64),    # BASELINE config 2 shape at reduced N
    (9999, 50, 6),      # the reference's default kmax
    (16384, 64, 8),
    (12345, 100, 5),
    (32768, 128, 16),   # headline K
    (20000, 200, 4),
    (32768, 256, 8),    # BASELINE config 3 K
]


@pytest.mark.parametrize("N,K,P", CASES)
@pytest.mark.parametrize("delta", [1.2, 0.8])
def test_apply_matches_oracle(spike, oracle, torch_cuda, N, K, P, delta):
    band = oracle.gen_band(N, K, delta=delta)
    f = oracle.gen_vec(N)
    ref = oracle.Spike(band, P)
    for variant, vname in ((0, "decoupled"), (1, "coupled")):
        sp = spike.Spike(partitions=P, variant=vname)
        sp.setup_band(band)               # host pointers through the C-ABI
        x = sp.apply(f)
        xo = ref.apply(f, variant)
        assert _rel(x, xo) <= TOL, (vname, _rel(x, xo))
        info = sp.info()
        assert info.P_local == P and info.K == K and info.nboost == ref.nboost
        sp.close()


@pytest.mark.parametrize("N,K,P", [(2 ** 18, 1, 0), (2 ** 16 + 77, 1, 64), (4096, 1, 16), (2 ** 18, 2, 0), (2 ** 16, 3, 32),
                                   (2 ** 17, 4, 0), (2 ** 17 + 33, 8, 0), (2 ** 16, 5, 16)])
def test_narrow_band_coupling_step(spike, oracle, torch_cuda, N, K, P):
    """K <= 8, one rank, coupled, stored spikes: the coupling step can be k_tips_small + k_couple_small (every chain solves its
    two K x K interface systems itself) instead of k_iface_apply + k_spike_correct (default for K = 1).  Same preconditioner: equal to the
    oracle with the same partitions, and (dominant system) to the exact solve."""
    band = oracle.gen_band(N, K, delta=1.2)
    f = oracle.gen_vec(N)
    sp = spike.Spike(partitions=P, variant="coupled")
    sp.set_option("small_coupling_kmax", 8)        # default 1: the path pays at K = 1 only, the kernel serves K <= 8
    sp.setup_band(band)
    i = sp.info()
    assert i.passes == 1 and 0 < 2 * i.spike_rows <= (N // i.chains_local) // 64 * 64     # the small path's precondition
    x = sp.apply(f)
    if P:
        assert _rel(x, oracle.Spike(band, P).apply(f, 1)) <= 1e-10      # sub-chains reproduce the P-partition preconditioner to rounding
    else:
        assert _rel(x, oracle.Spike(band, i.P_local).apply(f, 1)) <= TOL
    assert _rel(x, oracle.Spike(band, 1).apply(f, 0)) <= 1e-10
    x2 = sp.apply(f)                                                            # repeated: the saved tips are rewritten each time
    assert np.array_equal(x, x2)


@pytest.mark.parametrize("N,K,P", [(16384, 8, 8), (65536, 32, 16), (65536, 64, 8), (2 ** 17, 128, 8)])
def test_coupled_one_pass_and_two_pass_agree(spike, oracle, torch_cuda, N, K, P):
    """The coupled variant keeps the decayed part of the spikes when they are short and then needs ONE pass over the
    factors (y = g - W x_b - V x_t); with spike_storage=off it re-solves (two passes).  Both must match the oracle."""
    band = oracle.gen_band(N, K, delta=0.8)
    f = oracle.gen_vec(N)
    xo = oracle.Spike(band, P).apply(f, 1)
    sp1 = spike.Spike(partitions=P)
    sp1.set_option("subsplit", "off")      # one chain per partition: long chains, short spikes -> one pass
    sp1.setup_band(band)
    i1 = sp1.info()
    assert i1.passes == 1 and 0 < i1.spike_rows <= N // P
    assert _rel(sp1.apply(f), xo) <= TOL
    sp2 = spike.Spike(partitions=P)
    sp2.set_option("subsplit", "off")
    sp2.set_option("spike_storage", "off")
    sp2.setup_band(band)
    i2 = sp2.info()
    assert i2.passes == 2 and i2.spike_rows == 0
    assert _rel(sp2.apply(f), xo) <= TOL


def test_subsplit_is_undone_when_spikes_do_not_die(spike, oracle, torch_cuda):
    """-1,2,-1: the library first cuts the 2 caller partitions into chains, measures that the spikes reach across a
    chain, and falls back to one chain per partition -- the result must be the 2-partition preconditioner."""
    N, P = 2 ** 15, 2
    band = np.zeros((3, N)); band[0, 1:] = -1.0; band[1, :] = 2.0; band[2, :-1] = -1.0
    f = oracle.gen_vec(N)
    sp = spike.Spike(partitions=P).setup_band(band)
    assert sp.info().chains_local == P and sp.info().P_local == P
    assert _rel(sp.apply(f), oracle.Spike(band, P).apply(f, 1)) <= 1e-9
    # dominant system, same sizes: the cut is kept and changes nothing beyond rounding
    band = oracle.gen_band(N, 1, delta=1.2)
    sp = spike.Spike(partitions=P).setup_band(band)
    assert sp.info().chains_local > P
    assert _rel(sp.apply(f), oracle.Spike(band, P).apply(f, 1)) <= 1e-10


def test_spikes_that_do_not_decay(spike, oracle, torch_cuda):
    """-1,2,-1 (not strictly dominant): the spikes decay only linearly, never below the drop level.  Narrow band:
    the library keeps the FULL spikes (cheap at K=1) -> still one pass; wide band: it must fall back to re-solving.
    Either way the result is the oracle's (same linear algebra)."""
    N, P = 8192, 8
    band = np.zeros((3, N)); band[0, 1:] = -1.0; band[1, :] = 2.0; band[2, :-1] = -1.0
    f = oracle.gen_vec(N)
    sp = spike.Spike(partitions=P).setup_band(band)
    assert _rel(sp.apply(f), oracle.Spike(band, P).apply(f, 1)) <= 1e-9
    i = sp.info()   # full spikes (2 doubles per row) against the scan path's 3 doubles per row: kept or not, same algebra
    assert (i.passes, i.spike_rows) in ((1, N // P), (2, 0))
    K = 40
    band = np.zeros((2 * K + 1, N))
    for d in range(2 * K + 1):
        band[d, :] = -1.0 / (1 + abs(d - K))
    band[K, :] = 2.0 * np.sum(1.0 / (1 + np.arange(1, K + 1))) * 1.0000001
    for d in range(2 * K + 1):          # zero the out-of-range corners
        off = d - K
        if off < 0: band[d, :-off] = 0.0
        if off > 0: band[d, N - off:] = 0.0
    sp = spike.Spike(partitions=P).setup_band(band)
    assert _rel(sp.apply(f), oracle.Spike(band, P).apply(f, 1)) <= 1e-8
    i = sp.info()   # full spikes cost 40/64 of a pass here: kept (1 pass) or not (2 passes) -- both are the same algebra
    assert (i.passes, i.spike_rows) in ((1, N // P), (2, 0))
    sp.set_option("spike_storage", "off")
    sp.setup_band(band)
    assert sp.info().passes == 2 and sp.info().spike_rows == 0
    assert _rel(sp.apply(f), oracle.Spike(band, P).apply(f, 1)) <= 1e-8


@pytest.mark.parametrize("N,K,P", [(8192, 8, 8), (16384, 64, 8), (32768, 128, 16)])
def test_spike_tips_match_oracle(spike, oracle, torch_cuda, N, K, P):
    band = oracle.gen_band(N, K, delta=0.8)
    ref = oracle.Spike(band, P)
    sp = spike.Spike(partitions=P).setup_band(band)
    V, W = sp.tips()
    Vo, Wo = ref.tips()
    assert np.abs(V - Vo).max() <= 1e-11 * max(np.abs(Vo).max(), 1)
    assert np.abs(W - Wo).max() <= 1e-11 * max(np.abs(Wo).max(), 1)


def test_device_pointers_and_reuse(spike, oracle, torch_cuda):
    torch = torch_cuda
    N, K, P = 40000, 40, 12
    band = oracle.gen_band(N, K)
    ref = oracle.Spike(band, P)
    sp = spike.Spike(partitions=P).setup_band(torch.from_numpy(band).cuda())
    for seed in (1, 2, 3):
        f = oracle.gen_vec(N, seed=seed)
        x = sp.apply(torch.from_numpy(f).cuda())
        torch.cuda.synchronize()
        assert _rel(x.cpu().numpy(), ref.apply(f, 1)) <= TOL
    # refactor with new values on the same handle (the reference calls PCSetUp(b->pc) every time, matbanded.c:178)
    band2 = oracle.gen_band(N, K, seed=99)
    sp.setup_band(band2)
    f = oracle.gen_vec(N)
    assert _rel(sp.apply(f), oracle.Spike(band2, P).apply(f, 1)) <= TOL


def test_single_partition_is_exact_band_solve(spike, oracle, torch_cuda):
    from scipy.linalg import solve_banded
    N, K = 5000, 24
    band = oracle.gen_band(N, K, delta=0.8)
    b = oracle.band_matvec(band, np.ones(N))
    sp = spike.Spike(partitions=1).setup_band(band)
    x = sp.apply(b)
    xe = solve_banded((K, K), oracle.to_lapack_ab(band), b)
    assert _rel(x, xe) <= 1e-11


def test_pivot_boost_counts_match(spike, oracle, torch_cuda):
    N, K, P = 4096, 4, 4
    band = oracle.gen_band(N, K)
    for i in (100, 2000, 3000):
        band[K, i] = 0.0
        band[:K, i] = 0.0
    ref = oracle.Spike(band, P, boost_rel=1e-8)
    sp = spike.Spike(partitions=P, boost=1e-8).setup_band(band)
    assert sp.info().nboost == ref.nboost >= 3
    f = oracle.gen_vec(N)
    assert _rel(sp.apply(f), ref.apply(f, 1)) <= 1e-8


def test_errors(spike, oracle, torch_cuda):
    band = oracle.gen_band(1000, 8)
    sp = spike.Spike(partitions=64)
    with pytest.raises(spike.SpikeError):
        sp.setup_band(band)           # 1000 rows = 16 blocks < 64 partitions
    sp2 = spike.Spike(partitions=10)
    with pytest.raises(spike.SpikeError):
        sp2.setup_band(oracle.gen_band(1300, 200))   # partitions shorter than K
    sp3 = spike.Spike()
    with pytest.raises(spike.SpikeError):
        sp3.apply(np.ones(10))        # apply before setup
    with pytest.raises(spike.SpikeError):
        sp3.set_option("nonsense", 1)


def test_matvec_and_gmres_match_oracle(spike, oracle, torch_cuda):
    torch = torch_cuda
    N, K, P = 32768, 32, 32
    band = oracle.gen_band(N, K, delta=0.8)
    u = np.ones(N)
    b = oracle.band_matvec(band, u)
    sp = spike.Spike(partitions=P).setup_band(band)
    db = sp.matvec(torch.from_numpy(u).cuda())
    assert _rel(db.cpu().numpy(), b) <= 1e-14
    ref = oracle.Spike(band, P)
    for variant, vname in ((1, "coupled"), (0, "decoupled")):
        sp.set_option("variant", vname)
        x = torch.zeros(N, dtype=torch.float64, device="cuda")
        it, rn, ms, ok = sp.gmres(torch.from_numpy(b).cuda(), x, restart=30, rtol=1e-5, maxit=500)
        xo, ito, rno, hist, oko = oracle.gmres(band, b, ref, variant=variant)
        assert ok and oko and abs(it - ito) <= 1, (it, ito)
        # the reference's own acceptance check: ||x - u|| (src/testbed2.c:130-132)
        assert np.linalg.norm(x.cpu().numpy() - u) <= 10 * max(np.linalg.norm(xo - u), 1e-12) + 1e-9


def test_setup_csr_uses_reference_rule(spike, oracle, torch_cuda):
    rng = np.random.default_rng(5)
    n = 6000
    rows, cols, vals = [], [], []
    for i in range(n):
        for j in range(max(0, i - 30), min(n, i + 31)):
            v = rng.uniform(-1, 1) * 0.6 ** abs(i - j)
            if i == j:
                v = 3.0
            rows.append(i); cols.append(j); vals.append(v)
    ia = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ia, np.array(rows) + 1, 1)
    ia = np.cumsum(ia)
    ja, a = np.array(cols), np.array(vals)
    sp = spike.Spike(partitions=4)
    k, f = sp.setup_csr(n, ia, ja, a, kmax=50, frac=0.95)
    ko, fo, ib, jb, bb = oracle.band_extract(n, ia, ja, a, 50, 0.95)
    assert k == ko and f == fo
    band = oracle.csr_to_band(n, ib, jb, bb, ko)
    rhs = oracle.gen_vec(n)
    assert _rel(sp.apply(rhs), oracle.Spike(band, 4).apply(rhs, 1)) <= TOL
    assert sp.info().k_extracted == ko
    # the 32-bit index entry point (PETSc's default PetscInt) is the same setup
    sp32 = spike.Spike(partitions=4)
    k32, f32 = sp32.setup_csr32(n, ia, ja, a, kmax=50, frac=0.95)
    assert (k32, f32) == (k, f)
    assert np.array_equal(sp32.apply(rhs), sp.apply(rhs))
    # a repeated (row, column) pair keeps the LAST stored value (INSERT_VALUES, matbanded.c:98), the same bits every run
    ia2 = ia.copy(); ia2[1:] += 1
    ja2 = np.concatenate([[ja[0]], ja]); a2 = np.concatenate([[123.0], a])      # row 0 stores its first entry twice
    spd = spike.Spike(partitions=4)
    kd, fd = spd.setup_csr(n, ia2, ja2, a2, kmax=50, frac=1.0)
    assert kd == 30
    band_d = oracle.csr_to_band(n, ia2, ja2, a2, kd)
    assert band_d[kd + int(ja[0]), 0] == a[0]
    x1 = spd.apply(rhs)
    assert _rel(x1, oracle.Spike(band_d, 4).apply(rhs, 1)) <= TOL
    spd.setup_csr(n, ia2, ja2, a2, kmax=50, frac=1.0)
    assert np.array_equal(spd.apply(rhs), x1)


def test_full_size_properties(spike, oracle, torch_cuda):
    """BASELINE headline size (N = 4*2^20, K = 128): too large for the oracle in a test, so check
    size-independent properties: M^{-1}(A u) == u (round trip), linearity, and that the coupled variant
    drives the true residual to rounding level on the diagonally dominant synthetic system."""
    torch = torch_cuda
    N, K = 4 * 2 ** 20, 128
    band = spike.gen_band_device(N, K, seed=12345, delta=1.2)
    sp = spike.Spike(partitions=0, variant="coupled").setup_band(band)
    u = torch.ones(N, dtype=torch.float64, device="cuda")
    b = sp.matvec(u)
    x = sp.apply(b)
    torch.cuda.synchronize()
    assert float((x - u).abs().max()) <= 1e-9
    v = torch.from_numpy(oracle.gen_vec(N)).cuda()
    bv = sp.matvec(v)
    xv = sp.apply(bv)
    xsum = sp.apply(b + 2.0 * bv)
    assert float((xsum - (x + 2.0 * xv)).abs().max()) <= 1e-9
    r = sp.matvec(xv) - bv
    assert float(r.norm() / bv.norm()) <= 1e-12
    info = sp.info()
    assert info.Kp == 128 and info.nboost == 0


@pytest.mark.parametrize("N,K,P", [(64, 0, 1), (1000, 0, 3), (64, 1, 1), (65, 3, 1), (100, 5, 1), (129, 8, 2), (200, 40, 1),
                                   (300, 100, 1), (1000, 128, 2), (5000, 256, 3), (4097, 33, 4), (777, 17, 3),
                                   (3000, 200, 2), (2103, 160, 1), (1377, 256, 1),
                                   (2103, 260, 1), (3000, 300, 2), (4096, 384, 2), (5000, 512, 3), (1100, 450, 1)])
def test_edge_sizes(spike, oracle, torch_cuda, N, K, P):
    """diagonal matrices (K = 0), a single 64-row block, ragged last blocks, partitions barely longer than K,
    unequal chain lengths (129 rows in 2 partitions: the stored-spike window must not drop the longer chain's tail), wide bands
    on chains with an odd number of 16-row blocks (the two-steps-per-pass factorisation ends on a single step), and (round 3)
    half-bandwidths above 256 -- -pc_banded_kmax is unbounded in the reference (matbanded.c:156): 260 ... 512 take the
    64-diagonals-per-wave sweeps and the generic setup paths."""
    band = oracle.gen_band(N, K, delta=0.9)
    f = oracle.gen_vec(N)
    sp = spike.Spike(partitions=P).setup_band(band)
    assert _rel(sp.apply(f), oracle.Spike(band, P).apply(f, 1)) <= 1e-13


@pytest.mark.parametrize("N,K,P", [(8192, 16, 4), (8192, 40, 4), (16384, 128, 2), (16384, 200, 2)])
def test_pivot_boost_in_the_blocked_mfma_factorisation(spike, oracle, torch_cuda, N, K, P):
    """rows whose diagonal and sub-diagonal entries are zero produce an exactly zero pivot; the blocked (MFMA) LU must
    boost the same pivots as the oracle's scalar LU and give the same preconditioner"""
    band = oracle.gen_band(N, K)
    for i in (777, N // 2 + 5, N - 300):
        band[:K + 1, i] = 0.0
    ref = oracle.Spike(band, P, boost_rel=1e-3)   # a mild boost keeps the boosted factors well conditioned
    sp = spike.Spike(partitions=P, boost=1e-3)
    sp.set_option("subsplit", "off")
    sp.setup_band(band)
    assert sp.info().nboost == ref.nboost >= 3
    f = oracle.gen_vec(N)
    assert _rel(sp.apply(f), ref.apply(f, 1)) <= 1e-7   # 1/boost amplifies rounding differences of the two LU orders


def test_gmres_on_a_nearby_banded_operator(spike, oracle, torch_cuda):
    """the preconditioner is built from A (delta 1.2), the Krylov solve runs on A' (delta 1.0, same off-diagonals):
    spike_set_operator_band + spike_gmres must solve A' x = b, and take more than one iteration"""
    torch = torch_cuda
    N, K, P = 32768, 24, 16
    A = oracle.gen_band(N, K, delta=1.2)
    A2 = oracle.gen_band(N, K, delta=1.0)
    u = oracle.gen_vec(N, seed=3)
    sp = spike.Spike(partitions=P).setup_band(A)
    sp.set_operator_band(torch.from_numpy(A2).cuda())
    b2 = sp.operator_matvec(torch.from_numpy(u).cuda())
    assert _rel(b2.cpu().numpy(), oracle.band_matvec(A2, u)) <= 1e-14
    x = torch.zeros(N, dtype=torch.float64, device="cuda")
    it, rn, ms, ok = sp.gmres(b2, x, restart=30, rtol=1e-10, maxit=200)
    assert ok and 2 <= it <= 60
    assert _rel(x.cpu().numpy(), u) <= 1e-7
    xo, ito, *_ = oracle.gmres(A2, oracle.band_matvec(A2, u), oracle.Spike(A, P), variant=1, rtol=1e-10, maxit=200)
    assert abs(it - ito) <= 1
    sp.set_operator_band(None)
    assert _rel(sp.matvec(torch.from_numpy(u).cuda()).cpu().numpy(), oracle.band_matvec(A, u)) <= 1e-14


def test_gmres_refinement_types_and_bitwise_reproducibility(spike, oracle, torch_cuda):
    """PETSc's -ksp_gmres_cgs_refinement_type names; the Krylov reductions are fixed-order (no floating-point atomics),
    so the same solve gives the same BITS twice -- with and without the preconditioner, n not a multiple of anything"""
    torch = torch_cuda
    N, K, P = 40000 + 37, 24, 8
    A = oracle.gen_band(N, K, delta=1.2)
    A2 = oracle.gen_band(N, K, delta=0.9)
    u = oracle.gen_vec(N, seed=5)
    sp = spike.Spike(partitions=P).setup_band(A)
    sp.set_operator_band(torch.from_numpy(A2).cuda())
    b = sp.operator_matvec(torch.from_numpy(u).cuda())
    sols = {}
    for rt in ("refine_never", "refine_ifneeded", "refine_always"):
        sp.set_option("gmres_cgs_refinement_type", rt)
        x = torch.zeros(N, dtype=torch.float64, device="cuda")
        it, rn, ms, ok = sp.gmres(b, x, restart=30, rtol=1e-11, maxit=300)
        assert ok and it >= 3
        assert _rel(x.cpu().numpy(), u) <= 1e-8
        x2 = torch.zeros(N, dtype=torch.float64, device="cuda")
        it2, rn2, _, ok2 = sp.gmres(b, x2, restart=30, rtol=1e-11, maxit=300)
        assert it2 == it and rn2 == rn and torch.equal(x, x2)            # bit for bit
        sols[rt] = (it, x.cpu().numpy())
    assert abs(sols["refine_never"][0] - sols["refine_always"][0]) <= 1
    with pytest.raises(Exception):
        sp.set_option("gmres_cgs_refinement_type", "sometimes")
    # unpreconditioned, restart shorter than the iteration count (restarts exercise the pending normalisation)
    sp.set_option("gmres_cgs_refinement_type", "refine_never")
    x = torch.zeros(N, dtype=torch.float64, device="cuda")
    it, rn, ms, ok = sp.gmres(b, x, restart=5, rtol=1e-9, maxit=400, use_pc=False)
    assert ok and it > 5 and _rel(x.cpu().numpy(), u) <= 1e-6


@pytest.mark.parametrize("restart", [1, 2, 33, 64])
def test_gmres_restart_lengths(spike, oracle, torch_cuda, restart):
    """restart lengths around the 32-vector reduction launch and the 64-coefficient update kernel; > 64 is refused"""
    torch = torch_cuda
    N, K, P = 6000, 5, 4
    A = oracle.gen_band(N, K, delta=1.3)
    A2 = oracle.gen_band(N, K, delta=0.7)
    u = oracle.gen_vec(N, seed=9)
    sp = spike.Spike(partitions=P).setup_band(A)
    sp.set_operator_band(torch.from_numpy(A2).cuda())
    b = sp.operator_matvec(torch.from_numpy(u).cuda())
    x = torch.zeros(N, dtype=torch.float64, device="cuda")
    it, rn, ms, ok = sp.gmres(b, x, restart=restart, rtol=1e-10, maxit=2000)
    assert ok, (it, rn)
    assert _rel(x.cpu().numpy(), u) <= 1e-6
    # unpreconditioned with a long basis: every reduction width up to restart+2 is exercised
    x.zero_()
    it, rn, ms, ok = sp.gmres(b, x, restart=restart, rtol=1e-8, maxit=4000, use_pc=False)
    if restart >= 33:
        assert ok and it > 33 and _rel(x.cpu().numpy(), u) <= 1e-5
    with pytest.raises(Exception):
        sp.gmres(b, x, restart=65, rtol=1e-8, maxit=10)


@pytest.mark.parametrize("K,P", [(33, 8), (65, 4), (90, 4), (91, 3), (100, 4), (127, 4), (129, 2), (255, 2)])
def test_odd_half_bandwidths(spike, oracle, torch_cuda, K, P):
    """half-bandwidths around the configuration boundaries (32/64/96/128/256 streamed diagonals) and around the
    64-KiB LDS boundary of the in-LDS interface inversion (K = 90 / 91)"""
    torch = torch_cuda
    N = 64 * 40 * P
    band = oracle.gen_band(N, K, delta=0.9)
    f = oracle.gen_vec(N)
    sp = spike.Spike(partitions=P, variant="coupled").setup_band(torch.from_numpy(band).cuda())
    x = sp.apply(torch.from_numpy(f).cuda()).cpu().numpy()
    xo = oracle.Spike(band, P).apply(f, 1)
    assert _rel(x, xo) <= 1e-10


def test_random_shapes_match_oracle(spike, oracle, torch_cuda):
    """Regression net over the configuration space: 36 seeded random (N, K, partitions, dominance, variant) cases -- ragged
    sizes, every kernel family (scan, narrow tiles, two-chains-per-wave, look-ahead and in-place factorisations, strip pack,
    block-TRSM, stored spikes and re-solve) -- against the oracle with the same partitions."""
    rng = np.random.default_rng(20261004)
    ks = [1, 2, 3, 5, 8, 13, 16, 27, 32, 33, 48, 64, 77, 96, 128, 129, 160, 200, 255, 256]
    for case in range(36):
        K = int(ks[case % len(ks)] if case < len(ks) else rng.choice(ks))
        P = int(rng.integers(1, 7))
        rows = int(rng.integers(max(2 * K + 2, 70), max(2 * K + 3, 3000)))      # rows per partition, roughly
        N = P * rows + int(rng.integers(0, 64))
        delta = float(rng.choice([0.7, 0.9, 1.2]))
        variant = "coupled" if case % 3 else "decoupled"
        band = oracle.gen_band(N, K, seed=100 + case, delta=delta)
        f = oracle.gen_vec(N, seed=7 + case)
        sp = spike.Spike(partitions=P, variant=variant).setup_band(band)
        x = sp.apply(f)
        ref = oracle.Spike(band, sp.info().P_local).apply(f, 1 if variant == "coupled" else 0)
        assert _rel(x, ref) <= 1e-10, (case, N, K, P, delta, variant, _rel(x, ref))
        sp.close()


@pytest.mark.parametrize("N,K,P", [(2 ** 13, 384, 4), (2 ** 12, 512, 2)])
def test_wide_bands_above_256(spike, oracle, torch_cuda, N, K, P):
    """256 < K <= 512 (supported, not tuned): against the oracle for the caller's partitions, both variants, and the
    exact-solution round trip on the dominant system"""
    band = oracle.gen_band(N, K, delta=1.2)
    f = oracle.gen_vec(N)
    ref = oracle.Spike(band, P)
    for variant, vname in ((1, "coupled"), (0, "decoupled")):
        sp = spike.Spike(partitions=P, variant=vname).setup_band(band)
        assert _rel(sp.apply(f), ref.apply(f, variant)) <= TOL
        i = sp.info()
        assert i.K == K and i.Kp == 64 * ((K + 63) // 64) and i.nboost == ref.nboost
        sp.close()
    sp = spike.Spike(partitions=0).setup_band(band)
    u = np.ones(N)
    assert np.abs(sp.apply(oracle.band_matvec(band, u)) - u).max() <= 1e-10
    with pytest.raises(spike.SpikeError):
        spike.Spike(partitions=1).setup_band(oracle.gen_band(2048, 513, delta=1.2))


@pytest.mark.parametrize("K", [1, 2, 3])
def test_wavefront_scan_four_rows_per_lane(spike, oracle, torch_cuda, K):
    """K <= 3 (round 3): k_nscan_solve / k_nscan_sweep (a lane owns 4 consecutive rows, one 64-lane scan of K x K affine maps per
    256 rows).  Against the oracle for: every register-resident length class (chains of <= 1024 / 2048 / 4096 rows) and the
    two-launch form (longer chains), chain lengths that are no multiple of 4 or 256, vectors at 8-byte-only alignment (the
    scalar load / store variant), both variants (the two-pass form sends corrections into the kernel), and against the
    tile path / the one-row-per-lane scan it replaces (options narrow_scan_kmax / narrow_scan_rows)."""
    torch = torch_cuda
    for (N, P, sub) in [(1000, 1, "off"), (4099, 3, "off"), (2 ** 13 + 5, 4, "off"), (2 ** 14, 4, "off"), (20001, 2, "off"),
                        (70001, 7, "on"), (2 ** 18 + 3, 0, "on")]:
        band = oracle.gen_band(N, K, delta=1.1)
        f = oracle.gen_vec(N, seed=N % 97)
        for vname, variant in (("coupled", 1), ("decoupled", 0)):
            sp = spike.Spike(partitions=P, variant=vname)
            sp.set_option("subsplit", sub)
            sp.setup_band(band)
            i = sp.info()
            x = sp.apply(f)
            if P and sub == "off":
                assert i.chains_local == P
                assert _rel(x, oracle.Spike(band, P).apply(f, variant)) <= TOL, (N, P, vname)
            elif variant == 1:
                assert _rel(x, oracle.Spike(band, 1).apply(f, 0)) <= 1e-10, (N, P, vname)    # dominant: the exact solve to rounding
            # vectors at an odd multiple of 8 bytes
            fb = torch.zeros(N + 1, dtype=torch.float64, device="cuda"); fb[1:] = torch.from_numpy(f).cuda()
            xb = torch.zeros(N + 1, dtype=torch.float64, device="cuda")
            sp.apply(fb[1:], xb[1:])
            torch.cuda.synchronize()
            assert np.array_equal(xb[1:].cpu().numpy(), x), (N, P, vname, "alignment variant differs")
            # the two-pass form of the coupled variant (no stored spikes: the corrections enter the scan kernel's right-hand side)
            if variant == 1:
                s2 = spike.Spike(partitions=P, variant=vname)
                s2.set_option("subsplit", sub); s2.set_option("spike_storage", "off")
                s2.setup_band(band)
                i2 = s2.info()
                assert i2.spike_rows == 0 and i2.passes == (2 if i2.chains_local > 1 else 1)
                assert _rel(s2.apply(f), x) <= TOL, (N, P, "two passes")
                s2.close()
            # the paths it replaces
            alts = [("narrow_scan_kmax", 1)] if K > 1 else [("narrow_scan_rows", 1)]
            for key, val in alts:
                so = spike.Spike(partitions=P, variant=vname)
                so.set_option("subsplit", sub); so.set_option(key, val); so.set_option("twist", "off")
                so.setup_band(band)
                assert _rel(so.apply(f), x) <= TOL, (N, P, vname, key)
                so.close()
            sp.close()


def test_workspace_is_recycled_between_setups(spike, oracle, torch_cuda):
    """Refactorisation on one handle (the reference: PCSetUp(b->pc) on every call, matbanded.c:178): the handle's device blocks
    are handed out again by size instead of going through hipFree / hipMalloc (option workspace_cache, default on).  Same
    results as without recycling, bit for bit; a changed shape does not pile up memory; spike_reset gives everything back."""
    torch = torch_cuda
    N, K, P = 2 ** 16, 40, 8
    f = oracle.gen_vec(N)
    bands = [oracle.gen_band(N, K, seed=s) for s in (1, 2, 3)]
    res = {}
    for mode in ("on", "off"):
        sp = spike.Spike(partitions=P)
        sp.set_option("workspace_cache", mode)
        out = []
        for b in bands + [bands[0]]:
            sp.setup_band(b)
            out.append(sp.apply(f))
        res[mode] = out
        assert np.array_equal(out[0], out[3])                       # the same matrix again: the same bits
        sp.close()
    for a, b in zip(res["on"], res["off"]):
        assert np.array_equal(a, b)
    for b, x in zip(bands, res["on"]):
        assert _rel(x, oracle.Spike(b, P).apply(f, 1)) <= TOL
    # shapes that change from setup to setup: idle blocks of a shape no longer in use are released after one setup
    sp = spike.Spike()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    held = []
    for (n, k) in [(2 ** 18, 64), (2 ** 16, 8), (2 ** 18, 64), (2 ** 15, 128), (2 ** 15, 128), (2 ** 15, 128)]:
        band = spike.gen_band_device(n, k, seed=5, delta=1.2)
        sp.setup_band(band)
        del band
        torch.cuda.synchronize(); torch.cuda.empty_cache()           # (torch's cache of the generated band is not the library's)
        held.append(free0 - torch.cuda.mem_get_info()[0])
    assert held[-1] == held[-2]                                      # steady state: nothing grows
    assert held[-1] < held[0]                                        # the large first shape did not stay behind
    assert sp.L.spike_reset(sp.h) == 0
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    assert free0 - torch.cuda.mem_get_info()[0] <= 64 * 2 ** 20
    sp.close()


@pytest.mark.parametrize("K,shape", [(96, "16,6,2"), (96, "16,6,4"), (128, "16,8,2"), (128, "32,4,4"), (128, "16,8,4"), (200, "16,16,2"), (256, "16,16,2")])
def test_sweep_shapes_match_oracle(spike, oracle, torch_cuda, K, shape):
    """The shapes the setup-time measurement may choose between (diagonals per wave x waves x bundles in flight; measurement
    knob SPIKE_SWEEP_SHAPE pins one): same preconditioner as the oracle's, whichever shape sweeps."""
    import os
    N, P = 2 ** 15, 4
    band = oracle.gen_band(N, K, delta=1.1)
    f = oracle.gen_vec(N)
    xo = oracle.Spike(band, P).apply(f, 1)
    os.environ["SPIKE_SWEEP_SHAPE"] = shape
    try:
        sp = spike.Spike(partitions=P).setup_band(band)
    finally:
        os.environ.pop("SPIKE_SWEEP_SHAPE", None)
    assert _rel(sp.apply(f), xo) <= TOL
    sp.close()


def test_sweep_shape_is_timed_at_setup_for_long_chains(spike, oracle, torch_cuda):
    """K > 64 and chains of >= 8192 rows: the first setup of a shape times the candidate sweep shapes on the real factors and keeps
    the fastest (a refactorisation keeps the choice); option sweep_autotune = off pins the base shape.  Same result to rounding."""
    torch = torch_cuda
    N, K = 2 ** 21, 128
    band = spike.gen_band_device(N, K, seed=7, delta=1.2)
    u = torch.ones(N, dtype=torch.float64, device="cuda")
    sp = spike.Spike().setup_band(band)
    assert "sweep shape =" in sp.view() and sp.info().chains_local * 8192 <= N
    b = sp.matvec(u)
    x = sp.apply(b)
    assert float((x - u).abs().max()) <= 1e-10
    t1 = sp.info().setup_ms
    sp.setup_band(band)                       # refactorisation: no second timing
    assert sp.info().setup_ms < t1 and "sweep shape =" in sp.view()
    x2 = sp.apply(b)
    assert torch.equal(x, x2)
    off = spike.Spike()
    off.set_option("sweep_autotune", "off")
    off.setup_band(band)
    assert "sweep shape =" not in off.view()
    xo = off.apply(b)
    assert float((x - xo).abs().max()) <= 1e-12
