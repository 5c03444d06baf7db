"""Twisted (two-ended) factorisation: chains 2t / 2t+1 are the top and bottom half of one diagonal block, factored from its
two ends; an exact 2K x 2K seam system links the halves between the inward and the outward sweep launch, and only the
blocks' outer ends are truncated interfaces with stored spikes.  The preconditioner for a caller-chosen
partition count must not change: every case is compared with the CPU oracle for the caller's P (relative 2-norm 1e-10, the
stated fp64 tolerance) and with the library's own untwisted result."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _mk(spike, P, variant="coupled", twist="auto", **opts):
    sp = spike.Spike(partitions=P, variant=variant)
    sp.set_option("twist", twist)
    for k, v in opts.items():
        sp.set_option(k, v)
    return sp


# N, K, P: every tile configuration (R = 4/8/16/32: several chains per wave, seam matrices off the diagonal-major LU scratch;
# R = 64 with 2..8 waves: seam matrices from the block TRSM), caller partitions that get cut into 2, 4, 6 ... chains, odd
# sub-split factors, partitions with an odd number of 64-row blocks (halves of unequal length), a ragged last block (K <= 32)
CASES = [
    (2 ** 16, 2, 4), (2 ** 16, 3, 7), (2 ** 16 + 33, 4, 8), (2 ** 17, 8, 16), (2 ** 16, 13, 5), (2 ** 17 + 21, 16, 8), (2 ** 17, 32, 8),
    (2 ** 17, 50, 4), (2 ** 17, 64, 8), (3 * 2 ** 15, 100, 3), (2 ** 18, 128, 8), (2 ** 17, 128, 5), (2 ** 17 + 64 * 7, 128, 3),
    (2 ** 16, 200, 4), (2 ** 15, 256, 2),
]


@pytest.mark.parametrize("N,K,P", CASES)
def test_twisted_equals_oracle_and_untwisted(spike, oracle, N, K, P):
    band = oracle.gen_band(N, K, delta=1.2)
    f = oracle.gen_vec(N)
    ref = oracle.Spike(band, P)
    narrow = {"narrow_scan_kmax": 1} if K <= 3 else {}     # K = 2, 3 default to the wavefront scan (no tiles, nothing to twist)
    for variant, vname in ((1, "coupled"), (0, "decoupled")):
        xo = ref.apply(f, variant)
        tw = _mk(spike, P, vname, "auto", **narrow).setup_band(band)
        assert "(twisted pairs)" in tw.view(), tw.view()          # the case really exercises the twisted path
        i = tw.info()
        assert i.twisted == 1 and i.seams_local * 2 == i.chains_local and 0 < i.spike_rows_fp64 <= i.spike_rows
        assert i.P_local == P and i.passes == 1 and i.chains_local % (2 * P) == 0 and i.nboost == ref.nboost
        xt = tw.apply(f)
        assert _rel(xt, xo) <= TOL, (vname, _rel(xt, xo))
        off = _mk(spike, P, vname, "off", **narrow).setup_band(band)
        assert "(twisted pairs)" not in off.view() and off.info().twisted == 0 and off.info().seams_local == 0
        xu = off.apply(f)
        assert _rel(xu, xo) <= TOL
        assert _rel(xt, xu) <= TOL
        # half the stored-spike traffic for the same number of chains
        io = off.info()
        if io.chains_local == i.chains_local and io.spike_rows == i.spike_rows:
            assert 2 * i.spike_bytes == io.spike_bytes
        tw.close(); off.close()


@pytest.mark.parametrize("N,K", [(2 ** 19, 128), (2 ** 19, 32), (2 ** 20, 8), (2 ** 19, 64), (2 ** 19 + 640, 256)])
def test_auto_partitions_twisted_is_exact_on_dominant_systems(spike, oracle, N, K):
    """partitions = 0: the library pairs its own chains.  On the dominant system truncated SPIKE equals the band solve to
    rounding: M^-1 (A 1) = 1, residual of a random solve, linearity; tips of the caller-level interfaces in natural orientation."""
    import torch
    band = spike.gen_band_device(N, K, seed=12345, delta=1.2)
    sp = _mk(spike, 0).setup_band(band)
    assert "(twisted pairs)" in sp.view()
    i = sp.info()
    assert i.chains_local == 2 * i.P_local and i.passes == 1 and i.spike_rows > 0
    u = torch.ones(N, dtype=torch.float64, device="cuda")
    b = sp.matvec(u)
    x = sp.apply(b)
    assert float((x - u).abs().max()) <= 1e-10
    v = torch.from_numpy(oracle.gen_vec(N)).cuda()
    bv = sp.matvec(v)
    xv = sp.apply(bv)
    assert float((sp.matvec(xv) - bv).norm() / bv.norm()) <= 1e-12
    assert float((sp.apply(b + 3.0 * bv) - (x + 3.0 * xv)).abs().max()) <= 1e-9
    # same partitions without twisting: the same preconditioner to rounding
    off = _mk(spike, i.P_local, twist="off", subsplit="off").setup_band(band)
    assert float((off.apply(bv) - xv).abs().max()) <= 1e-10
    if i.P_local > 1:
        Vt, Wt = sp.tips()
        Vo, Wo = off.tips()
        assert np.abs(Vt - Vo).max() <= 1e-12 * max(1.0, np.abs(Vo).max()) and np.abs(Wt - Wo).max() <= 1e-12 * max(1.0, np.abs(Wo).max())


def test_slowly_decaying_spikes_fall_back_to_ordinary_chains(spike, oracle):
    """the -1, 2, -1 stencil: spikes never die inside a chain => setup measures that and starts over untwisted (and without
    cutting the caller's partitions); the result is still the P-partition preconditioner"""
    N, P = 2 ** 15, 4
    band = np.zeros((5, N))
    band[1, 1:] = -1.0; band[2, :] = 2.0; band[3, :-1] = -1.0          # K = 2 storage of a tridiagonal matrix
    f = oracle.gen_vec(N)
    sp = _mk(spike, P).setup_band(band)
    assert "(twisted pairs)" not in sp.view() and sp.info().chains_local == P
    assert _rel(sp.apply(f), oracle.Spike(band, P).apply(f, 1)) <= 1e-8


@pytest.mark.parametrize("K,G,Pl", [(128, 4, 2), (16, 3, 4), (64, 2, 0)])
def test_twisted_across_ranks(spike, oracle, K, G, Pl):
    """thread ranks over the loopback transport: rank-boundary interfaces lie between a bottom half (flipped) of one rank
    and a top half of the next -- the exchanged tips and the gathered [W_first | V_last] are in natural orientation"""
    import torch
    n_rank = 2 ** 16
    N = G * n_rank
    band = oracle.gen_band(N, K, seed=12345, delta=1.2)
    f = oracle.gen_vec(N)
    out, views, infos, err = [None] * G, [None] * G, [None] * G, [None] * G

    def work(r):
        try:
            sp = _mk(spike, Pl)
            sp.comm_init_local(G, r, 4200 + K)
            r0 = r * n_rank
            sp.setup_band(np.ascontiguousarray(band[:, r0:r0 + n_rank]), n_global=N, row0=r0)
            views[r] = sp.view(); infos[r] = sp.info().P_local
            out[r] = sp.apply(torch.from_numpy(f[r0:r0 + n_rank].copy()).cuda()).cpu().numpy()
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    for e in err:
        if e is not None:
            raise e
    assert all("(twisted pairs)" in v for v in views), views
    x = np.concatenate(out)
    assert _rel(x, oracle.Spike(band, sum(infos)).apply(f, 1)) <= TOL


def test_twisted_gmres_iteration_counts(spike, oracle):
    """the twisted PCApply inside the Krylov loop: iteration counts as the oracle's GMRES with the P-partition preconditioner"""
    import torch
    N, K, P = 2 ** 17, 32, 8
    band = oracle.gen_band(N, K, delta=0.8)
    u = np.ones(N)
    b = oracle.band_matvec(band, u)
    sp = _mk(spike, P).setup_band(band)
    assert "(twisted pairs)" in sp.view()
    x = torch.zeros(N, dtype=torch.float64, device="cuda")
    it, rn, ms, ok = sp.gmres(torch.from_numpy(b).cuda(), x, restart=30, rtol=1e-5, maxit=500)
    xo, ito, rno, hist, oko = oracle.gmres(band, b, oracle.Spike(band, P), variant=1)
    assert ok and oko and abs(it - ito) <= 1
    assert np.abs(x.cpu().numpy() - u).max() <= 1e-4


@pytest.mark.parametrize("N,K,P,twist", [(2 ** 19, 128, 0, "auto"), (2 ** 19, 128, 16, "off"), (2 ** 20, 32, 64, "auto"), (2 ** 20, 16, 0, "off"),
                                         (2 ** 19, 64, 8, "auto"), (2 ** 20, 4, 0, "auto")])
def test_fp32_spike_tail_changes_nothing_visible(spike, oracle, N, K, P, twist):
    """mixed-precision spike storage: where every entry of a window is below 2^-28 of the spikes' peak the window is kept in
    fp32 (rounding 2^-52 of the peak).  Same result as all-fp64 storage to ~1e-15, fewer bytes; and the drop level default
    (1e-13) against the old 1e-16: invisible at the 1e-10 parity bar by orders of magnitude."""
    import torch
    band = spike.gen_band_device(N, K, seed=12345, delta=1.2)
    f = torch.from_numpy(oracle.gen_vec(N)).cuda()
    a = _mk(spike, P, twist=twist).setup_band(band)
    b = _mk(spike, P, twist=twist, spike_fp32="off").setup_band(band)
    c = _mk(spike, P, twist=twist, spike_fp32="off", spike_tol="1e-16").setup_band(band)
    ia, ib, ic = a.info(), b.info(), c.info()
    assert ia.spike_rows == ib.spike_rows <= ic.spike_rows and ia.spike_bytes <= ib.spike_bytes <= ic.spike_bytes
    if K >= 32:
        assert ia.spike_bytes < ib.spike_bytes < ic.spike_bytes      # long windows: both levers really apply
    assert "in fp64" in a.view()
    xa, xb, xc = a.apply(f), b.apply(f), c.apply(f)
    assert float((xa - xb).norm() / xb.norm()) <= 1e-15
    assert float((xb - xc).norm() / xc.norm()) <= 1e-15
    u = torch.ones(N, dtype=torch.float64, device="cuda")
    assert float((a.apply(a.matvec(u)) - u).abs().max()) <= 1e-12


@pytest.mark.parametrize("N,K,P,twist", [(2 ** 17, 128, 8, "auto"), (2 ** 17, 128, 8, "off"), (2 ** 16, 100, 3, "auto"), (2 ** 16, 8, 16, "auto"),
                                         (2 ** 16, 3, 5, "off"), (2 ** 16, 32, 8, "auto"), (2 ** 15, 200, 2, "auto")])
def test_one_stage_interface_solves_equal_the_staged_ones(spike, oracle, N, K, P, twist):
    """iface_form = matrix (default): [x_b; x_t] = M [g_b; g_t] with M multiplied out at setup, dealt to 2K/64 workgroups per
    interface (and for the seams, with the inputs staged by the forward launch); = staged: the three dependent mat-vecs.
    Same preconditioner: both against the oracle, coupled and decoupled."""
    band = oracle.gen_band(N, K, delta=1.2)
    f = oracle.gen_vec(N)
    ref = oracle.Spike(band, P)
    for variant, vname in ((1, "coupled"), (0, "decoupled")):
        xo = ref.apply(f, variant)
        xm = _mk(spike, P, vname, twist, iface_form="matrix").setup_band(band).apply(f)
        xs = _mk(spike, P, vname, twist, iface_form="staged").setup_band(band).apply(f)
        assert _rel(xm, xo) <= TOL and _rel(xs, xo) <= TOL and _rel(xm, xs) <= 1e-12


@pytest.mark.parametrize("N,K,P,delta", [(2 ** 17, 64, 8, 0.9), (2 ** 16, 16, 8, 0.8), (2 ** 17, 128, 4, 1.0), (2 ** 16, 40, 6, 0.7)])
def test_weak_dominance_whatever_path_setup_takes(spike, oracle, N, K, P, delta):
    """not diagonally dominant (delta < 1): spikes decay slowly or not at all, so setup may twist, may fall back to ordinary
    chains, may undo the sub-split and may re-solve instead of storing spikes -- whichever it measures, the result is the
    caller's P-partition preconditioner (both variants), and a second setup of the same handle gives the same bits"""
    band = oracle.gen_band(N, K, delta=delta)
    f = oracle.gen_vec(N)
    ref = oracle.Spike(band, P)
    for variant, vname in ((1, "coupled"), (0, "decoupled")):
        sp = _mk(spike, P, vname).setup_band(band)
        x = sp.apply(f)
        assert _rel(x, ref.apply(f, variant)) <= 1e-9, (vname, sp.view())
        assert sp.info().nboost == ref.nboost
        sp.setup_band(band)
        assert np.array_equal(sp.apply(f), x)


@pytest.mark.parametrize("N,K,chains", [(16384, 64, 64), (8192, 128, 16), (32768, 16, 256)])
def test_halves_too_short_for_their_spikes_fall_back(spike, oracle, monkeypatch, N, K, chains):
    """chains barely longer than a few K (forced through the measurement knob SPIKE_AUTO_CHAINS): the spikes do not die inside
    a half, setup measures that and starts over with the same chains untwisted -- the result is the truncated-SPIKE
    preconditioner of that many partitions, as the oracle computes it"""
    monkeypatch.setenv("SPIKE_AUTO_CHAINS", str(chains))
    band = oracle.gen_band(N, K, delta=1.2)
    f = oracle.gen_vec(N)
    sp = _mk(spike, 0).setup_band(band)
    i = sp.info()
    assert "(twisted pairs)" not in sp.view() and i.chains_local == chains == i.P_local
    assert _rel(sp.apply(f), oracle.Spike(band, chains).apply(f, 1)) <= TOL
