"""One parity test per BASELINE.json config (the GPU ones), at the sizes BASELINE.json names.

config 1 (tridiagonal N=16384, 4 partitions, CPU plumbing) lives in tests/test_oracle.py / test_spike_gpu.py CASES.
config 4 (MC64 + Fiedler -> band -> PCSPIKE in GMRES) is tests/test_host_gpu.py::test_config4_pipeline_*.
config 5 (8 GPUs, 8 partitions/GPU) runs here with 8 THREAD ranks on one GPU through the loopback transport (same
algorithm, buffers and call order as the RCCL path): once at 2^14 rows per rank against the oracle, and once at the
REAL per-GPU size (N = 2^24, 2^21 rows per rank, ~100 GB of the 288 GB of HBM) through size-independent properties."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_config2_banded_1M_k32_64_partitions(spike, oracle):
    """Banded N=1M half-bw=32 fp64, 64 partitions: full size against the oracle (it finishes in seconds)."""
    import torch
    N, K, P = 2 ** 20, 32, 64
    band = oracle.gen_band(N, K, seed=12345, delta=1.2)
    f = oracle.gen_vec(N)
    ref = oracle.Spike(band, P)
    sp = spike.Spike(partitions=P, variant="coupled").setup_band(torch.from_numpy(band).cuda())
    x = sp.apply(torch.from_numpy(f).cuda()).cpu().numpy()
    assert _rel(x, ref.apply(f, 1)) <= 1e-10
    i = sp.info()
    assert i.P_local == 64 and i.Kp == 32 and i.rows_per_block == 32
    assert i.chains_local > 64 and i.chains_local % 64 == 0      # 64 partitions swept as many chains (verified cut)
    sp_off = spike.Spike(partitions=P, variant="coupled")
    sp_off.set_option("subsplit", "off")
    sp_off.setup_band(torch.from_numpy(band).cuda())
    assert sp_off.info().chains_local == 64
    assert _rel(sp_off.apply(torch.from_numpy(f).cuda()).cpu().numpy(), ref.apply(f, 1)) <= 1e-10
    sp.set_option("variant", "decoupled")
    xd = sp.apply(torch.from_numpy(f).cuda()).cpu().numpy()
    assert _rel(xd, ref.apply(f, 0)) <= 1e-10
    # exact-solution round trip, the reference's acceptance check (src/testbed2.c:120-132)
    sp.set_option("variant", "coupled")
    u = np.ones(N)
    b = oracle.band_matvec(band, u)
    assert np.abs(sp.apply(b) - u).max() <= 1e-10


def test_config3_banded_4M_k256(spike, oracle):
    """Banded N=4M half-bw=256 (the MFMA block-LU path, in-place window): size-independent properties."""
    import torch
    N, K = 4 * 2 ** 20, 256
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
    assert float((sp.matvec(xv) - bv).norm() / bv.norm()) <= 1e-12       # residual of the preconditioned solve
    assert float((sp.apply(b + 3.0 * bv) - (x + 3.0 * xv)).abs().max()) <= 1e-9   # linearity
    i = sp.info()
    assert i.Kp == 256 and i.waves_per_chain == 8 and i.nboost == 0
    # a slice of the same system small enough for the oracle: first 2^17 rows as their own system
    n = 2 ** 17
    bs = oracle.gen_band(n, K, seed=12345, delta=1.2)
    f = oracle.gen_vec(n)
    sps = spike.Spike(partitions=16).setup_band(bs)
    assert _rel(sps.apply(f), oracle.Spike(bs, 16).apply(f, 1)) <= 1e-10


def test_config5_eight_ranks_eight_partitions_each(spike, oracle):
    """Banded half-bw=128, 8 partitions per GPU, 8 ranks with the tips exchanged by all-gather."""
    import torch
    G, Pl, K = 8, 8, 128
    n_rank = 2 ** 14
    N = G * n_rank
    band = oracle.gen_band(N, K, seed=12345, delta=1.2)
    f = oracle.gen_vec(N)
    out, err = [None] * G, [None] * G

    def work(r):
        try:
            sp = spike.Spike(partitions=Pl)
            sp.comm_init_local(G, r, 555)
            r0 = r * n_rank
            sp.setup_band(np.ascontiguousarray(band[:, r0:r0 + n_rank]), n_global=N, row0=r0)
            out[r] = sp.apply(torch.from_numpy(f[r0:r0 + n_rank].copy()).cuda()).cpu().numpy()
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    for e in err:
        if e is not None:
            raise e
    x = np.concatenate(out)
    assert _rel(x, oracle.Spike(band, G * Pl).apply(f, 1)) <= 1e-10


def test_config5_full_size_eight_ranks_on_one_gpu(spike, oracle):
    """BASELINE config 5 at its real size: N = 2^24, half-bw 128, 8 ranks x 8 partitions, every rank's band generated on
    the device (bit-identical to the oracle's generator).  Size-independent properties, every one over ALL ranks:
    M^{-1}(A 1) = 1, residual of a random solve, linearity -- the tips really cross the rank boundaries (a rank that
    ignored its neighbours would miss 1 by the coupling term)."""
    import torch
    G, Pl, K = 8, 8, 128
    N = 2 ** 24
    n_rank = N // G
    v_host = oracle.gen_vec(N)
    res, err = [None] * G, [None] * G

    def work(r):
        try:
            r0 = r * n_rank
            band = spike.gen_band_device(N, K, seed=12345, delta=1.2, row0=r0, nrows=n_rank)
            sp = spike.Spike(partitions=Pl)
            sp.comm_init_local(G, r, 777)
            sp.setup_band(band, n_global=N, row0=r0)
            del band
            i = sp.info()
            u = torch.ones(n_rank, dtype=torch.float64, device="cuda")
            b = sp.matvec(u)                      # collective: halo rows come from the neighbouring ranks
            x = sp.apply(b)                       # collective: tips all-gathered
            v = torch.from_numpy(v_host[r0:r0 + n_rank].copy()).cuda()
            bv = sp.matvec(v)
            xv = sp.apply(bv)
            rv = sp.matvec(xv) - bv
            lin = sp.apply(b + 3.0 * bv) - (x + 3.0 * xv)
            torch.cuda.synchronize()
            res[r] = dict(err1=float((x - u).abs().max()), r2=float((rv * rv).sum()), b2=float((bv * bv).sum()),
                          lin=float(lin.abs().max()), errv=float((xv - v).abs().max()), P=i.P_local, chains=i.chains_local,
                          nranks=i.nranks, Pg=i.P_global, nboost=i.nboost, passes=i.passes, n=i.n_local)
            sp.close()
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join(timeout=900) for t in th]
    for e in err:
        if e is not None:
            raise e
    assert all(q is not None for q in res)
    assert max(q["err1"] for q in res) <= 1e-10
    assert max(q["errv"] for q in res) <= 1e-10
    assert np.sqrt(sum(q["r2"] for q in res) / sum(q["b2"] for q in res)) <= 1e-12
    assert max(q["lin"] for q in res) <= 1e-9
    for q in res:
        assert q["P"] == Pl and q["Pg"] == G * Pl and q["nranks"] == G and q["n"] == n_rank and q["nboost"] == 0
        assert q["chains"] >= Pl and q["chains"] % Pl == 0


@pytest.mark.parametrize("P", [8, 64])
def test_headline_size_with_the_survey_partition_counts(spike, oracle, P):
    """SURVEY.md 8d names P = 8 G and 64 G for the headline (N = 4 2^20, K = 128).  A caller-chosen P is honoured -- the
    preconditioner is P-defined -- but swept as many chains (verified cuts, twisted pairs).  Full size: size-independent
    properties; the P-partition preconditioner itself is pinned on a slice the oracle can factor (same rows per partition)."""
    import torch
    N, K = 4 * 2 ** 20, 128
    band = spike.gen_band_device(N, K, seed=12345, delta=1.2)
    sp = spike.Spike(partitions=P, variant="coupled").setup_band(band)
    i = sp.info()
    assert i.P_local == P and i.chains_local % P == 0 and i.chains_local >= 128 and i.passes == 1 and i.nboost == 0
    u = torch.ones(N, dtype=torch.float64, device="cuda")
    b = sp.matvec(u)
    x = sp.apply(b)
    assert float((x - u).abs().max()) <= 1e-10
    v = torch.from_numpy(oracle.gen_vec(N)).cuda()
    bv = sp.matvec(v)
    xv = sp.apply(bv)
    assert float((sp.matvec(xv) - bv).norm() / bv.norm()) <= 1e-12
    assert float((sp.apply(b + 3.0 * bv) - (x + 3.0 * xv)).abs().max()) <= 1e-9
    del band
    # the oracle's partitioned preconditioner on a slice it factors in seconds (2 partitions)
    n = 2 ** 15 if P == 64 else 2 ** 16
    bs = oracle.gen_band(n, K, seed=12345, delta=1.2)
    f = oracle.gen_vec(n)
    sps = spike.Spike(partitions=2).setup_band(bs)
    assert _rel(sps.apply(f), oracle.Spike(bs, 2).apply(f, 1)) <= 1e-10
