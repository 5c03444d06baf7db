"""RCCL call sites on ONE GPU: SPIKE_RCCL_SELFTEST=1 makes spike_comm_init(nranks=1) create a real one-rank RCCL
communicator and routes every exchange step (setup all-reduce / all-gather, per-apply all-gather, mat-vec halo, dot
product all-reduce) through librccl.  This checks symbol loading, enum values and call signatures against the RCCL the
process really has; the multi-rank ALGORITHM is covered by tests/test_multirank_gpu.py (loopback transport)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_every_rccl_call_site_runs_with_one_rank(spike, oracle):
    import torch
    os.environ["SPIKE_RCCL_SELFTEST"] = "1"
    try:
        N, K, P = 32768, 48, 8
        band = oracle.gen_band(N, K, delta=0.8)
        f = oracle.gen_vec(N)
        sp = spike.Spike(partitions=P)
        sp.comm_init(1, 0, spike.unique_id())           # ncclGetUniqueId + ncclCommInitRank
        sp.setup_band(band)                             # ncclAllReduce(max) + ncclAllGather(tips)
        x = sp.apply(torch.from_numpy(f).cuda())        # ncclAllGather(2K)
        torch.cuda.synchronize()
        xo = oracle.Spike(band, P).apply(f, 1)
        assert np.linalg.norm(x.cpu().numpy() - xo) <= 1e-10 * np.linalg.norm(xo)
        u = np.ones(N)
        b = sp.matvec(torch.from_numpy(u).cuda())       # halo all-gather
        assert np.linalg.norm(b.cpu().numpy() - oracle.band_matvec(band, u)) <= 1e-13 * np.sqrt(N)
        xg = torch.zeros(N, dtype=torch.float64, device="cuda")
        it, rn, ms, ok = sp.gmres(b, xg, restart=30, rtol=1e-8, maxit=100)   # ncclAllReduce(sum) per Gram-Schmidt pass
        assert ok and float((xg - 1).abs().max()) <= 1e-6
        sp.close()                                      # ncclCommDestroy
    finally:
        os.environ.pop("SPIKE_RCCL_SELFTEST", None)
