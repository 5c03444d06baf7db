"""How far may the stored spikes be cut?  For each drop level (option spike_tol, relative to the spikes' peak) at the
strong-scaling per-rank size and at the headline size: rows kept, ms per PCApply, and the damage -- max |M^-1 (A 1) - 1| and the
relative 2-norm distance of M^-1 f (random f) from the result with the 1e-16 level.  Parity is judged at 1e-10 (relative
2-norm against the oracle): the default must stay orders of magnitude inside that.  Run on the GPU box."""
import sys, os, time; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import spike_petsc_amd as S
import oracle as O
for N, K, delta in ((524288, 128, 1.2), (4 * 2 ** 20, 128, 1.2), (2 ** 20, 32, 1.2), (2 ** 21, 8, 1.2)):
    band = S.gen_band_device(N, K, seed=12345, delta=delta)
    f = torch.from_numpy(O.gen_vec(N)).cuda()
    ref = None
    for tol in (1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10):
        sp = S.Spike(partitions=0)
        sp.set_option("spike_tol", repr(tol))
        sp.setup_band(band)
        i = sp.info()
        u = torch.ones(N, dtype=torch.float64, device="cuda")
        b = sp.matvec(u)
        x = torch.empty_like(b)
        for _ in range(5): sp.apply(b, x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): sp.apply(b, x)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
        e1 = float((x - u).abs().max())
        y = sp.apply(f)
        if ref is None: ref = y.clone()
        d = float((y - ref).norm() / ref.norm())
        print("N=%-8d K=%-3d tol=%.0e  m=%-5d chains=%-4d ms=%.4f  max|M^-1 A1 - 1|=%.2e  rel diff to 1e-16: %.2e" % (N, K, tol, i.spike_rows, i.chains_local, ms, e1, d), flush=True)
        sp.close()
    del band
