import sys, time
sys.path.insert(0, '.')
import torch
import spike_petsc_amd as S
N, K = 4 * 2 ** 20, 128
band = S.gen_band_device(N, K, seed=12345, delta=1.2, row0=0, nrows=N)
sp = S.Spike(partitions=0, variant="coupled")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sp.setup_band(band, n_global=N, row0=0)
    torch.cuda.synchronize(); print("setup %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
