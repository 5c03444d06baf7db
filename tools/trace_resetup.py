import sys, os
sys.path.insert(0, '.')
import torch
import spike_petsc_amd as S
N, K = 4 * 2 ** 20, int(sys.argv[1]) if len(sys.argv) > 1 else 128
band = S.gen_band_device(N, K, seed=12345, delta=1.2, row0=0, nrows=N)
sp = S.Spike(partitions=0, variant="coupled")
sp.setup_band(band, n_global=N, row0=0)
torch.cuda.synchronize()
os.environ["SPIKE_SETUP_TRACE"] = "1"
sp.setup_band(band, n_global=N, row0=0)
torch.cuda.synchronize()
print("setup_ms", sp.info().setup_ms)
