import sys, time
sys.path.insert(0, '.')
import torch
import spike_petsc_amd as S
N, K = 4 * 2 ** 20, 128
band = S.gen_band_device(N, K, seed=12345, delta=1.2, row0=0, nrows=N)
import os
for mode in ("on", "off"):
  sp = S.Spike(partitions=0, variant="coupled")
  sp.set_option("workspace_cache", mode)
  print("workspace_cache =", mode, flush=True)
  for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sp.setup_band(band, n_global=N, row0=0)
    torch.cuda.synchronize(); print("  setup %d: wall %.1f ms, setup_ms %.1f, device memory held %.1f GiB" % (i, (time.perf_counter() - t0) * 1e3, sp.info().setup_ms, (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2 ** 30), flush=True)
  sp.close()
