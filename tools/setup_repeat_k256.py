"""K = 256, N = 4M: setup three times in one process (fresh handle each time, previous one closed), wall time and the
library's own setup_ms; then the same with a small warm-up setup first (what bench.py does).  Looks for allocation stalls."""
import sys, os, time; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import spike_petsc_amd as S
N, K = 4 * 2 ** 20, int(sys.argv[1]) if len(sys.argv) > 1 else 256
warm = len(sys.argv) > 2
if warm:
    w = S.Spike(partitions=0)
    w.setup_band(S.gen_band_device(1 << 20, K, seed=1, delta=1.2))
    w.apply(torch.ones(1 << 20, dtype=torch.float64, device="cuda"))
    torch.cuda.synchronize(); w.close()
    print("warm-up done, torch reserved %.1f GB" % (torch.cuda.memory_reserved() / 1e9), flush=True)
band = S.gen_band_device(N, K, seed=12345, delta=1.2)
torch.cuda.synchronize()
for rep in range(3):
    sp = S.Spike(partitions=0)
    t0 = time.perf_counter()
    sp.setup_band(band)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("setup %d: wall %.3f s, library setup_ms %.1f, free %.1f GB" % (rep, dt, sp.info().setup_ms, torch.cuda.mem_get_info()[0] / 1e9), flush=True)
    sp.close()
