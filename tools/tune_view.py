import sys; sys.path.insert(0, '.')
import torch, spike_petsc_amd as S
N, K = 4 * 2 ** 20, int(sys.argv[1]) if len(sys.argv) > 1 else 128
band = S.gen_band_device(N, K, seed=12345, delta=1.2)
sp = S.Spike(); sp.setup_band(band); print(sp.view()); print("first setup ms", sp.info().setup_ms)
sp.setup_band(band); print("re-setup ms", sp.info().setup_ms)
