"""Repeated setup / apply / gmres / reset on one handle and across handles: device memory must return to its level."""
import sys
sys.path.insert(0, '.')
import torch
import spike_petsc_amd as S

N, K = 2 ** 20, 64
band = S.gen_band_device(N, K, seed=1, delta=1.2, row0=0, nrows=N)
b = torch.ones(N, dtype=torch.float64, device='cuda')
x = torch.zeros_like(b)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
for rep in range(6):
    sp = S.Spike(partitions=0, variant="coupled")
    for i in range(3):
        sp.setup_band(band, n_global=N, row0=0)
        sp.apply(b, x)
        sp.gmres(b, x, restart=30, rtol=1e-8, maxit=40)
    del sp
    torch.cuda.synchronize()
    print("after handle %d: free delta %.1f MiB" % (rep, (free0 - torch.cuda.mem_get_info()[0]) / 2 ** 20), flush=True)
