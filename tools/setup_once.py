"""one setup at the headline size (for rocprofv3 counter passes over the setup kernels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import spike_petsc_amd as S
K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = 4 * 2 ** 20
band = S.gen_band_device(N, K)
sp = S.Spike().setup_band(band)
torch.cuda.synchronize()
print(sp.view())
