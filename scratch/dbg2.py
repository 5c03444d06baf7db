import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle as O, spike_petsc_amd as S
for (N,K,P) in [(512,8,1),(2048,8,2),(4096,32,2)]:
    band=O.gen_band(N,K); f=O.gen_vec(N)
    sp=S.Spike(partitions=P,variant="decoupled").setup_band(band)
    x=sp.apply(f); xo=O.Spike(band,P).apply(f,0)
    err=np.abs(x-xo)
    bad=np.nonzero(err>1e-9*np.abs(xo).max())[0]
    print(N,K,P,"nbad",len(bad), bad[:20], bad[-5:] if len(bad) else "")
