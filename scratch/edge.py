import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle as O, spike_petsc_amd as S
def rel(a,b): return np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300)
cases=[(64,0,1),(1000,0,3),(64,1,1),(65,3,1),(100,5,1),(129,8,2),(200,40,1),(300,100,1),(1000,128,2),(5000,256,3),(4097,33,4),(777,17,3)]
for N,K,P in cases:
    try:
        band=O.gen_band(N,K,delta=0.9); f=O.gen_vec(N)
        sp=S.Spike(partitions=P).setup_band(band)
        x=sp.apply(f); xo=O.Spike(band,P).apply(f,1)
        i=sp.info()
        print(N,K,P,"rel",rel(x,xo),"chains",i.chains_local,"passes",i.passes,"m",i.spike_rows)
    except Exception as e:
        print(N,K,P,"EXC",e)
