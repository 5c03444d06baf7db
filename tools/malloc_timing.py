import time, ctypes as C, sys
sys.path.insert(0,'.')
import torch
import spike_petsc_amd as S
L=S.lib()
torch.cuda.init(); torch.zeros(1,device='cuda'); torch.cuda.synchronize()
def t(sz,reps=3):
    out=[]
    for _ in range(reps):
        p=C.c_void_p()
        t0=time.perf_counter(); L.spike_dev_malloc(C.byref(p), C.c_size_t(sz)); t1=time.perf_counter(); L.spike_dev_free(p); t2=time.perf_counter()
        out.append(((t1-t0)*1e3,(t2-t1)*1e3))
    return out
for sz in [1<<12, 1<<20, 1<<26, 1<<30, 8<<30]:
    print(sz, ["%.3f/%.3f ms"%x for x in t(sz)])
