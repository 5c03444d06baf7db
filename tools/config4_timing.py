"""BASELINE config 4 at ASIC_320k scale on the circuit-like stand-in: times MC64, Fiedler, RCM and the nested-reorder KSP
(run on the GPU box: python tools/config4_timing.py).  Numbers quoted in DESIGN.md section 4."""
import sys, os, time; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ctypes as C, torch
import spike_petsc_amd.host as H
from matrices import circuit_like
L=H.lib(); H.chk(L.SpikePetscRegisterAll())
n=321821
t=time.time(); A=circuit_like(n,seed=7,band=24); print("gen %.1fs nnz %d"%(time.time()-t,A.nnz))
t=time.time(); perm,u,v,num=H.mc64_job5(n,A.indptr,A.indices,A.data); print("mc64 job5 %.2fs num %d"%(time.time()-t,num))
B=A[perm]  # row permutation (mat_wbm_rows)
B=B.tocsr(); B.sort_indices()
t=time.time(); o,vec=H.fiedler_order(n,B.indptr,B.indices,B.data,use_device=True); tf=time.time()-t
t=time.time(); oh,vech=H.fiedler_order(n,B.indptr,B.indices,B.data,use_device=False); print("fiedler host-only %.2fs, identical to device-assisted: %s"%(time.time()-t, bool(np.array_equal(o,oh))))
print("fiedler %.2fs profile/bw"%tf, H.profile_bandwidth(n,B.indptr,B.indices), H.profile_bandwidth(n,B.indptr,B.indices,o))
t=time.time(); o2=H.rcm_order(n,B.indptr,B.indices); print("rcm %.2fs"%(time.time()-t), H.profile_bandwidth(n,B.indptr,B.indices,o2))
for second in ("fiedler","rcm"):
    H.options(ksp_type="reorder", mat_ordering_type="wbm", mat_wbm_rows=1, reorder_ksp_type="reorder", reorder_mat_ordering_type=second,
              reorder_reorder_ksp_type="gmres", reorder_reorder_ksp_rtol=1e-5, reorder_reorder_ksp_max_it=500,
              reorder_reorder_pc_type="banded", reorder_reorder_pc_banded_kmax=50)
    M=H.Mat.from_scipy(A); uvec,b,x=H.Vec(n),H.Vec(n),H.Vec(n)
    H.chk(L.VecSet(uvec.h,1.0)); H.chk(L.MatMult(M.h,uvec.h,b.h))
    ksp=C.c_void_p(); H.chk(L.KSPCreate(C.byref(ksp))); H.chk(L.KSPSetOperators(ksp,M.h,M.h)); H.chk(L.KSPSetFromOptions(ksp))
    t=time.time(); H.chk(L.KSPSetUp(ksp)); ts=time.time()-t
    t=time.time(); H.chk(L.KSPSolve(ksp,b.h,x.h)); tsol=time.time()-t
    its=C.c_int64(); H.chk(L.KSPGetIterationNumber(ksp,C.byref(its)))
    print(second,"setup %.2fs solve %.3fs its %d err %.3e"%(ts,tsol,its.value,np.abs(x.array-1).max()))
