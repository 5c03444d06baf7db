"""ctypes front-end of libspike_petsc_host.so -- the C host mirror of the reference's PETSc plugin surface
(include/spike_petsc_host.h).  Test/bench tooling: the functions keep the reference's names so that a test
reads like /root/reference/src/testbed2.c."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__)) if "__file__" in globals() else os.getcwd()
# SPIKE_HOST_LIB selects another build of the same sources (the address/UB-sanitizer build, tools/run_asan_tests.sh)
HOST_LIB_PATH = os.environ.get("SPIKE_HOST_LIB") or os.path.join(_HERE, "libspike_petsc_host.so")

i64 = C.c_int64
i64p = C.POINTER(C.c_int64)
dp = C.POINTER(C.c_double)
vp = C.c_void_p

HOST_SYMBOLS = [
    "SpikeHostLastError", "SpikePetscRegisterAll", "PetscOptionsSetValue", "PetscOptionsClearValue", "PetscOptionsClear",
    "MatCreateSeqAIJWithArrays", "MatDestroy", "MatGetSize", "MatSeqAIJGetCSR", "MatMult", "MatPermute",
    "MatComputeBandwidth", "MatCreateSubMatrixBanded", "VecCreateSeq", "VecDestroy", "VecGetArray", "VecGetSize",
    "VecSet", "VecCopy", "VecAXPY", "VecNorm2", "VecPermute", "ISCreateGeneral", "ISCreateStride", "ISDestroy",
    "ISGetIndices", "MatOrderingRegister", "MatGetOrdering", "MatGetOrdering_WBM", "MatGetOrdering_AWBM", "MatGetOrdering_Fiedler",
    "MatGetOrdering_Natural", "MatGetOrdering_RCM", "spike_rcm_order", "PCRegister", "PCCreate", "PCSetType", "PCSetOptionsPrefix", "PCAppendOptionsPrefix",
    "PCSetOperators", "PCSetFromOptions", "PCSetUp", "PCApply", "PCReset", "PCDestroy", "PCView", "PCGetDiagonalScale",
    "PCCreate_Banded", "PCCreate_Spike", "PCCreate_None", "PCBandedSetMaxHalfBandwidth", "PCBandedSetNormFraction",
    "PCBandedGetInfo", "PCGetSpikeHandle", "KSPRegister", "KSPCreate", "KSPSetType", "KSPSetOptionsPrefix",
    "KSPAppendOptionsPrefix", "KSPSetOperators", "KSPGetPC", "KSPSetTolerances", "KSPSetFromOptions", "KSPSetUp",
    "KSPSolve", "KSPGetConvergedReason", "KSPGetIterationNumber", "KSPGetResidualNorm", "KSPView", "KSPDestroy",
    "KSPCreate_Reorder", "KSPCreate_GMRES", "KSPReorderGetOrdering", "spike_mc64_job5", "spike_fiedler_order", "spike_fiedler_order_ex", "spike_fiedler_halves_order", "MatGetOrdering_FiedlerHalves",
    "spike_profile_bandwidth", "spike_awbm", "spike_awbm_dist_rowmin", "spike_awbm_dist_match", "spike_mc64_job5_i32", "spike_awbm_i32", "spike_fiedler_order_i32", "spike_rcm_order_i32", "MatLoad", "MatLoadMatrixMarket", "MatViewMatrixMarket", "MatViewBinary",
]

_L = None


class HostError(RuntimeError):
    pass


def lib():
    global _L
    if _L is not None:
        return _L
    import spike_petsc_amd as S
    S.lib()  # engine first (and PyTorch's HIP runtime before it)
    if not os.path.exists(HOST_LIB_PATH):
        raise HostError("libspike_petsc_host.so is missing: build with `make -C spike-petsc_amd/csrc`")
    L = C.CDLL(HOST_LIB_PATH)
    L.SpikeHostLastError.restype = C.c_char_p
    L.PetscOptionsSetValue.argtypes = [C.c_char_p, C.c_char_p]
    L.PetscOptionsClearValue.argtypes = [C.c_char_p]
    L.MatCreateSeqAIJWithArrays.argtypes = [i64, i64p, i64p, dp, C.POINTER(vp)]
    L.MatDestroy.argtypes = [C.POINTER(vp)]
    L.MatSeqAIJGetCSR.argtypes = [vp, i64p, C.POINTER(i64p), C.POINTER(i64p), C.POINTER(dp)]
    L.MatMult.argtypes = [vp, vp, vp]
    L.MatPermute.argtypes = [vp, vp, vp, C.POINTER(vp)]
    L.MatComputeBandwidth.argtypes = [vp, C.c_double, i64p]
    L.MatCreateSubMatrixBanded.argtypes = [vp, i64p, dp, C.POINTER(vp)]
    L.VecCreateSeq.argtypes = [i64, C.POINTER(vp)]
    L.VecDestroy.argtypes = [C.POINTER(vp)]
    L.VecGetArray.argtypes = [vp, C.POINTER(dp)]
    L.VecSet.argtypes = [vp, C.c_double]
    L.VecCopy.argtypes = [vp, vp]
    L.VecAXPY.argtypes = [vp, C.c_double, vp]
    L.VecNorm2.argtypes = [vp, dp]
    L.VecPermute.argtypes = [vp, vp, C.c_int]
    L.ISCreateGeneral.argtypes = [i64, i64p, C.POINTER(vp)]
    L.ISDestroy.argtypes = [C.POINTER(vp)]
    L.ISGetIndices.argtypes = [vp, i64p, C.POINTER(i64p)]
    L.MatGetOrdering.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(vp)]
    L.PCCreate.argtypes = [C.POINTER(vp)]
    L.PCSetType.argtypes = [vp, C.c_char_p]
    L.PCSetOptionsPrefix.argtypes = [vp, C.c_char_p]
    L.PCSetOperators.argtypes = [vp, vp, vp]
    L.PCSetFromOptions.argtypes = [vp]
    L.PCSetUp.argtypes = [vp]
    L.PCApply.argtypes = [vp, vp, vp]
    L.PCReset.argtypes = [vp]
    L.PCDestroy.argtypes = [C.POINTER(vp)]
    L.PCBandedSetMaxHalfBandwidth.argtypes = [vp, i64]
    L.PCBandedSetNormFraction.argtypes = [vp, C.c_double]
    L.PCBandedGetInfo.argtypes = [vp, i64p, dp, i64p, dp]
    L.KSPCreate.argtypes = [C.POINTER(vp)]
    L.KSPSetType.argtypes = [vp, C.c_char_p]
    L.KSPSetOptionsPrefix.argtypes = [vp, C.c_char_p]
    L.KSPSetOperators.argtypes = [vp, vp, vp]
    L.KSPGetPC.argtypes = [vp, C.POINTER(vp)]
    L.KSPSetTolerances.argtypes = [vp, C.c_double, i64]
    L.KSPSetFromOptions.argtypes = [vp]
    L.KSPSetUp.argtypes = [vp]
    L.KSPSolve.argtypes = [vp, vp, vp]
    L.KSPGetConvergedReason.argtypes = [vp, C.POINTER(C.c_int)]
    L.KSPGetIterationNumber.argtypes = [vp, i64p]
    L.KSPGetResidualNorm.argtypes = [vp, dp]
    L.KSPDestroy.argtypes = [C.POINTER(vp)]
    L.KSPReorderGetOrdering.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.spike_mc64_job5.argtypes = [i64, i64p, i64p, dp, i64p, dp, dp, i64p]
    L.spike_awbm.argtypes = [i64, i64p, i64p, dp, i64p, dp, dp]
    L.spike_awbm_dist_rowmin.argtypes = [i64, i64, i64p, i64p, dp, dp]
    L.spike_awbm_dist_match.argtypes = [i64, i64, i64, i64p, i64p, dp, dp, i64p, dp, dp]
    L.MatLoad.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.MatLoadMatrixMarket.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.MatViewMatrixMarket.argtypes = [vp, C.c_char_p]
    L.MatViewBinary.argtypes = [vp, C.c_char_p]
    L.spike_rcm_order.argtypes = [i64, i64p, i64p, i64p]
    L.spike_fiedler_order.argtypes = [i64, i64p, i64p, dp, i64p, dp]
    L.spike_fiedler_order_ex.argtypes = [i64, i64p, i64p, dp, i64p, dp, C.c_int]
    L.spike_fiedler_halves_order.argtypes = [i64, i64p, i64p, dp, i64p, i64p, i64p, C.c_int]
    L.spike_profile_bandwidth.argtypes = [i64, i64p, i64p, i64p, i64p, i64p]
    _L = L
    return L


def chk(rc):
    if rc:
        raise HostError("PetscErrorCode %d: %s" % (rc, lib().SpikeHostLastError().decode()))


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def options(**kw):
    """PetscOptionsClear + PetscOptionsSetValue('-key', value) for every keyword (underscores kept)."""
    L = lib()
    L.PetscOptionsClear()
    for k, v in kw.items():
        chk(L.PetscOptionsSetValue(("-" + k).encode(), str(v).encode()))


class Mat:
    def __init__(self, n=None, ia=None, ja=None, a=None, handle=None):
        self.h = vp()
        if handle is not None:
            self.h = handle
        else:
            ia, ja, a = _i(ia), _i(ja), _d(a)
            chk(lib().MatCreateSeqAIJWithArrays(n, ia.ctypes.data_as(i64p), ja.ctypes.data_as(i64p), a.ctypes.data_as(dp),
                                                C.byref(self.h)))

    @classmethod
    def from_scipy(cls, A):
        A = A.tocsr()
        A.sort_indices()
        return cls(A.shape[0], A.indptr, A.indices, A.data)

    def csr(self):
        n = i64(0)
        ia, ja, a = i64p(), i64p(), dp()
        chk(lib().MatSeqAIJGetCSR(self.h, C.byref(n), C.byref(ia), C.byref(ja), C.byref(a)))
        n = n.value
        IA = np.ctypeslib.as_array(ia, (n + 1,)).copy()
        nnz = int(IA[n])
        return n, IA, np.ctypeslib.as_array(ja, (max(nnz, 1),))[:nnz].copy(), np.ctypeslib.as_array(a, (max(nnz, 1),))[:nnz].copy()

    def to_scipy(self):
        import scipy.sparse as sp
        n, ia, ja, a = self.csr()
        return sp.csr_matrix((a, ja, ia), shape=(n, n))

    def destroy(self):
        lib().MatDestroy(C.byref(self.h))


class Vec:
    def __init__(self, n=None, values=None):
        self.h = vp()
        if values is not None:
            n = len(values)
        chk(lib().VecCreateSeq(n, C.byref(self.h)))
        self.n = n
        if values is not None:
            self.array[:] = values

    @property
    def array(self):
        p = dp()
        chk(lib().VecGetArray(self.h, C.byref(p)))
        return np.ctypeslib.as_array(p, (self.n,))

    def destroy(self):
        lib().VecDestroy(C.byref(self.h))


def is_indices(h):
    n = i64(0)
    p = i64p()
    chk(lib().ISGetIndices(h, C.byref(n), C.byref(p)))
    return np.ctypeslib.as_array(p, (n.value,)).copy()


def mc64_job5(n, colptr, rowind, val):
    colptr, rowind, val = _i(colptr), _i(rowind), _d(val)
    perm = np.zeros(n, dtype=np.int64)
    u = np.zeros(n)
    v = np.zeros(n)
    num = i64(0)
    rc = lib().spike_mc64_job5(n, colptr.ctypes.data_as(i64p), rowind.ctypes.data_as(i64p), val.ctypes.data_as(dp),
                               perm.ctypes.data_as(i64p), u.ctypes.data_as(dp), v.ctypes.data_as(dp), C.byref(num))
    if rc:
        raise HostError("spike_mc64_job5 failed")
    return perm, u, v, num.value


def awbm(n, ia, ja, a):
    ia, ja, a = _i(ia), _i(ja), _d(a)
    perm = np.zeros(n, dtype=np.int64)
    rc = lib().spike_awbm(n, ia.ctypes.data_as(i64p), ja.ctypes.data_as(i64p), a.ctypes.data_as(dp),
                          perm.ctypes.data_as(i64p), None, None)
    if rc:
        raise HostError("spike_awbm failed (%d)" % rc)
    return perm


def awbm_dist(row0, N, ia, ja, a, group=None, scalings=False):
    """The approximate matching of a matrix distributed by rows (reference: MatComputeMatching_MPIAIJ, src/wbm.c:201-440):
    this rank owns rows [row0, row0 + n_local) in CSR with GLOBAL column indices.  Collective over `group` (torch.distributed;
    None with no initialised process group = one rank): the per-column minima are reduced with MIN -- the operation the
    reference's comment at :270 asks for.  Returns the rank's local permutation (perm[match[c]] = c) and the reduced u."""
    ia, ja, a = _i(ia), _i(ja), _d(a)
    n_local = len(ia) - 1
    u = np.empty(N, dtype=np.float64)
    rc = lib().spike_awbm_dist_rowmin(n_local, N, ia.ctypes.data_as(i64p), ja.ctypes.data_as(i64p), a.ctypes.data_as(dp), u.ctypes.data_as(dp))
    if rc:
        raise HostError("spike_awbm_dist_rowmin failed (%d)" % rc)
    try:
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized()
    except ImportError:
        multi = False
    if multi:
        import torch
        t = torch.from_numpy(u)
        if dist.get_backend(group) == "nccl":        # RCCL reduces device tensors
            t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            u = t.cpu().numpy()
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    perm = np.zeros(n_local, dtype=np.int64)
    sr = np.zeros(n_local) if scalings else None
    sc = np.zeros(n_local) if scalings else None
    rc = lib().spike_awbm_dist_match(n_local, row0, N, ia.ctypes.data_as(i64p), ja.ctypes.data_as(i64p), a.ctypes.data_as(dp),
                                     u.ctypes.data_as(dp), perm.ctypes.data_as(i64p),
                                     sr.ctypes.data_as(dp) if scalings else None, sc.ctypes.data_as(dp) if scalings else None)
    if rc:
        raise HostError("spike_awbm_dist_match failed (%d)" % rc)
    return (perm, u, sr, sc) if scalings else (perm, u)


def rcm_order(n, ia, ja):
    ia, ja = _i(ia), _i(ja)
    order = np.zeros(n, dtype=np.int64)
    if lib().spike_rcm_order(n, ia.ctypes.data_as(i64p), ja.ctypes.data_as(i64p), order.ctypes.data_as(i64p)):
        raise HostError("spike_rcm_order failed")
    return order


def fiedler_order(n, ia, ja, a, use_device=False):
    """use_device: the LOBPCG refinement of the large levels on the GPU (bit-identical permutation)"""
    ia, ja, a = _i(ia), _i(ja), _d(a)
    order = np.zeros(n, dtype=np.int64)
    vec = np.zeros(n)
    if lib().spike_fiedler_order_ex(n, ia.ctypes.data_as(i64p), ja.ctypes.data_as(i64p), a.ctypes.data_as(dp),
                                    order.ctypes.data_as(i64p), vec.ctypes.data_as(dp), int(bool(use_device))):
        raise HostError("spike_fiedler_order failed")
    return order, vec


def fiedler_halves_order(n, ia, ja, a, use_device=False):
    """Fiedler cut + RCM on each half (src/spectralPartition.c:326-417): (order, positive-half size, [bw pos before, after, neg before, after])"""
    ia, ja, a = _i(ia), _i(ja), _d(a)
    order = np.zeros(n, dtype=np.int64)
    bw = np.zeros(4, dtype=np.int64)
    npos = i64(0)
    if lib().spike_fiedler_halves_order(n, ia.ctypes.data_as(i64p), ja.ctypes.data_as(i64p), a.ctypes.data_as(dp),
                                        order.ctypes.data_as(i64p), C.byref(npos), bw.ctypes.data_as(i64p), int(bool(use_device))):
        raise HostError("spike_fiedler_halves_order failed")
    return order, npos.value, bw


def profile_bandwidth(n, ia, ja, order=None):
    ia, ja = _i(ia), _i(ja)
    p, b = i64(0), i64(0)
    o = None if order is None else _i(order)
    lib().spike_profile_bandwidth(n, ia.ctypes.data_as(i64p), ja.ctypes.data_as(i64p),
                                  None if o is None else o.ctypes.data_as(i64p), C.byref(p), C.byref(b))
    return p.value, b.value
