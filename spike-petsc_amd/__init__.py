"""spike-petsc_amd: ctypes front-end of libspike_mi355.so (the C-ABI of include/spike_mi355.h).

This Python layer is tooling for tests and bench.py only -- the product is the shared
library (HIP kernels + C-ABI) and the C host mirror of the reference's plugin surface.
There is no fallback: if the library is missing, or no HIP device is present, the calls fail.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__)) if "__file__" in globals() else os.getcwd()
if os.path.basename(_HERE) == "spike_petsc_amd":  # executed through the import shim
    _HERE = os.path.join(os.path.dirname(_HERE), "spike-petsc_amd")
LIB_PATH = os.environ.get("SPIKE_MI355_LIB") or os.path.join(_HERE, "libspike_mi355.so")   # (override: A/B of two builds)

i64 = C.c_int64
dptr = C.POINTER(C.c_double)
iptr = C.POINTER(C.c_int64)

VARIANT_DECOUPLED = 0
VARIANT_COUPLED = 1
UNIQUE_ID_BYTES = 128

# every symbol include/spike_mi355.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "spike_create", "spike_destroy", "spike_reset", "spike_last_error", "spike_set_option", "spike_set_stream",
    "spike_comm_unique_id", "spike_comm_init", "spike_comm_init_local", "spike_setup_band", "spike_setup_csr", "spike_apply", "spike_gmres",
    "spike_band_matvec", "spike_gen_band", "spike_get_info", "spike_view", "spike_get_tips", "spike_last_sweep_ms",
    "spike_set_operator_csr", "spike_clear_operator", "spike_dev_malloc", "spike_dev_free", "spike_dev_upload",
    "spike_dev_download", "spike_csr_band_k", "spike_csr_to_band", "spike_measure_read_bw",
    "spike_set_operator_band", "spike_operator_matvec", "spike_auto_partitions",
    "spike_setup_csr_dist", "spike_csr_band_weights", "spike_band_rule",
    "spike_setup_csr32", "spike_setup_csr_dist32", "spike_csr_band_k32", "spike_csr_band_weights32",
    "spike_permute_csr", "spike_permute_vec", "spike_awbm_device",
    "spike_device_count", "spike_fd_create", "spike_fd_destroy", "spike_fd_dots", "spike_fd_lap", "spike_fd_shift",
    "spike_fd_div", "spike_fd_fill_alternating", "spike_fd_download_x", "spike_fd_refine",
]


class SpikeInfo(C.Structure):
    _fields_ = [
        ("n_local", i64), ("n_global", i64), ("row0", i64), ("K", C.c_int32), ("Kp", C.c_int32),
        ("P_local", C.c_int32), ("P_global", C.c_int32), ("variant", C.c_int32), ("rows_per_block", C.c_int32),
        ("waves_per_chain", C.c_int32), ("nranks", C.c_int32), ("rank", C.c_int32), ("nboost", i64),
        ("factor_bytes", i64), ("iface_bytes", i64), ("setup_ms", C.c_double), ("k_extracted", C.c_int32),
        ("frac_extracted", C.c_double), ("passes", C.c_int32), ("spike_rows", C.c_int32), ("spike_bytes", i64),
        ("chains_local", C.c_int32), ("twisted", C.c_int32), ("spike_rows_fp64", C.c_int32), ("seams_local", C.c_int32),
    ]


class SpikeError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libspike_mi355.so for gfx950 with hipcc (works without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


_LIB = None


def lib():
    """Load the C-ABI library; fails loudly when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise SpikeError("libspike_mi355.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                         % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME).  A process must hold ONE HIP runtime, so when
    # torch is installed it is imported first and this library then binds to the runtime torch loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.spike_create.argtypes = [C.POINTER(vp)]
    L.spike_auto_partitions.argtypes = [C.c_int, i64]
    L.spike_destroy.argtypes = [vp]
    L.spike_reset.argtypes = [vp]
    L.spike_last_error.argtypes = [vp]
    L.spike_last_error.restype = C.c_char_p
    L.spike_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.spike_set_stream.argtypes = [vp, vp]
    L.spike_comm_unique_id.argtypes = [C.c_char_p]
    L.spike_comm_init.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.spike_comm_init_local.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.spike_setup_band.argtypes = [vp, i64, i64, i64, C.c_int, vp, i64, C.c_int]
    L.spike_setup_csr.argtypes = [vp, i64, iptr, iptr, dptr, C.c_int, C.c_double, C.POINTER(C.c_int),
                                  C.POINTER(C.c_double)]
    L.spike_setup_csr_dist.argtypes = [vp, i64, i64, i64, iptr, iptr, dptr, C.c_int, C.c_double, C.POINTER(C.c_int),
                                       C.POINTER(C.c_double)]
    L.spike_permute_csr.argtypes = [i64, iptr, iptr, dptr, iptr, iptr, iptr, iptr, dptr]
    L.spike_permute_vec.argtypes = [i64, vp, C.c_int, vp, vp, C.c_int]
    L.spike_awbm_device.argtypes = [i64, iptr, iptr, dptr, iptr, C.POINTER(C.c_int)]
    i32p = C.POINTER(C.c_int32)
    L.spike_setup_csr32.argtypes = [vp, i64, i32p, i32p, dptr, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.spike_setup_csr_dist32.argtypes = [vp, i64, i64, i64, i32p, i32p, dptr, C.c_int, C.c_double, C.POINTER(C.c_int),
                                         C.POINTER(C.c_double)]
    L.spike_csr_band_k32.argtypes = [i64, i32p, i32p, dptr, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.spike_csr_band_weights32.argtypes = [i64, i64, i64, i32p, i32p, dptr, C.c_int, dptr, dptr]
    L.spike_csr_band_weights.argtypes = [i64, i64, i64, iptr, iptr, dptr, C.c_int, dptr, dptr]
    L.spike_band_rule.argtypes = [i64, dptr, C.c_double, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.spike_apply.argtypes = [vp, vp, vp, C.c_int]
    L.spike_gmres.argtypes = [vp, vp, vp, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int),
                              C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.spike_band_matvec.argtypes = [vp, vp, vp]
    L.spike_gen_band.argtypes = [vp, i64, C.c_int, C.c_uint64, C.c_double, i64, i64, vp, i64]
    L.spike_get_info.argtypes = [vp, C.POINTER(SpikeInfo)]
    L.spike_view.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.spike_get_tips.argtypes = [vp, dptr, dptr]
    L.spike_last_sweep_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.spike_set_operator_band.argtypes = [vp, vp, i64]
    L.spike_operator_matvec.argtypes = [vp, vp, vp]
    L.spike_measure_read_bw.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.spike_csr_band_k.argtypes = [i64, iptr, iptr, dptr, C.c_int, C.c_double, C.POINTER(C.c_int),
                                   C.POINTER(C.c_double)]
    L.spike_csr_to_band.argtypes = [i64, iptr, iptr, dptr, C.c_int, dptr, i64]
    _LIB = L
    return L


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    """(address, on_device) of a contiguous fp64 numpy array or torch tensor."""
    if _is_torch(x):
        assert x.dtype.is_floating_point and x.element_size() == 8 and x.is_contiguous()
        return C.c_void_p(x.data_ptr()), int(x.is_cuda)
    assert x.dtype == np.float64 and x.flags["C_CONTIGUOUS"]
    return C.c_void_p(x.ctypes.data), 0


def unique_id():
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    rc = lib().spike_comm_unique_id(buf)
    if rc:
        raise SpikeError("spike_comm_unique_id failed (%d)" % rc)
    return buf.raw


def gen_band_device(n_global, K, seed=12345, delta=1.2, row0=0, nrows=None, stream=None):
    """Synthetic band of SURVEY.md 8d generated on the GPU (torch tensor [2K+1, nrows])."""
    import torch
    nrows = n_global if nrows is None else nrows
    band = torch.empty((2 * K + 1, nrows), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream if stream is None else stream
    rc = lib().spike_gen_band(C.c_void_p(st), n_global, K, seed, delta, row0, nrows, C.c_void_p(band.data_ptr()), nrows)
    if rc:
        raise SpikeError("spike_gen_band failed (%d)" % rc)
    return band


class Spike:
    """One handle of the engine = the inner PC of PCBANDED (reference src/matbanded.c:176-190)."""

    def __init__(self, partitions=0, variant="coupled", boost=None, profile=False, use_torch_stream=True):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.spike_create(C.byref(self.h))
        if rc:
            raise SpikeError("spike_create failed (%d): no HIP device?" % rc)
        self.set_option("partitions", partitions)
        self.set_option("variant", variant)
        if boost is not None:
            self.set_option("boost", repr(float(boost)))
        if profile:
            self.set_option("profile", 1)
        if use_torch_stream:
            try:
                import torch
                if torch.cuda.is_available():
                    self.set_stream(torch.cuda.current_stream().cuda_stream)
            except ImportError:
                pass

    def _chk(self, rc):
        if rc < 0:
            raise SpikeError("%s (status %d)" % (self.L.spike_last_error(self.h).decode(), rc))
        return rc

    def set_option(self, key, val):
        self._chk(self.L.spike_set_option(self.h, str(key).encode(), str(val).encode()))

    def set_stream(self, stream):
        self._chk(self.L.spike_set_stream(self.h, C.c_void_p(stream)))

    def comm_init(self, nranks, rank, uid):
        self._chk(self.L.spike_comm_init(self.h, nranks, rank, uid))

    def comm_init_local(self, nranks, rank, group=0):
        self._chk(self.L.spike_comm_init_local(self.h, nranks, rank, group))

    def setup_band(self, band, n_global=None, row0=0):
        nd, n = band.shape
        K = (nd - 1) // 2
        p, dev = _ptr(band)
        self._chk(self.L.spike_setup_band(self.h, n if n_global is None else n_global, row0, n, K, p, n, dev))
        return self

    def setup_csr(self, n, ia, ja, a, kmax=50, frac=0.95):
        ia = np.ascontiguousarray(ia, dtype=np.int64)
        ja = np.ascontiguousarray(ja, dtype=np.int64)
        a = np.ascontiguousarray(a, dtype=np.float64)
        k = C.c_int(0)
        f = C.c_double(0)
        self._chk(self.L.spike_setup_csr(self.h, n, ia.ctypes.data_as(iptr), ja.ctypes.data_as(iptr),
                                         a.ctypes.data_as(dptr), kmax, frac, C.byref(k), C.byref(f)))
        return k.value, f.value

    def setup_csr32(self, n, ia, ja, a, kmax=50, frac=0.95):
        """the 32-bit index entry point (PETSc's default PetscInt)"""
        ia = np.ascontiguousarray(ia, dtype=np.int32)
        ja = np.ascontiguousarray(ja, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        k = C.c_int(0)
        f = C.c_double(0)
        i32p = C.POINTER(C.c_int32)
        self._chk(self.L.spike_setup_csr32(self.h, n, ia.ctypes.data_as(i32p), ja.ctypes.data_as(i32p),
                                           a.ctypes.data_as(dptr), kmax, frac, C.byref(k), C.byref(f)))
        return k.value, f.value

    def setup_csr_dist(self, n_global, row0, ia, ja, a, kmax=50, frac=0.95):
        """this rank's rows [row0, row0 + len(ia) - 1) of a row-block-distributed CSR matrix, global column indices"""
        ia = np.ascontiguousarray(ia, dtype=np.int64)
        ja = np.ascontiguousarray(ja, dtype=np.int64)
        a = np.ascontiguousarray(a, dtype=np.float64)
        k = C.c_int(0)
        f = C.c_double(0)
        self._chk(self.L.spike_setup_csr_dist(self.h, n_global, row0, len(ia) - 1, ia.ctypes.data_as(iptr), ja.ctypes.data_as(iptr),
                                              a.ctypes.data_as(dptr), kmax, frac, C.byref(k), C.byref(f)))
        return k.value, f.value

    def apply(self, x, y=None):
        if y is None:
            y = x.clone() if _is_torch(x) else np.empty_like(x)
            if _is_torch(x):
                y.zero_()
        px, dx = _ptr(x)
        py, dy = _ptr(y)
        assert dx == dy
        self._chk(self.L.spike_apply(self.h, px, py, dx))
        return y

    def matvec(self, x, y=None):
        import torch
        y = torch.empty_like(x) if y is None else y
        self._chk(self.L.spike_band_matvec(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))
        return y

    def set_operator_band(self, band):
        """banded operator != preconditioner matrix (torch CUDA tensor [2K+1, n]); None clears"""
        if band is None:
            self._chk(self.L.spike_set_operator_band(self.h, None, 0))
        else:
            self._chk(self.L.spike_set_operator_band(self.h, C.c_void_p(band.data_ptr()), band.shape[1]))

    def operator_matvec(self, x, y=None):
        import torch
        y = torch.empty_like(x) if y is None else y
        self._chk(self.L.spike_operator_matvec(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))
        return y

    def gmres(self, b, x, restart=30, rtol=1e-5, maxit=500, use_pc=True):
        it = C.c_int(0)
        rn = C.c_double(0)
        ms = C.c_double(0)
        rc = self._chk(self.L.spike_gmres(self.h, C.c_void_p(b.data_ptr()), C.c_void_p(x.data_ptr()), restart, rtol,
                                          maxit, int(use_pc), C.byref(it), C.byref(rn), C.byref(ms)))
        return it.value, rn.value, ms.value, rc == 0

    def info(self):
        o = SpikeInfo()
        self._chk(self.L.spike_get_info(self.h, C.byref(o)))
        return o

    def view(self):
        buf = C.create_string_buffer(1024)
        self._chk(self.L.spike_view(self.h, buf, 1024))
        return buf.value.decode()

    def tips(self):
        o = self.info()
        n = max(o.P_local - 1, 0)
        V = np.zeros((n, o.K, o.K))
        W = np.zeros((n, o.K, o.K))
        self._chk(self.L.spike_get_tips(self.h, V.ctypes.data_as(dptr), W.ctypes.data_as(dptr)))
        return V, W

    def last_sweep_ms(self):
        ms = C.c_double(0)
        nl = C.c_int(0)
        self._chk(self.L.spike_last_sweep_ms(self.h, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def measure_read_bw(self, reps=10):
        g = C.c_double(0)
        self._chk(self.L.spike_measure_read_bw(self.h, reps, C.byref(g)))
        return g.value

    def reset(self):
        self._chk(self.L.spike_reset(self.h))

    def close(self):
        if self.h:
            self.L.spike_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def csr_band_k(n, ia, ja, a, kmax=50, frac=0.95):
    """Host step of spike_setup_csr: the reference's half-bandwidth rule (src/matbanded.c:38-56,104-105)."""
    ia = np.ascontiguousarray(ia, dtype=np.int64)
    ja = np.ascontiguousarray(ja, dtype=np.int64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    k = C.c_int(0)
    f = C.c_double(0)
    rc = lib().spike_csr_band_k(n, ia.ctypes.data_as(iptr), ja.ctypes.data_as(iptr), a.ctypes.data_as(dptr), kmax,
                                frac, C.byref(k), C.byref(f))
    if rc:
        raise SpikeError("spike_csr_band_k failed (%d)" % rc)
    return k.value, f.value


def permute_csr(A, rowp, colp):
    """device MatPermute on a scipy CSR matrix (host arrays in and out): (ib, jb, b)"""
    n = A.shape[0]
    ia = np.ascontiguousarray(A.indptr, dtype=np.int64)
    ja = np.ascontiguousarray(A.indices, dtype=np.int64)
    a = np.ascontiguousarray(A.data, dtype=np.float64)
    rowp = np.ascontiguousarray(rowp, dtype=np.int64)
    colp = np.ascontiguousarray(colp, dtype=np.int64)
    ib = np.zeros(n + 1, dtype=np.int64)
    jb = np.zeros(len(ja), dtype=np.int64)
    b = np.zeros(len(ja))
    rc = lib().spike_permute_csr(n, ia.ctypes.data_as(iptr), ja.ctypes.data_as(iptr), a.ctypes.data_as(dptr), rowp.ctypes.data_as(iptr),
                                 colp.ctypes.data_as(iptr), ib.ctypes.data_as(iptr), jb.ctypes.data_as(iptr), b.ctypes.data_as(dptr))
    if rc:
        raise SpikeError("spike_permute_csr failed (%d)" % rc)
    return ib, jb, b


def permute_vec(x, idx, inverse=False):
    """device VecPermute, out of place: numpy arrays (staged) or torch CUDA tensors (in place on the device)"""
    n = len(x)
    if _is_torch(x):
        import torch
        y = torch.empty_like(x)
        rc = lib().spike_permute_vec(n, C.c_void_p(idx.data_ptr()), int(inverse), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 1)
    else:
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        y = np.empty_like(x)
        rc = lib().spike_permute_vec(n, C.c_void_p(idx.ctypes.data), int(inverse), C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), 0)
    if rc:
        raise SpikeError("spike_permute_vec failed (%d)" % rc)
    return y


def awbm_device(n, ia, ja, a):
    """AWBM with the greedy phases on the device: (perm, [rounds of phase 1, rounds of phase 3])"""
    ia = np.ascontiguousarray(ia, dtype=np.int64)
    ja = np.ascontiguousarray(ja, dtype=np.int64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    perm = np.zeros(n, dtype=np.int64)
    rounds = (C.c_int * 2)(0, 0)
    rc = lib().spike_awbm_device(n, ia.ctypes.data_as(iptr), ja.ctypes.data_as(iptr), a.ctypes.data_as(dptr), perm.ctypes.data_as(iptr), rounds)
    if rc:
        raise SpikeError("spike_awbm_device failed (%d)" % rc)
    return perm, [rounds[0], rounds[1]]
