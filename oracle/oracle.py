"""ctypes front-end of oracle/liboracle.so -- the CPU ORACLE.

Test infrastructure only: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under spike-petsc_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i64 = C.c_int64
dptr = C.POINTER(C.c_double)
iptr = C.POINTER(C.c_int64)


def build(native=False):
    """Compile liboracle.so (gcc).  native=True -> -march=native copy for timing on this host."""
    if native:
        out = os.path.join(_HERE, "liboracle_native.so")
        subprocess.check_call(
            ["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-o", out,
             os.path.join(_HERE, "spike_oracle.c"), "-lm"])
        return out
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "liboracle.so")


def _p(a, t=dptr):
    return a.ctypes.data_as(t)


def lib(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    so = path or os.environ.get("SPIKE_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # env: the sanitizer build
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    L.orc_gen_band.argtypes = [i64, C.c_int, C.c_uint64, C.c_double, i64, i64, dptr, i64]
    L.orc_gen_vec.argtypes = [i64, i64, C.c_uint64, dptr]
    L.orc_band_matvec.argtypes = [i64, C.c_int, dptr, i64, dptr, dptr]
    L.orc_partition.argtypes = [i64, C.c_int, iptr]
    L.orc_band_lu.argtypes = [C.c_int, dptr, i64, i64, i64, C.c_double]
    L.orc_band_lu.restype = i64
    L.orc_band_lusolve.argtypes = [C.c_int, dptr, i64, i64, i64, dptr, dptr]
    L.orc_spike_setup.argtypes = [i64, C.c_int, C.c_int, dptr, i64, C.c_double]
    L.orc_spike_setup.restype = C.c_void_p
    L.orc_spike_setup_ex.argtypes = [i64, C.c_int, C.c_int, dptr, i64, C.c_double, i64]
    L.orc_spike_setup_ex.restype = C.c_void_p
    L.orc_spike_free.argtypes = [C.c_void_p]
    L.orc_spike_apply.argtypes = [C.c_void_p, C.c_int, dptr, dptr]
    L.orc_spike_nboost.argtypes = [C.c_void_p]
    L.orc_spike_nboost.restype = i64
    L.orc_spike_get_starts.argtypes = [C.c_void_p, iptr]
    L.orc_spike_get_tips.argtypes = [C.c_void_p, dptr, dptr]
    L.orc_band_extract_k.argtypes = [i64, iptr, iptr, dptr, C.c_int, C.c_double,
                                     C.POINTER(C.c_int), C.POINTER(C.c_double), iptr]
    L.orc_band_extract_fill.argtypes = [i64, iptr, iptr, dptr, C.c_int, iptr, iptr, dptr]
    L.orc_csr_to_band.argtypes = [i64, iptr, iptr, dptr, C.c_int, dptr, i64]
    L.orc_gmres.argtypes = [i64, C.c_int, dptr, i64, C.c_void_p, C.c_int, dptr, dptr, C.c_int, C.c_double,
                            C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), dptr]
    L.orc_num_threads.restype = C.c_int
    if path is None:
        _LIB = L
    return L


def gen_band(N, K, seed=12345, delta=1.2, row0=0, nrows=None, L=None):
    """Synthetic banded system of SURVEY.md section 8d; returns band[(2K+1), nrows] (diagonal-major)."""
    L = L or lib()
    nrows = N if nrows is None else nrows
    band = np.zeros((2 * K + 1, nrows), dtype=np.float64)
    L.orc_gen_band(N, K, seed, delta, row0, nrows, _p(band), nrows)
    return band


def gen_vec(n, seed=54321, row0=0):
    v = np.zeros(n, dtype=np.float64)
    lib().orc_gen_vec(row0, n, seed, _p(v))
    return v


def band_matvec(band, x, L=None):
    L = L or lib()
    nd, N = band.shape
    K = (nd - 1) // 2
    y = np.zeros(N, dtype=np.float64)
    L.orc_band_matvec(N, K, _p(band), N, _p(np.ascontiguousarray(x, dtype=np.float64)), _p(y))
    return y


def partition(N, P):
    s = np.zeros(P + 1, dtype=np.int64)
    if lib().orc_partition(N, P, _p(s, iptr)):
        raise ValueError("cannot split %d rows into %d partitions of 64-row blocks" % (N, P))
    return s


class Spike:
    """Truncated-SPIKE preconditioner, CPU restatement (variant 0 = decoupled, 1 = coupled)."""

    def __init__(self, band, P, boost_rel=1e-10, L=None, tip_rows=0):
        """tip_rows > 0: bench.py's cpu_baseline only (spike tips from solves on tip_rows rows next to the interfaces,
        so that setup takes seconds at N = 4M); every test uses the plain setup (tip_rows = 0)."""
        self.L = L or lib()
        band = np.ascontiguousarray(band, dtype=np.float64)
        nd, N = band.shape
        self.N, self.K, self.P = N, (nd - 1) // 2, P
        if tip_rows:
            self.h = self.L.orc_spike_setup_ex(N, self.K, P, _p(band), N, boost_rel, int(tip_rows))
        else:
            self.h = self.L.orc_spike_setup(N, self.K, P, _p(band), N, boost_rel)
        if not self.h:
            raise ValueError("orc_spike_setup failed (partition shorter than K?)")

    def apply(self, f, variant=1):
        f = np.ascontiguousarray(f, dtype=np.float64)
        x = np.zeros(self.N, dtype=np.float64)
        self.L.orc_spike_apply(self.h, variant, _p(f), _p(x))
        return x

    @property
    def nboost(self):
        return int(self.L.orc_spike_nboost(self.h))

    def starts(self):
        s = np.zeros(self.P + 1, dtype=np.int64)
        self.L.orc_spike_get_starts(self.h, _p(s, iptr))
        return s

    def tips(self):
        n = max(self.P - 1, 0)
        V = np.zeros((n, self.K, self.K))
        W = np.zeros((n, self.K, self.K))
        if n and self.K:
            self.L.orc_spike_get_tips(self.h, _p(V), _p(W))
        return V, W

    def __del__(self):
        try:
            if self.h:
                self.L.orc_spike_free(self.h)
                self.h = None
        except Exception:
            pass


def band_extract(n, ia, ja, a, kmax=50, frac=0.95):
    """MatCreateSubMatrixBanded restated (reference src/matbanded.c:22-107).

    Returns (k, frac_out, ib, jb, b)."""
    L = lib()
    ia = np.ascontiguousarray(ia, dtype=np.int64)
    ja = np.ascontiguousarray(ja, dtype=np.int64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    k = C.c_int(0)
    fo = C.c_double(0)
    nnz = C.c_int64(0)
    L.orc_band_extract_k(n, _p(ia, iptr), _p(ja, iptr), _p(a), kmax, frac, C.byref(k), C.byref(fo), C.byref(nnz))
    ib = np.zeros(n + 1, dtype=np.int64)
    jb = np.zeros(nnz.value, dtype=np.int64)
    b = np.zeros(nnz.value, dtype=np.float64)
    L.orc_band_extract_fill(n, _p(ia, iptr), _p(ja, iptr), _p(a), k.value, _p(ib, iptr), _p(jb, iptr), _p(b))
    return k.value, fo.value, ib, jb, b


def csr_to_band(n, ia, ja, a, K):
    band = np.zeros((2 * K + 1, n), dtype=np.float64)
    lib().orc_csr_to_band(n, _p(np.ascontiguousarray(ia, dtype=np.int64), iptr),
                          _p(np.ascontiguousarray(ja, dtype=np.int64), iptr),
                          _p(np.ascontiguousarray(a, dtype=np.float64)), K, _p(band), n)
    return band


def gmres(band, b, pc=None, variant=1, restart=30, rtol=1e-5, maxit=500, x0=None):
    """Left-preconditioned GMRES(restart); returns (x, iters, rnorm, history, converged)."""
    L = lib()
    nd, N = band.shape
    K = (nd - 1) // 2
    x = np.zeros(N) if x0 is None else np.array(x0, dtype=np.float64)
    it = C.c_int(0)
    rn = C.c_double(0)
    hist = np.zeros(maxit + 2)
    rc = L.orc_gmres(N, K, _p(band), N, pc.h if pc is not None else None, variant,
                     _p(np.ascontiguousarray(b, dtype=np.float64)), _p(x), restart, rtol, maxit,
                     C.byref(it), C.byref(rn), _p(hist))
    return x, it.value, rn.value, hist[: it.value + 1], rc == 0


def to_lapack_ab(band):
    """diagonal-major band[d, i] = A[i, i+d-K]  ->  LAPACK/scipy ab[u + i - j, j] (l = u = K)."""
    nd, N = band.shape
    K = (nd - 1) // 2
    ab = np.zeros((2 * K + 1, N))
    for d in range(nd):
        off = d - K  # column = row + off
        # ab[K - off, j] = A[j - off, j]
        if off >= 0:
            ab[K - off, off:] = band[d, : N - off]
        else:
            ab[K - off, : N + off] = band[d, -off:]
    return ab
