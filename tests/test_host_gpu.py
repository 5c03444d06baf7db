"""GPU tests of the host mirror: they read like /root/reference/src/testbed2.c -- register the plugins (:61-73),
load/construct A, u = 1, b = A u (:110-122), KSPSetOperators(A,A), KSPSetFromOptions, KSPSolve (:125-128), report
||x - u|| (:130-132) -- with everything selected through the options database, prefixes as the reference builds them
(kspreorder.c:219-221 "reorder_", matbanded.c:279-281 "banded_")."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from matrices import circuit_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import torch
    assert torch.cuda.is_available()
    from conftest import _ensure_built
    _ensure_built()
    import spike_petsc_amd.host as H
    H.chk(H.lib().SpikePetscRegisterAll())
    return H


def _testbed2(H, A, **opts):
    """the body of testbed2.c main(): returns (error norm, iterations, reason, ksp handle kept alive for queries)"""
    L = H.lib()
    H.options(**opts)
    n = A.shape[0]
    M = H.Mat.from_scipy(A)
    u, b, x = H.Vec(n), H.Vec(n), H.Vec(n)
    H.chk(L.VecSet(u.h, 1.0))
    H.chk(L.MatMult(M.h, u.h, b.h))
    ksp = C.c_void_p()
    H.chk(L.KSPCreate(C.byref(ksp)))
    H.chk(L.KSPSetOperators(ksp, M.h, M.h))
    H.chk(L.KSPSetFromOptions(ksp))
    b0 = b.array.copy()
    H.chk(L.KSPSolve(ksp, b.h, x.h))
    assert np.array_equal(b.array, b0)          # KSPSolve_Reorder restores the caller's rhs (kspreorder.c:127)
    H.chk(L.VecAXPY(x.h, -1.0, u.h))
    err = C.c_double(0)
    H.chk(L.VecNorm2(x.h, C.byref(err)))
    its, reason = C.c_int64(0), C.c_int(0)
    H.chk(L.KSPGetIterationNumber(ksp, C.byref(its)))
    H.chk(L.KSPGetConvergedReason(ksp, C.byref(reason)))
    return err.value, its.value, reason.value, ksp, M


def _banded_line(H, ksp, tmp_path):
    """(k, kmax, frac) of the PCBANDED somewhere below `ksp`, parsed from KSPView's output"""
    import re
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    path = str(tmp_path / "view.txt")
    fp = libc.fopen(path.encode(), b"w")
    assert fp
    H.chk(H.lib().KSPView(ksp, C.c_void_p(fp)))
    libc.fclose(C.c_void_p(fp))
    m = re.search(r"Banded: k = (\d+) \((\d+) max\), frac = (\S+) ", open(path).read())
    assert m, open(path).read()
    return int(m.group(1)), int(m.group(2)), float(m.group(3))


def test_banded_pc_on_banded_matrix(H, oracle):
    # src/makefile:18: -ksp_type gmres -ksp_rtol 1.0e-5 -ksp_max_it 500 with -pc_type banded
    N, K = 20000, 20
    band = oracle.gen_band(N, K, delta=0.8)
    A = sp.diags([band[d, max(0, K - d):N - max(0, d - K)] for d in range(2 * K + 1)], [d - K for d in range(2 * K + 1)]).tocsr()
    err, its, reason, ksp, M = _testbed2(H, A, ksp_type="gmres", ksp_rtol=1e-5, ksp_max_it=500, pc_type="banded",
                                         banded_pc_spike_partitions=8)
    assert reason == 2 and its <= 12 and err <= 1e-4 * np.sqrt(N)
    L = H.lib()
    pc = C.c_void_p()
    H.chk(L.KSPGetPC(ksp, C.byref(pc)))
    k, f, kmax, frac = C.c_int64(), C.c_double(), C.c_int64(), C.c_double()
    H.chk(L.PCBandedGetInfo(pc, C.byref(k), C.byref(f), C.byref(kmax), C.byref(frac)))
    ko, fo, *_ = oracle.band_extract(N, A.indptr, A.indices, A.data, 50, 0.95)
    assert (k.value, f.value, kmax.value, frac.value) == (ko, fo, 50, 0.95)     # defaults of matbanded.c:261-262
    # unpreconditioned GMRES needs many more iterations on the same system
    err0, its0, reason0, *_ = _testbed2(H, A, ksp_type="gmres", ksp_rtol=1e-5, ksp_max_it=500, pc_type="none")
    assert its0 > its
    H.chk(L.KSPDestroy(C.byref(ksp)))


def test_pcapply_banded_host_vectors_match_oracle(H, oracle):
    L = H.lib()
    N, K, P = 12000, 12, 6
    band = oracle.gen_band(N, K, delta=0.9)
    A = sp.diags([band[d, max(0, K - d):N - max(0, d - K)] for d in range(2 * K + 1)], [d - K for d in range(2 * K + 1)]).tocsr()
    H.options(pc_banded_kmax=50, pc_banded_frac=1.0, banded_pc_spike_partitions=P)
    M = H.Mat.from_scipy(A)
    pc = C.c_void_p()
    H.chk(L.PCCreate(C.byref(pc)))
    H.chk(L.PCSetType(pc, b"banded"))
    H.chk(L.PCSetOperators(pc, M.h, M.h))
    H.chk(L.PCSetFromOptions(pc))
    H.chk(L.PCSetUp(pc))
    f = oracle.gen_vec(N)
    x, y = H.Vec(values=f), H.Vec(N)
    H.chk(L.PCApply(pc, x.h, y.h))
    ref = oracle.Spike(band, P).apply(f, 1)
    assert np.linalg.norm(y.array - ref) <= 1e-10 * np.linalg.norm(ref)
    assert L.PCApply(pc, x.h, x.h) != 0                                   # x == y is an error, as in PETSc
    H.chk(L.PCBandedSetMaxHalfBandwidth(pc, 3))                           # the working setter (matbanded.c:215-223)
    kmax = C.c_int64()
    H.chk(L.PCBandedGetInfo(pc, None, None, C.byref(kmax), None))
    assert kmax.value == 3
    H.chk(L.PCDestroy(C.byref(pc)))


def test_config4_pipeline_mc64_fiedler_band_gmres(H, oracle):
    """BASELINE config 4 on the circuit-like stand-in (ASIC_320k is not available offline): MC64 (as a row
    permutation: the matrix is unsymmetric) -> Fiedler -> band extraction (kmax 50) -> PCSPIKE inside GMRES,
    selected purely through nested KSPREORDER options."""
    L = H.lib()
    n = 30000
    A = circuit_like(n, seed=11)
    assert (np.abs(A.diagonal()) == 0).sum() > n // 2          # the diagonal really is destroyed
    opts = dict(ksp_type="reorder", mat_ordering_type="wbm", mat_wbm_rows=1,
                reorder_ksp_type="reorder", reorder_mat_ordering_type="fiedler",
                reorder_reorder_ksp_type="gmres", reorder_reorder_ksp_rtol=1e-5, reorder_reorder_ksp_max_it=500,
                reorder_reorder_pc_type="banded", reorder_reorder_pc_banded_kmax=50,
                reorder_reorder_banded_pc_spike_partitions=16)
    err, its, reason, ksp, M = _testbed2(H, A, **opts)
    assert reason == 2 and its <= 60 and err <= 1e-3 * np.sqrt(n), (err, its, reason)
    # the orderings are the ones the stand-alone kernels give
    r, c = C.c_void_p(), C.c_void_p()
    H.chk(L.KSPReorderGetOrdering(ksp, C.byref(r), C.byref(c)))
    perm, *_ = H.mc64_job5(n, A.indptr, A.indices, A.data)
    assert np.array_equal(H.is_indices(r), perm) and np.array_equal(H.is_indices(c), np.arange(n))
    # ... and that kernel is pinned to the ORACLE (oracle/mc64_oracle.py, written separately from the reference text),
    # bit for bit, on a tie-heavy matrix of the same family small enough for the pure-Python oracle
    from oracle import mc64_oracle as MO
    As = sp.csc_matrix(circuit_like(1500, seed=11))
    As.sort_indices()
    pp, pu, pv, pnum = H.mc64_job5(1500, As.indptr, As.indices, As.data)
    po, uo, vo, no = MO.mc64_job5(1500, As.indptr, As.indices, As.data)
    assert pnum == no and np.array_equal(pp, np.array(po)) and np.array_equal(pu, np.array(uo)) and np.array_equal(pv, np.array(vo))
    H.chk(L.KSPDestroy(C.byref(ksp)))
    # without the reorderings the same banded PC is useless (band holds almost nothing of A)
    err2, its2, reason2, ksp2, _ = _testbed2(H, A, ksp_type="gmres", ksp_rtol=1e-5, ksp_max_it=60, pc_type="none")
    assert reason2 != 2 or its2 > its
    H.chk(L.KSPDestroy(C.byref(ksp2)))


def test_config4_pipeline_at_asic320k_size(H, tmp_path):
    """BASELINE config 4 at the SIZE of ASIC_320k (n = 321 821; the file itself is not available offline, the circuit-like
    stand-in has its row/nonzero counts): wbm (MC64 job 5 as a row permutation) -> fiedler / rcm -> band extraction
    (kmax 50) -> PCSPIKE inside GMRES(30), rtol 1e-5, max_it 500 (src/makefile:18), all through nested KSPREORDER options."""
    L = H.lib()
    n = 321821
    A = circuit_like(n, seed=7, band=24)
    assert (np.abs(A.diagonal()) == 0).sum() > n // 2
    for second in ("fiedler", "rcm"):
        opts = dict(ksp_type="reorder", mat_ordering_type="wbm", mat_wbm_rows=1,
                    reorder_ksp_type="reorder", reorder_mat_ordering_type=second,
                    reorder_reorder_ksp_type="gmres", reorder_reorder_ksp_rtol=1e-5, reorder_reorder_ksp_max_it=500,
                    reorder_reorder_pc_type="banded", reorder_reorder_pc_banded_kmax=50)
        err, its, reason, ksp, M = _testbed2(H, A, **opts)
        assert reason == 2 and its <= 60, (second, err, its, reason)
        assert err <= 1e-3 * np.sqrt(n), (second, err)
        # the banded PC found a band: read "Banded: k = .. (.. max), frac = .." off KSPView (kspreorder.c:155-170 nests the
        # inner views, matbanded.c:196-211 prints the line)
        k, kmax, frac = _banded_line(H, ksp, tmp_path)
        assert 0 < k <= 50 and kmax == 50 and frac > 0.5, (second, k, frac)
        r, c = C.c_void_p(), C.c_void_p()
        H.chk(L.KSPReorderGetOrdering(ksp, C.byref(r), C.byref(c)))
        assert sorted(H.is_indices(r).tolist()) == list(range(n))
        H.chk(L.KSPDestroy(C.byref(ksp)))


def test_c_driver_testbed2_runs_the_reference_pipeline(H, tmp_path):
    """examples/testbed2.c is the reference's driver (src/testbed2.c) written against the host mirror in C: run the
    compiled program on a MatrixMarket file with the nested reorder options and read its 'Error in solution' line."""
    import os
    import subprocess
    import scipy.io
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "testbed2")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    n = 20000
    A = circuit_like(n, seed=21)
    path = str(tmp_path / "circuit.mtx")
    scipy.io.mmwrite(path, A, precision=17)
    cmd = [exe, "-mat", path, "-ksp_type", "reorder", "-mat_ordering_type", "wbm", "-mat_wbm_rows", "1",
           "-reorder_ksp_type", "reorder", "-reorder_mat_ordering_type", "rcm",
           "-reorder_reorder_ksp_type", "gmres", "-reorder_reorder_ksp_rtol", "1e-5", "-reorder_reorder_ksp_max_it", "500",
           "-reorder_reorder_pc_type", "banded", "-reorder_reorder_banded_pc_spike_partitions", "8"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    err = float([l for l in out.stdout.splitlines() if l.startswith("Error in solution:")][0].split(":")[1])
    its = int([l for l in out.stdout.splitlines() if l.startswith("Iterations:")][0].split()[1])
    assert err <= 1e-3 * np.sqrt(n) and its <= 60
    assert "reordering type = wbm" in out.stdout and "Banded: k =" in out.stdout and "SPIKE (MI355X)" in out.stdout


@pytest.mark.parametrize("n,seed", [(30000, 11), (321821, 5)])
def test_fiedler_device_equals_host_bit_for_bit(H, n, seed):
    """SURVEY 8f-4 / north star "bit-exact Fiedler permutation": the LOBPCG refinement on the GPU (spike_fd_*, levels of
    >= 12288 vertices) and on the host execute the same IEEE operations in the same (published) reduction order, so the
    Fiedler vector and the permutation are identical to the last bit -- on the circuit-like stand-in at the driver-run size
    and at the size of ASIC_320k (BASELINE config 4; the real file is not available offline)."""
    import time
    A = circuit_like(n, seed=seed)
    S = sp.csr_matrix(A + A.T)       # the ordering symmetrises anyway; this keeps the input a plain CSR
    S.sort_indices()
    t0 = time.perf_counter()
    od, vd = H.fiedler_order(n, S.indptr, S.indices, S.data, use_device=True)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    oh, vh = H.fiedler_order(n, S.indptr, S.indices, S.data, use_device=False)
    t_host = time.perf_counter() - t0
    print("fiedler n=%d: device-assisted %.3f s, host %.3f s" % (n, t_dev, t_host))
    assert np.array_equal(od, oh)
    assert np.array_equal(vd.view(np.uint64), vh.view(np.uint64))          # the vectors agree bit for bit, not just the order
    assert sorted(od.tolist()) == list(range(n))
