"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/spike_mi355.h declares, and refuses to compute without a device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "spike_mi355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spike_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(spike):
    L = spike.lib()
    decl = _declared_symbols()
    assert len(decl) >= 15
    missing = [s for s in decl if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(decl) == sorted(spike.ABI_SYMBOLS)


def test_no_cpu_fallback(spike):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present; this test checks the no-device behaviour")
    with pytest.raises(spike.SpikeError):
        spike.Spike()


def test_host_band_rule_matches_oracle(spike, oracle):
    """spike_csr_band_k is host C in the product (like the reference's own host loop); compare it with
    the oracle's restatement of src/matbanded.c:38-56 on a tie-heavy matrix and a random one."""
    rng = np.random.default_rng(0)
    for trial in range(4):
        n = 300
        rows, cols, vals = [], [], []
        for i in range(n):
            for j in range(max(0, i - 20), min(n, i + 21)):
                if trial % 2 == 0:
                    v = [1.0, 0.5, -1.0, 0.25][(i + 3 * j) % 4] if (i * 7 + j) % 3 else 0.0
                else:
                    v = rng.uniform(-1, 1) * 0.7 ** abs(i - j)
                if v != 0.0:
                    rows.append(i); cols.append(j); vals.append(v)
        ia = np.zeros(n + 1, dtype=np.int64)
        np.add.at(ia, np.array(rows) + 1, 1)
        ia = np.cumsum(ia)
        ja, a = np.array(cols), np.array(vals)
        for kmax, frac in [(50, 0.95), (5, 0.99), (50, 0.5), (1, 0.95)]:
            k1, f1 = spike.csr_band_k(n, ia, ja, a, kmax, frac)
            k0, f0, *_ = oracle.band_extract(n, ia, ja, a, kmax, frac)
            assert k1 == k0 and f1 == f0  # same order of summation -> bit-identical


def test_auto_partition_rule_is_cu_balanced(spike):
    """partitions = 0: workgroups in whole multiples of the 256 CUs with >= 4 waves per CU (measured table in
    auto_partitions, spike_engine.hip); host logic, no device needed"""
    L = spike.lib()
    N = 4 * 2 ** 20
    assert L.spike_auto_partitions(128, N) == 256          # headline: one 4-wave workgroup per CU
    assert L.spike_auto_partitions(256, N) == 256          # 8 waves per chain
    assert L.spike_auto_partitions(64, N) == 512           # 2 waves per chain -> 2 workgroups per CU
    assert L.spike_auto_partitions(96, N) == 512
    assert L.spike_auto_partitions(32, N) == 512           # 2 chains per wave: one workgroup per CU (measured: fewer, longer chains win for K <= 32)
    assert L.spike_auto_partitions(8, 2 * N) == 4096       # 8 chains per wave, two waves per CU
    assert L.spike_auto_partitions(4, 2 * N) == 8192       # 16 chains per wave, two waves per CU, >= 1024 rows per chain
    # wavefront scan, four rows per lane (K <= 3): the longest chains the one-launch kernel takes (4096 rows; K = 3: 2048),
    # at least 8 one-wave chains per CU while that leaves 512 rows per chain
    assert L.spike_auto_partitions(1, 2 ** 24) == 4096
    assert L.spike_auto_partitions(2, 2 ** 23) == 2048
    assert L.spike_auto_partitions(3, 2 ** 23) == 4096
    assert L.spike_auto_partitions(2, 2 ** 20) == 2048
    assert L.spike_auto_partitions(2, 16384) == 32
    assert L.spike_auto_partitions(128, 32768) == 11       # short systems: a chain may be as short as two spike windows + a block
    assert L.spike_auto_partitions(128, 524288) == 182     # N/8 rows per GPU of the headline: 182 chains of 2880 rows, not 128 of 4096
    assert L.spike_auto_partitions(32, 2 ** 20) == 512     # BASELINE config 2: a chain keeps >= 64 K rows (measured: 512 chains 0.138 ms, 1024 0.162)
    assert L.spike_auto_partitions(300, N) == 256          # 256 < K <= 512 (round 3): 5..8 waves of 64 diagonals, one workgroup per CU
    assert L.spike_auto_partitions(513, N) < 0             # K > 512 is refused


def test_band_rule_32bit_indices_and_last_value_wins(spike, oracle):
    """spike_csr_band_k32 / spike_csr_band_weights32 (PETSc's default 32-bit PetscInt) equal the 64-bit rule; the host
    CSR -> band conversion keeps the LAST value of a repeated (row, column) pair (INSERT_VALUES, matbanded.c:98), like the
    oracle and like the device scatter."""
    import ctypes as C
    L = spike.lib()
    rng = np.random.default_rng(3)
    n = 400
    rows, cols, vals = [], [], []
    for i in range(n):
        for j in range(max(0, i - 9), min(n, i + 10)):
            rows.append(i); cols.append(j); vals.append(float(rng.choice([1.0, 0.5, -0.25])) if i != j else 4.0)
    ia = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ia, np.array(rows) + 1, 1)
    ia = np.cumsum(ia)
    ja, a = np.array(cols, dtype=np.int64), np.array(vals)
    k64, f64 = spike.csr_band_k(n, ia, ja, a, kmax=50, frac=0.9)
    i32p, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    ia32, ja32 = ia.astype(np.int32), ja.astype(np.int32)
    k, f = C.c_int(0), C.c_double(0)
    assert L.spike_csr_band_k32(n, ia32.ctypes.data_as(i32p), ja32.ctypes.data_as(i32p), a.ctypes.data_as(dp), 50, 0.9,
                                C.byref(k), C.byref(f)) == 0
    assert (k.value, f.value) == (k64, f64)
    w64, w32, n64, n32 = np.zeros(50), np.zeros(50), C.c_double(0), C.c_double(0)
    i64p = C.POINTER(C.c_int64)
    assert L.spike_csr_band_weights(n, 0, n, ia.ctypes.data_as(i64p), ja.ctypes.data_as(i64p), a.ctypes.data_as(dp), 50,
                                    w64.ctypes.data_as(dp), C.byref(n64)) == 0
    assert L.spike_csr_band_weights32(n, 0, n, ia32.ctypes.data_as(i32p), ja32.ctypes.data_as(i32p), a.ctypes.data_as(dp), 50,
                                      w32.ctypes.data_as(dp), C.byref(n32)) == 0
    assert np.array_equal(w64, w32) and n64.value == n32.value
    # a repeated pair: the last stored value is the band's value
    ia2 = np.array([0, 3, 5], dtype=np.int64)
    ja2 = np.array([0, 1, 0, 0, 1], dtype=np.int64)
    a2 = np.array([1.0, 2.0, 7.0, 3.0, 4.0])
    band = np.zeros((3, 2))
    assert L.spike_csr_to_band(2, ia2.ctypes.data_as(i64p), ja2.ctypes.data_as(i64p), a2.ctypes.data_as(dp), 1,
                               band.ctypes.data_as(dp), 2) == 0
    assert band[1, 0] == 7.0 and band[2, 0] == 2.0 and band[0, 1] == 3.0 and band[1, 1] == 4.0
    assert np.array_equal(band, oracle.csr_to_band(2, ia2, ja2, a2, 1))
