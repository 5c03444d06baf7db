"""File-based fixtures (tests/golden/, written by tools/make_golden.py -- see its header for provenance)."""
import glob
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.hostbox
def test_mc64_reference_known_answer_fixture():
    from conftest import _ensure_built
    _ensure_built()
    import spike_petsc_amd.host as H
    d = json.load(open(os.path.join(G, "mc64_wbm_3x3.json")))
    perm, u, v, num = H.mc64_job5(d["n"], d["ia"], d["ja"], d["a"])
    assert list(perm + 1) == d["perm_1based"] and num == d["num"]
    assert np.array_equal(u, d["u"]) and np.array_equal(v, d["v"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "spike_*.npz"))))
def test_oracle_reproduces_its_golden_vectors(oracle, path):
    d = np.load(path)
    band = oracle.gen_band(int(d["N"]), int(d["K"]), seed=int(d["seed"]), delta=float(d["delta"]))
    f = oracle.gen_vec(int(d["N"]), seed=int(d["rhs_seed"]))
    sp = oracle.Spike(band, int(d["P"]))
    assert np.abs(sp.apply(f, 1)[d["idx"]] - d["x_coupled"]).max() <= 1e-13 * np.abs(d["x_coupled"]).max()
    assert np.abs(sp.apply(f, 0)[d["idx"]] - d["x_decoupled"]).max() <= 1e-13 * np.abs(d["x_decoupled"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "spike_*.npz"))))
def test_gpu_matches_golden_vectors(spike, oracle, path):
    d = np.load(path)
    band = oracle.gen_band(int(d["N"]), int(d["K"]), seed=int(d["seed"]), delta=float(d["delta"]))
    f = oracle.gen_vec(int(d["N"]), seed=int(d["rhs_seed"]))
    for variant, key in (("coupled", "x_coupled"), ("decoupled", "x_decoupled")):
        sp = spike.Spike(partitions=int(d["P"]), variant=variant).setup_band(band)
        x = sp.apply(f)
        assert np.abs(x[d["idx"]] - d[key]).max() <= 1e-10 * np.abs(d[key]).max()
