"""world_size-2 (and 3) CPU test of the N>1 path over torch.distributed/gloo: the multi-rank restatement
(tests/dist_oracle.py: same exchange steps as the engine's RCCL path) must reproduce the single-process oracle."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, K, Pl, variant, q):
    import torch
    import torch.distributed as dist
    import oracle as O
    from dist_oracle import DistSpike, row_split
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def allgather(v):
        t = torch.from_numpy(np.ascontiguousarray(v))
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [o.numpy() for o in out]

    cuts = row_split(N, world)
    r0, r1 = cuts[rank], cuts[rank + 1]
    band = O.gen_band(N, K, delta=0.8, row0=r0, nrows=r1 - r0)   # each rank generates only its rows
    f = O.gen_vec(r1 - r0, row0=r0)
    ds = DistSpike(N, r0, band, Pl, rank, world, allgather)
    x = ds.apply(f, variant)
    q.put((rank, x))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N,K,Pl,variant", [(2, 4096, 8, 2, 1), (2, 4096, 8, 2, 0), (3, 6144, 5, 2, 1), (2, 2048, 1, 4, 1)])
def test_gloo_sharded_oracle_matches_single(world, N, K, Pl, variant):
    import torch.multiprocessing as mp
    import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, K, Pl, variant, q)) for r in range(world)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=240) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    x = np.concatenate([res[r] for r in range(world)])
    band = O.gen_band(N, K, delta=0.8)
    f = O.gen_vec(N)
    ref = O.Spike(band, world * Pl).apply(f, variant)
    assert np.linalg.norm(x - ref) <= 1e-11 * np.linalg.norm(ref)


def test_bench_row_split_is_block_aligned():
    from dist_oracle import row_split
    for N in (4 * 2 ** 20, 1000, 16384 + 5):
        for w in (1, 2, 4, 8):
            c = row_split(N, w)
            assert c[0] == 0 and c[-1] == N and all(x % 64 == 0 for x in c[:-1]) and sorted(c) == c


def test_bench_algorithmic_bytes_match_the_survey_figures():
    """BYTES(4M,128,1) = 8 623 489 024 + 67 108 864 = 8 690 597 888 B per pass; the reduced-system term is
    (P-1)*[(2K)^2 + 2*(2K)]*8"""
    import bench
    N, K = 4 * 2 ** 20, 128
    assert bench.alg_bytes(N, K, 1, 1, coupled=False) == 8690597888
    assert bench.alg_bytes(N, K, 2, 1, coupled=False) == 2 * 8690597888
    assert bench.alg_bytes(N, K, 1, 64, coupled=True) - 8690597888 == 63 * ((2 * K) ** 2 + 4 * K) * 8
    assert bench.HBM_PEAK_GBPS == 8000.0


def test_bench_metric_is_the_baseline_metric():
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert bench.BASELINE_METRIC == json.load(open(os.path.join(root, "BASELINE.json")))["metric"]
