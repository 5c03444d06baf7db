#!/usr/bin/env python3
"""bench.py -- PCApply GB/s (+ KSP iterations/s) of the MI355X SPIKE banded preconditioner.

Metric (BASELINE.json): "PCApply GB/s + KSP iters/sec, N=4M half-bw=128 fp64, 1/2/4/8 GPU".
A "step" is one PCApply (spike_apply through the C-ABI) on device-resident vectors of the synthetic
banded system of SURVEY.md 8d (N = 4*2^20, K = 128, delta = 1.2, seed 12345), generated on the GPU.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  SPIKE partitions are the independent units of this path, so by default every GPU
gets the same number of rows (--n per GPU: "scaling": "weak"; the global system has --n * N rows); --scaling strong
keeps the total at --n.  Each rank factors its own partitions; the rank-boundary interface systems are assembled by
an RCCL allgather inside the library (spike_comm_init) -- the only data-path exchange (2K doubles per rank per apply).

Algorithmic bytes per PCApply (SURVEY.md 8d / BASELINE.md 3):
    BYTES(N,K,p) = p*[(2K+1)*N*8 + 2*N*8] + (P-1)*[(2K)^2 + 4K]*8      p = 1 decoupled, p = 2 coupled
`value` = BYTES(N,K,p the run really makes) / time: the coupled variant makes ONE pass when the library could keep the
decayed part of the spikes (spike_info.passes == 1; the extra spike/interface bytes it reads are NOT counted),
two passes when it has to re-solve.  The JSON also carries the strict one-pass figure (BYTES(N,K,1)/time).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BASELINE_METRIC = "PCApply GB/s + KSP iters/sec, N=4M half-bw=128 fp64, 1/2/4/8 GPU"   # BASELINE.json "metric", verbatim
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def alg_bytes(N, K, p, P, coupled=True):
    red = max(P - 1, 0) * ((2 * K) ** 2 + 4 * K) * 8 if coupled else 0  # reduced-system term: coupled variant only
    return p * ((2 * K + 1) * N * 8 + 2 * N * 8) + red


def pmc_traffic(N, K, world):
    """HBM bytes per pass (forward+backward k_sweep launch) from the committed rocprofv3 PMC passes
    (profiles/pmc_current.json, produced by tools/parse_pmc.py with the gfx950 FETCH_SIZE x2 correction)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_current.json")))
        if d["N"] == N and d["K"] == K and world == 1:
            return d["traffic_per_pass_bytes"]
    except Exception:
        pass
    return None


def cpu_baseline(K, rows_per_part, variant, budget_rows=None, max_parts=64):
    """The oracle (CPU port of the same algorithm) timed on this host, on a bounded sample."""
    import numpy as np
    import oracle as O
    try:
        L = O.lib(O.build(native=True))
    except Exception:
        L = O.lib()
    # the oracle is parallel over partitions: give every host thread one (up to max_parts), same rows/partition as the GPU
    nthreads = int(L.orc_num_threads())
    rows_gpu = rows_per_part
    rows_per_part = min(rows_per_part, 8192)   # bounded sample: the oracle's setup is O(rows * K^2) per partition
    P = max(1, min(nthreads, max_parts)) if budget_rows is None else max(1, budget_rows // rows_per_part)
    while P > 1 and P * rows_per_part * (2 * K + 1) * 8 > 3 * 2 ** 30:   # bounded sample: at most 3 GiB of band
        P //= 2
    N = P * rows_per_part
    band = O.gen_band(N, K, L=L)
    f = O.gen_vec(N)
    t0 = time.perf_counter()
    sp = O.Spike(band, P, L=L)
    t_setup = time.perf_counter() - t0
    sp.apply(f, variant)
    reps, t = 0, 0.0
    t0 = time.perf_counter()
    while reps < 3 or (t < 3.0 and reps < 50):
        sp.apply(f, variant)
        reps += 1
        t = time.perf_counter() - t0
    per = t / reps
    p = 2 if variant == 1 else 1
    # one thread = what a single PETSc rank of the reference does; bounded to a few applies
    one = None
    try:
        L.orc_set_num_threads(1)
        t0 = time.perf_counter()
        r1 = 0
        while r1 < 1 or (time.perf_counter() - t0 < 4.0 and r1 < 5):
            sp.apply(f, variant)
            r1 += 1
        one = alg_bytes(N, K, p, P, variant == 1) / ((time.perf_counter() - t0) / r1) / 1e9
    finally:
        L.orc_set_num_threads(nthreads)
    return {
        "value": alg_bytes(N, K, p, P, variant == 1) / per / 1e9, "unit": "GB/s", "cores": min(nthreads, P), "kind": "port",
        "value_one_thread": one,
        "sample": "oracle/spike_oracle.c (OpenMP over partitions), N=%d K=%d P=%d (%d rows/partition; the GPU run has %d), "
                  "%s variant (two passes over the factors), %d applies on %d of %d host threads, setup %.1f s not timed"
                  % (N, K, P, rows_per_part, rows_gpu, "coupled" if variant else "decoupled", reps, min(nthreads, P), nthreads, t_setup),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=4 * 2 ** 20)
    ap.add_argument("--k", type=int, default=128)
    ap.add_argument("--partitions", type=int, default=0, help="per GPU; 0 = auto")
    ap.add_argument("--variant", default="coupled", choices=["coupled", "decoupled"])
    ap.add_argument("--delta", type=float, default=1.2)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --n rows PER GPU, the partitions of every GPU are independent units; strong: --n rows in total")
    ap.add_argument("--subsplit", default="auto", choices=["auto", "off"],
                    help="auto (library default): a caller-chosen --partitions may be swept as several chains each; off: one chain per partition")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ksp", action="store_true")
    ap.add_argument("--ksp-iters", type=int, default=30)
    ap.add_argument("--ksp-delta", type=float, default=1.0,
                    help="second KSP measurement on the operator with this diagonal dominance (0 = skip)")
    args = ap.parse_args()

    import torch
    import spike_petsc_amd as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    K = args.k
    N = args.n * world if args.scaling == "weak" else args.n
    # contiguous row blocks on 64-row boundaries
    nblk = (N + 63) // 64
    r0 = (nblk * rank // world) * 64
    r1 = N if rank == world - 1 else (nblk * (rank + 1) // world) * 64
    n_local = r1 - r0

    sp = S.Spike(partitions=args.partitions, variant=args.variant, profile=True)
    if world > 1:
        uid = torch.zeros(S.UNIQUE_ID_BYTES, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = torch.frombuffer(bytearray(S.unique_id()), dtype=torch.uint8).cuda()
        dist.broadcast(uid, 0)
        sp.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()))

    band = S.gen_band_device(N, K, seed=12345, delta=args.delta, row0=r0, nrows=n_local)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sp.set_option("keep_band", 1)
    sp.set_option("subsplit", args.subsplit)
    sp.setup_band(band, n_global=N, row0=r0)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    del band
    info = sp.info()
    P_total = info.P_local * world

    u = torch.ones(n_local, dtype=torch.float64, device="cuda")
    b = sp.matvec(u)          # rhs = A*1, as /root/reference/src/testbed2.c:120-122
    x = torch.empty_like(b)
    for _ in range(args.warmup):
        sp.apply(b, x)
    barrier()
    t0 = time.perf_counter()
    sweep_ms, sweep_launches = 0.0, 0
    for _ in range(args.steps):
        sp.apply(b, x)
    barrier()
    dt = time.perf_counter() - t0
    # kernel time of the LAST apply from HIP events recorded on the handle's stream
    sweep_ms, sweep_launches = sp.last_sweep_ms()
    err = float((x - u).abs().max())

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    p = int(info.passes)          # 1: decoupled, or coupled with stored spikes; 2: coupled re-solving
    coupled = args.variant == "coupled"
    gbps = alg_bytes(N, K, p, P_total, coupled) / (dt / args.steps) / 1e9
    gbps_1 = alg_bytes(N, K, 1, P_total, coupled) / (dt / args.steps) / 1e9

    # dominant kernel: k_sweep (forward + backward launch = one pass over the factors of the local rows)
    n_pass = max(sweep_launches // 2, 1)
    pass_bytes = (2 * K + 1) * n_local * 8 + 2 * n_local * 8
    pass_ms = sweep_ms / n_pass if sweep_launches else float("nan")
    achieved = pass_bytes / (pass_ms * 1e-3) / 1e9 if sweep_launches else float("nan")

    read_ceiling = sp.measure_read_bw(10)   # pure read stream over the same factors (GB/s), this device, this run

    # SURVEY.md 8d timing protocol: device time of single applies (events on the stream the library launches on --
    # the handle uses torch's current stream), median and minimum over the same number of applies; outside the timed region
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(args.steps, 1))]
    for e0, e1 in evs:
        e0.record()
        sp.apply(b, x)
        e1.record()
    torch.cuda.synchronize()
    per_apply = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    apply_ms_median, apply_ms_min = per_apply[len(per_apply) // 2], per_apply[0]

    # Krylov: fixed number of left-preconditioned GMRES(30) iterations (rtol=0 so it never stops early)
    ksp = None
    if not args.no_ksp:
        xg = torch.zeros_like(b)
        sp.set_option("profile", 0)
        sp.gmres(b, xg, restart=30, rtol=0.0, maxit=3)  # warm-up (allocates the Krylov basis)
        xg.zero_()
        barrier()
        it, rn, ms, ok = sp.gmres(b, xg, restart=30, rtol=0.0, maxit=args.ksp_iters)
        barrier()
        tk = torch.tensor([ms], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        ksp = {"iters": it, "solve_ms": float(tk.item()), "iters_per_sec": it / (float(tk.item()) * 1e-3)}
        # and one real solve to the reference's tolerance (src/makefile:18: rtol 1e-5, max_it 500)
        xg.zero_()
        it2, rn2, ms2, ok2 = sp.gmres(b, xg, restart=30, rtol=1e-5, maxit=500)
        ksp.update({"converged_iters_rtol1e-5": it2, "converged_solve_ms": ms2, "converged": bool(ok2),
                    "error_inf": float((xg - u).abs().max()),
                    "note": "iters_per_sec = steady-state rate over a fixed number of left-preconditioned GMRES(30) "
                            "iterations (rtol 0); the solve to rtol 1e-5 (src/makefile:18) needs converged_iters iterations"})

    # The same Krylov loop on an operator the preconditioner does NOT invert exactly (the usual case: PC built from a
    # nearby matrix): A' = the same off-diagonals with diagonal dominance --ksp-delta instead of --delta.
    if ksp is not None and args.ksp_delta > 0:
        opband = S.gen_band_device(N, K, seed=12345, delta=args.ksp_delta, row0=r0, nrows=n_local)
        sp.set_operator_band(opband)
        del opband
        b2 = sp.operator_matvec(u)
        xg.zero_()
        barrier()
        itp, rnp, msp, okp = sp.gmres(b2, xg, restart=30, rtol=0.0, maxit=args.ksp_iters)
        barrier()
        tk = torch.tensor([msp], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        xg.zero_()
        itc, rnc, msc, okc = sp.gmres(b2, xg, restart=30, rtol=1e-5, maxit=500)
        ksp["nearby_operator"] = {"operator_delta": args.ksp_delta, "iters": itp, "solve_ms": float(tk.item()),
                                  "iters_per_sec": itp / (float(tk.item()) * 1e-3), "converged_iters_rtol1e-5": itc,
                                  "converged": bool(okc), "error_inf": float((xg - u).abs().max())}
        sp.set_operator_band(None)

    if rank == 0:
        out = {
            "metric": BASELINE_METRIC, "metric_note": "value = PCApply GB/s; KSP iterations/s in ksp.iters_per_sec", "value": gbps, "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "banded N=%d half-bw=%d fp64, SPIKE %s, %d partitions (%d/GPU), delta=%.2f, seed 12345"
                                   % (N, K, args.variant, P_total, info.P_local, args.delta),
                       "N": N, "N_per_gpu": n_local, "K": K, "partitions": P_total, "variant": args.variant,
                       "passes_over_factors": p, "rows_per_partition": n_local // info.P_local,
                       "stored_spike_rows": int(info.spike_rows)},
            "GBps_single_pass_bytes": gbps_1,
            "apply_ms_median_device": apply_ms_median, "apply_ms_min_device": apply_ms_min,
            "max_abs_error_vs_exact_solution": err,
            "setup_s": setup_s,
            "ksp": ksp,
            "roofline": {"bound": "hbm", "kernel": "k_sweep (forward+backward launch pair = one pass)",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(N, K, world),
                         "alg_bytes_per_pass": pass_bytes, "pass_ms": pass_ms,
                         "measured_read_ceiling_GBps": read_ceiling, "frac_of_measured_ceiling": achieved / read_ceiling},
        }
        if not args.no_cpu and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(K, max(n_local // info.P_local, 64), 1 if args.variant == "coupled" else 0)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
