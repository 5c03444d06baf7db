#!/usr/bin/env python3
"""bench.py -- PCApply GB/s (+ KSP iterations/s) of the MI355X SPIKE banded preconditioner.

Metric (BASELINE.json): "PCApply GB/s + KSP iters/sec, N=4M half-bw=128 fp64, 1/2/4/8 GPU".
A "step" is one PCApply (spike_apply through the C-ABI) on device-resident vectors of the synthetic
banded system of SURVEY.md 8d (N = 4*2^20, K = 128, delta = 1.2, seed 12345), generated on the GPU.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  The metric names ONE system ("N=4M ... 1/2/4/8 GPU"), so the default is
"scaling": "strong": --n rows IN TOTAL, contiguous row blocks of --n / N rows per GPU.  The weak-scaled figure (--n rows
PER GPU) is the main line with --scaling weak, or an extra key "weak" of the strong run with --extra-scaling.
Each rank factors its own partitions; the rank-boundary interface systems are assembled by an RCCL allgather inside the
library (spike_comm_init) -- the only data-path exchange (2K doubles per rank per apply).

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
            return d["traffic_per_pass_bytes"], d.get("commit", "unknown")
    except Exception:
        pass
    return None, None


def host_cores():
    """CPUs this process may really use: affinity mask and cgroup quota (a GPU box gives a one-GPU job a share of the
    host, e.g. 16 of 128 hardware threads; OpenMP's default would oversubscribe it 8x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(N, K, P, variant, tip_rows=2048, setup_budget_s=12.0, apply_budget_s=8.0):
    """The oracle (CPU port of the same algorithm: oracle/spike_oracle.c) timed on this host with the GPU run's own
    K and rows per partition, OpenMP over partitions on the host cores this job may use (count printed).  Setup is not
    timed but must stay bounded: its spike tips come from solves on `tip_rows` rows next to the interfaces (bench only,
    the tests use the plain setup), and when factoring all P partitions would not fit `setup_budget_s` on this host the
    sample keeps the first P_s of them (a multiple of the core count; partitions are independent units, so GB/s does not
    depend on how many there are beyond one per core) and says so."""
    import numpy as np
    import oracle as O
    try:
        L = O.lib(O.build(native=True))
    except Exception:
        L = O.lib()
    cores = host_cores()
    L.orc_set_num_threads(cores)
    rows_per_part = max(N // P, 64)
    tips = tip_rows if rows_per_part > 2 * tip_rows else 0
    p = 2 if variant == 1 else 1   # the oracle's coupled variant re-solves: two passes over the factors

    def make(Ps):
        t0 = time.perf_counter()
        band = O.gen_band(Ps * rows_per_part, K, L=L)
        sp = O.Spike(band, Ps, L=L, tip_rows=tips)
        return sp, time.perf_counter() - t0

    P0 = min(P, cores)
    sp, t_setup = make(P0)            # one partition per core: tells what a round of partitions costs on this host
    Ps, note = P0, ""
    if P > P0:
        rounds = int(setup_budget_s // max(t_setup, 1e-3))
        if t_setup * (P / P0) <= setup_budget_s:
            Ps = P
        elif rounds >= 2:
            Ps = min(P, P0 * rounds)
        if Ps != P0:
            del sp
            sp, t_setup = make(Ps)
        if Ps < P:
            note = (" -- the first %d of the GPU run's %d partitions: factoring all of them (N=%d) would take about %.0f s "
                    "on this host's %d cores" % (Ps, P, N, t_setup * P / Ps, cores))
    Ns = Ps * rows_per_part
    f = O.gen_vec(Ns)
    t0 = time.perf_counter()
    sp.apply(f, variant)           # warm-up, also tells how many timed applies fit the budget
    t_one = time.perf_counter() - t0
    reps = int(max(2, min(20, apply_budget_s // max(t_one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        sp.apply(f, variant)
    per = (time.perf_counter() - t0) / reps
    one = None
    try:   # one thread = what a single PETSc rank of the reference does; one apply, bounded
        if per * cores * 0.7 < apply_budget_s:
            L.orc_set_num_threads(1)
            t0 = time.perf_counter()
            sp.apply(f, variant)
            one = alg_bytes(Ns, K, p, Ps, variant == 1) / (time.perf_counter() - t0) / 1e9
    finally:
        L.orc_set_num_threads(cores)
    return {
        "value": alg_bytes(Ns, K, p, Ps, variant == 1) / per / 1e9, "unit": "GB/s", "cores": min(cores, Ps), "kind": "port",
        "value_one_thread": one, "N": Ns, "K": K, "partitions": Ps, "rows_per_partition": rows_per_part,
        "sample": "oracle/spike_oracle.c (OpenMP over partitions), N=%d K=%d P=%d (%d rows/partition, as the GPU run)%s; "
                  "%s variant (two passes over the factors), %d applies of %.2f s on %d cores (the share of the host's %d "
                  "hardware threads this job may use), generation+setup %.1f s not timed"
                  % (Ns, K, Ps, rows_per_part, note, "coupled" if variant else "decoupled", reps, per, min(cores, Ps),
                     os.cpu_count() or 0, t_setup),
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
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default, the metric: ONE system of --n rows over all GPUs); weak: --n rows PER GPU. "
                         "With more than one GPU the other one is measured too and reported under an extra key")
    ap.add_argument("--subsplit", default="auto", choices=["auto", "off"],
                    help="auto (library default): a caller-chosen --partitions may be swept as several chains each; off: one chain per partition")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ksp", action="store_true")
    ap.add_argument("--extra-scaling", action="store_true",
                    help="N > 1: also measure the OTHER scaling (weak when the main line is strong) and report it under an extra key. "
                         "Off by default: the measurement is full of collectives, and an exception on one rank only (say an "
                         "out-of-memory in the weak-scaled setup) would leave the other ranks waiting inside one -- the run would "
                         "hang and lose the metric's line")
    ap.add_argument("--no-extra-scaling", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--rccl-selftest", default=None, choices=["overlap", "serial"],
                    help="one GPU: join a REAL one-rank RCCL communicator (SPIKE_RCCL_SELFTEST=1) so that the exchange step of the "
                         "multi-rank apply (tip copy + ncclAllGather, overlapped with the interior sweeps or serial) runs and is timed")
    ap.add_argument("--ksp-iters", type=int, default=30)
    ap.add_argument("--ksp-delta", type=float, default=1.0,
                    help="KSP operator: same off-diagonals, this diagonal dominance (the PC is built with --delta); 0 = skip")
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
    if os.environ.get("SPIKE_BENCH_SIDE_STREAM"):   # measurement knob: run on a non-default stream
        torch.cuda.set_stream(torch.cuda.Stream())
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(v):
        t = torch.tensor([v], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    K = args.k
    coupled = args.variant == "coupled"


    def measure(N, with_ksp, with_ceiling):
        """setup + the timed PCApply loop (+ KSP) on one system of N rows over all ranks"""
        nblk = (N + 63) // 64   # contiguous row blocks on 64-row boundaries
        r0 = (nblk * rank // world) * 64
        r1 = N if rank == world - 1 else (nblk * (rank + 1) // world) * 64
        n_local = r1 - r0
        sp = S.Spike(partitions=args.partitions, variant=args.variant, profile=False)
        if world == 1 and args.rccl_selftest:
            os.environ["SPIKE_RCCL_SELFTEST"] = "1"
            sp.comm_init(1, 0, S.unique_id())
            sp.set_option("overlap_exchange", "on" if args.rccl_selftest == "overlap" else "off")
        if world > 1:
            uid = torch.zeros(S.UNIQUE_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid = torch.frombuffer(bytearray(S.unique_id()), dtype=torch.uint8).cuda()
            dist.broadcast(uid, 0)
            sp.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()))
        # (Until round 3 a separate handle was set up and closed here to warm the process up.  Giving ~35 GB back to the driver
        #  right before the timed handle allocates was measured to cost the APPLY 2-4 %: 1.373 / 1.413 / 1.414 ms per apply (median) with
        #  it, 1.399 / 1.364 / 1.366 without, three processes each on one box -- the timed handle then lives in memory that was
        #  just freed.  The handle's own first setup is the warm-up now; the timed setup is its second, as a refactorisation is.)
        band = S.gen_band_device(N, K, seed=12345, delta=args.delta, row0=r0, nrows=n_local)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sp.set_option("keep_band", 1)
        sp.set_option("subsplit", args.subsplit)
        sp.setup_band(band, n_global=N, row0=r0)
        torch.cuda.synchronize()
        setup_first_s = allmax(time.perf_counter() - t0)
        # the reference refactors on the same PC object (PCSetUp(b->pc) per call, matbanded.c:178): a second setup on this
        # handle, whose device blocks are recycled from the first one (engine option workspace_cache; a first setup's
        # hipMalloc calls cost 5 ms on some boxes and a second at 8.6 GB on others -- the driver clears what it hands out).
        # setup_first_s above also holds the process's first-use costs (code objects, 0.4-0.5 s on a fresh box).
        t0 = time.perf_counter()
        sp.setup_band(band, n_global=N, row0=r0)
        torch.cuda.synchronize()
        setup_s = allmax(time.perf_counter() - t0)
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
        for _ in range(args.steps):
            sp.apply(b, x)
        barrier()
        dt = allmax(time.perf_counter() - t0)
        # device time of the dominant kernel (the forward + backward sweep launches): HIP events recorded by the library
        # on the stream it launches on, around each sweep launch -- in a few EXTRA applies outside the timed region (the
        # event records themselves cost ~5 us per launch, which would distort the timed loop at the small sizes)
        sp.set_option("profile", 1)
        samples = []
        for _ in range(5):
            sp.apply(b, x)
            samples.append(sp.last_sweep_ms())
        sp.set_option("profile", 0)
        samples.sort()
        sweep_ms, sweep_launches = samples[len(samples) // 2]
        err = allmax(float((x - u).abs().max()))
        p = int(info.passes)          # 1: decoupled, or coupled with stored spikes; 2: coupled re-solving
        res = {"N": N, "n_local": n_local, "info": info, "P_total": P_total, "setup_s": setup_s, "setup_first_s": setup_first_s, "dt": dt, "err": err, "passes": p,
               "ms_per_step": dt / args.steps * 1e3,
               "gbps": alg_bytes(N, K, p, P_total, coupled) / (dt / args.steps) / 1e9,
               "gbps_1": alg_bytes(N, K, 1, P_total, coupled) / (dt / args.steps) / 1e9}
        # dominant kernel: k_sweep (forward + backward launch = one pass over the factors of the local rows)
        n_pass = max(sweep_launches // 2, 1)
        res["pass_bytes"] = (2 * K + 1) * n_local * 8 + 2 * n_local * 8
        res["pass_ms"] = sweep_ms / n_pass if sweep_launches else float("nan")
        res["achieved"] = res["pass_bytes"] / (res["pass_ms"] * 1e-3) / 1e9 if sweep_launches else float("nan")
        res["read_ceiling"] = sp.measure_read_bw(10) if with_ceiling else None   # pure read stream over the same factors
        # SURVEY.md 8d timing protocol: device time of single applies (events on the stream the library launches on --
        # the handle uses torch's current stream), median and minimum; outside the timed region
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(args.steps, 1))]
        for e0, e1 in evs:
            e0.record()
            sp.apply(b, x)
            e1.record()
        torch.cuda.synchronize()
        per_apply = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        res["apply_ms_median"], res["apply_ms_min"] = per_apply[len(per_apply) // 2], per_apply[0]

        ksp = None
        if with_ksp:
            # Left-preconditioned GMRES(30), fixed iteration count (rtol = 0 so it never stops early).  The reported figure
            # is on an operator the preconditioner does NOT invert exactly -- the usual case (PC from a nearby or lagged
            # matrix): A' = the same off-diagonals with diagonal dominance --ksp-delta, PC built with --delta.
            xg = torch.zeros_like(b)
            sp.gmres(b, xg, restart=30, rtol=0.0, maxit=3)  # warm-up (allocates the Krylov basis)
            ksp = {}
            if args.ksp_delta > 0:
                opband = S.gen_band_device(N, K, seed=12345, delta=args.ksp_delta, row0=r0, nrows=n_local)
                sp.set_operator_band(opband)
                del opband
                b2 = sp.operator_matvec(u)
                xg.zero_()
                barrier()
                itp, rnp, msp, okp = sp.gmres(b2, xg, restart=30, rtol=0.0, maxit=args.ksp_iters)
                barrier()
                msp = allmax(msp)
                xg.zero_()
                itc, rnc, msc, okc = sp.gmres(b2, xg, restart=30, rtol=1e-5, maxit=500)   # src/makefile:18: rtol 1e-5, max_it 500
                ksp = {"iters_per_sec": itp / (msp * 1e-3), "iters": itp, "solve_ms": msp, "operator_delta": args.ksp_delta,
                       "pc_delta": args.delta, "converged_iters_rtol1e-5": itc, "converged_solve_ms": msc, "converged": bool(okc),
                       "error_inf": float((xg - u).abs().max()),
                       "note": "iters_per_sec = steady-state rate over a fixed number of left-preconditioned GMRES(30) iterations "
                               "(rtol 0) on the operator with diagonal dominance operator_delta, preconditioned by SPIKE built "
                               "from the matrix with pc_delta; the solve to rtol 1e-5 (src/makefile:18) needs converged_iters"}
                sp.set_operator_band(None)
            # the same loop on the factored matrix itself (the PC is then the exact inverse: 1 iteration to converge)
            xg.zero_()
            barrier()
            it, rn, ms, ok = sp.gmres(b, xg, restart=30, rtol=0.0, maxit=args.ksp_iters)
            barrier()
            ms = allmax(ms)
            xg.zero_()
            it2, rn2, ms2, ok2 = sp.gmres(b, xg, restart=30, rtol=1e-5, maxit=500)
            exact = {"iters": it, "solve_ms": ms, "iters_per_sec": it / (ms * 1e-3), "converged_iters_rtol1e-5": it2,
                     "converged": bool(ok2), "error_inf": float((xg - u).abs().max())}
            if ksp:
                ksp["exact_pc_operator"] = exact
            else:
                ksp = exact
        res["ksp"] = ksp
        sp.close()
        return res

    N_main = args.n if args.scaling == "strong" else args.n * world
    m = measure(N_main, not args.no_ksp, True)
    extra = None
    if world > 1 and args.extra_scaling and not args.no_extra_scaling:
        # explicit request only, and NOT wrapped in a per-rank try/except: a rank that swallowed its own exception would
        # skip ahead while the others wait inside a collective.  A failure here fails the run loudly on every rank.
        other = "weak" if args.scaling == "strong" else "strong"
        N_other = args.n * world if other == "weak" else args.n
        e = measure(N_other, False, False)
        extra = {"scaling": other, "N": N_other, "N_per_gpu": e["n_local"], "value": e["gbps"], "unit": "GB/s",
                 "ms_per_step": e["ms_per_step"], "apply_ms_median_device": e["apply_ms_median"], "partitions": e["P_total"],
                 "passes_over_factors": e["passes"], "max_abs_error_vs_exact_solution": e["err"], "setup_s": e["setup_s"]}

    if rank == 0:
        info = m["info"]
        out = {
            "metric": BASELINE_METRIC, "metric_note": "value = PCApply GB/s; KSP iterations/s in ksp.iters_per_sec", "value": m["gbps"],
            "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": m["ms_per_step"], "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "banded N=%d half-bw=%d fp64, SPIKE %s, %d partitions (%d/GPU), delta=%.2f, seed 12345"
                                   % (m["N"], K, args.variant, m["P_total"], info.P_local, args.delta),
                       "N": m["N"], "N_per_gpu": m["n_local"], "K": K, "partitions": m["P_total"], "variant": args.variant,
                       "passes_over_factors": m["passes"], "rows_per_partition": m["n_local"] // info.P_local,
                       "chains_per_gpu": int(info.chains_local), "stored_spike_rows": int(info.spike_rows)},
            "GBps_single_pass_bytes": m["gbps_1"],
            "apply_ms_median_device": m["apply_ms_median"], "apply_ms_min_device": m["apply_ms_min"],
            "max_abs_error_vs_exact_solution": m["err"],
            "setup_s": m["setup_s"], "setup_first_s": m["setup_first_s"],
            "setup_note": "setup_s = a refactorisation: the second setup on the handle (the reference calls PCSetUp on the same PC every "
                          "time, matbanded.c:178), device blocks recycled from the first; setup_first_s = the handle's first setup: "
                          "its hipMalloc calls (the driver clears what it hands out: 5 ms on some boxes, 1 s on others) and, in a fresh "
                          "process, the loading of the code objects (0.4-0.5 s)",
            "ksp": m["ksp"],
            "roofline": {"bound": "hbm", "kernel": "k_sweep (forward+backward launch pair = one pass)",
                         "achieved": m["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": m["achieved"] / HBM_PEAK_GBPS, "traffic": pmc_traffic(m["N"], K, world)[0],
                         "traffic_source": "static: profiles/pmc_current.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                           "this command, captured at commit %s; not re-measured in this run)" % pmc_traffic(m["N"], K, world)[1],
                         "alg_bytes_per_pass": m["pass_bytes"], "pass_ms": m["pass_ms"],
                         "measured_read_ceiling_GBps": m["read_ceiling"],
                         "frac_of_measured_ceiling": m["achieved"] / m["read_ceiling"] if m["read_ceiling"] else None},
        }
        if extra is not None:
            out[extra["scaling"]] = extra
        if not args.no_cpu and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(m["N"], K, info.P_local, 1 if coupled else 0)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
