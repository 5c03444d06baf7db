"""Randomised parity sweep on the GPU box (not part of the suite: minutes of oracle time): random (N, K, partitions, dominance,
variant) at sizes where sub-splitting and twisting engage, crossed with the round-3 options (twist, spike_fp32, iface_form,
spike_tol), single rank and 2-3 thread ranks, against the oracle with the partitions the run really used.
usage: python tools/fuzz_parity.py [cases] [seed] [K,K,...]   (a K list of 1..3 also varies the scan options and the sizes)"""
import sys, os, threading, time; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import spike_petsc_amd as S
import oracle as O
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)
ks = [2, 3, 4, 7, 8, 12, 16, 24, 32, 33, 40, 64, 65, 96, 100, 128, 129, 192, 200, 256]
if len(sys.argv) > 3: ks = [int(t) for t in sys.argv[3].split(",")]
narrow = max(ks) <= 3
bad = 0
t00 = time.time()
for case in range(ncases):
    K = int(rng.choice(ks))
    P = int(rng.choice([0, 0, 1, 2, 3, 4, 5, 8, 16]))
    logn = int(rng.integers(10 if narrow else 14, 19 if K <= 128 else 18))
    N = 2 ** logn + int(rng.choice([0, 0, 64, 16 * int(rng.integers(1, 40)), int(rng.integers(1, 500))]))
    delta = float(rng.choice([0.9, 1.0, 1.2, 1.2, 1.5]))
    variant = str(rng.choice(["coupled", "coupled", "decoupled"]))
    opts = {"twist": str(rng.choice(["auto", "auto", "off"])), "spike_fp32": str(rng.choice(["auto", "off"])),
            "iface_form": str(rng.choice(["matrix", "staged"])), "spike_tol": str(rng.choice(["1e-13", "1e-16", "1e-12"]))}
    if narrow:
        opts["narrow_scan_rows"] = str(rng.choice(["1", "4", "4"]))
        opts["narrow_scan_kmax"] = str(rng.choice(["1", "3", "3", "3"]))
    G = int(rng.choice([1, 1, 1, 2, 3]))
    if narrow and P > (N // 64) // G: P = max(1, (N // 64) // G)   # every rank needs a 64-row block per partition
    band = O.gen_band(N, K, seed=1000 + case, delta=delta)
    f = O.gen_vec(N, seed=50 + case)
    info = {}
    try:
        if G == 1:
            sp = S.Spike(partitions=P, variant=variant)
            for k, v in opts.items(): sp.set_option(k, v)
            sp.setup_band(band)
            x = sp.apply(f)
            i = sp.info(); info = dict(P=i.P_local, chains=i.chains_local, m=i.spike_rows, passes=i.passes, view=sp.view().split("chains = ")[-1].strip())
            Ptot = i.P_local
            sp.close()
        else:
            nb = (N + 63) // 64
            cuts = [0] + [((nb * (r + 1)) // G) * 64 for r in range(G - 1)] + [N]
            out, err, pl = [None] * G, [None] * G, [0] * G
            def work(r):
                try:
                    sp = S.Spike(partitions=P, variant=variant)
                    for k, v in opts.items(): sp.set_option(k, v)
                    sp.comm_init_local(G, r, 9000 + case)
                    r0, r1 = cuts[r], cuts[r + 1]
                    sp.setup_band(np.ascontiguousarray(band[:, r0:r1]), n_global=N, row0=r0)
                    out[r] = sp.apply(torch.from_numpy(f[r0:r1].copy()).cuda()).cpu().numpy()
                    pl[r] = sp.info().P_local
                    if r == 0: info.update(view=sp.view().split("chains = ")[-1].strip(), m=sp.info().spike_rows)
                    sp.close()
                except BaseException as e:
                    err[r] = e
            th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
            [t.start() for t in th]; [t.join(timeout=300) for t in th]
            for e in err:
                if e is not None: raise e
            x = np.concatenate(out); Ptot = sum(pl)
            info["P"] = pl
        if G > 1 and variant == "decoupled":
            # block-Jacobi depends on where the partition boundaries are, and a rank splits ITS rows: the reference is the
            # oracle on every rank's rows as a system of their own (couplings across rank boundaries are dropped by both)
            ref = np.concatenate([O.Spike(np.ascontiguousarray(band[:, cuts[r]:cuts[r + 1]]), pl[r]).apply(f[cuts[r]:cuts[r + 1]].copy(), 0) for r in range(G)])
        elif G > 1 and len(set(pl)) > 1:
            print("case %3d skipped in the comparison (ranks chose different partition counts %s: no single-P oracle)" % (case, pl)); continue
        else:
            ref = O.Spike(band, Ptot).apply(f, 1 if variant == "coupled" else 0)
        rel = np.linalg.norm(x - ref) / np.linalg.norm(ref)
        tag = "ok " if rel <= 1e-10 else "BAD"
        if rel > 1e-10: bad += 1
        print("%s case %3d N=%-7d K=%-3d P=%-2d G=%d delta=%.1f %-9s %s -> rel %.2e  %s" % (tag, case, N, K, P, G, delta, variant, opts, rel, info), flush=True)
    except S.SpikeError as e:
        print("ERR case %3d N=%d K=%d P=%d G=%d: %s" % (case, N, K, P, G, e), flush=True)
        bad += 1
print("done: %d cases, %d bad, %.0f s" % (ncases, bad, time.time() - t00))
sys.exit(1 if bad else 0)
