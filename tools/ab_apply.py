"""A/B timing of PCApply variants in ONE process on ONE box (box-to-box HBM differences are +-3 %, more than most of the
effects looked for): every variant = a dict of handle options; the variants take turns, `rounds` rounds of `reps` applies each,
device time by events; prints the median per variant and its kernel-level breakdown hint (info).
usage: python tools/ab_apply.py N K P 'name:key=val,key=val' 'name2:...' ...   (env assignments as ENV.NAME=val)"""
import sys, os, time; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import spike_petsc_amd as S
N, K, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variants = []
for a in sys.argv[4:]:
    name, _, rest = a.partition(":")
    opts = dict(kv.split("=") for kv in rest.split(",") if kv)
    variants.append((name, opts))
band = S.gen_band_device(N, K, seed=12345, delta=1.2)
u = torch.ones(N, dtype=torch.float64, device="cuda")
hs = []
for name, opts in variants:
    env = {k[4:]: v.replace(";", ",") for k, v in opts.items() if k.startswith("ENV.")}   # (";" stands for "," inside a value)
    for k, v in env.items(): os.environ[k] = v
    sp = S.Spike(partitions=P)
    for k, v in opts.items():
        if not k.startswith("ENV."): sp.set_option(k, v)
    sp.setup_band(band)
    for k in env: os.environ.pop(k, None)
    b = sp.matvec(u)
    hs.append((name, sp, b, torch.empty_like(b)))
rounds, reps = 7, 30
res = {name: [] for name, *_ in hs}
for r in range(rounds):
    for name, sp, b, x in hs:
        for _ in range(3): sp.apply(b, x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): sp.apply(b, x)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / reps)
for name, sp, b, x in hs:
    i = sp.info()
    t = sorted(res[name])
    print("%-28s ms/apply median %.4f  min %.4f  max %.4f | chains %d P %d m %d spike_bytes %.1f MB err %.1e | %s" % (
        name, t[len(t) // 2], t[0], t[-1], i.chains_local, i.P_local, i.spike_rows, i.spike_bytes / 1e6, float((x - u).abs().max()),
        sp.view().split("\n")[1].strip()[-60:]), flush=True)
