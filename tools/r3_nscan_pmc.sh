#!/bin/bash
# HBM traffic of the narrow-band scan kernel (separate --pmc passes, nothing else collected), K = 2 at N = 8M and K = 1 at N = 16M
set -e
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_nscan_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "2 8388608" "1 16777216"; do
  set -- $cfg
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f_k$1 -- python3 $ROOT/bench.py --n $2 --k $1 --steps 5 --warmup 2 --no-cpu --no-ksp > /dev/null 2> $OUT/f_k$1.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w_k$1 -- python3 $ROOT/bench.py --n $2 --k $1 --steps 5 --warmup 2 --no-cpu --no-ksp > /dev/null 2> $OUT/w_k$1.err
  python3 $ROOT/tools/parse_pmc.py $OUT/f_k$1 $OUT/w_k$1 $OUT/pmc_k$1.json "bench.py --n $2 --k $1 --steps 5 --warmup 2 --no-cpu --no-ksp"
  echo "K=$1 done" >> $OUT/progress.log
done
python3 - <<PY
import json
for k, n in ((2, 8388608), (1, 16777216)):
    d = json.load(open("$OUT/pmc_k%d.json" % k))
    for name, v in d["kernels"].items():
        if "k_nscan_solve" in name and ", 0, 2>" in name:
            tot = v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
            alg = (2 * k + 3) * 8 * n
            print("K=%d N=%d %s: read %.1f MB written %.1f MB total %.1f MB, algorithmic %.1f MB -> %.3fx" % (k, n, name[:60], v["hbm_read_bytes_corrected"] / 1e6, v["hbm_write_bytes"] / 1e6, tot / 1e6, alg / 1e6, tot / alg))
PY
