#!/bin/bash
# narrow bands (K = 1, 2, 3) through the four-rows-per-lane scan: twisted-tile test file, bench lines, kernel stats
set -e
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_nscan
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_twisted_gpu.py -x -q > $OUT/twisted_tests.log 2>&1 || { tail -20 $OUT/twisted_tests.log; exit 1; }
tail -2 $OUT/twisted_tests.log
cd /tmp && export TMPDIR=/tmp
for cfg in "1 16777216" "2 8388608" "3 8388608" "2 16777216"; do
  set -- $cfg
  python3 $ROOT/bench.py --n $2 --k $1 --steps 100 --warmup 10 --no-cpu --no-ksp > $OUT/nb_k$1_n$2.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_k$1_n$2 -- python3 $ROOT/bench.py --n $2 --k $1 --steps 100 --warmup 10 --no-cpu --no-ksp > $OUT/nb_k$1_n$2_under_rocprof.json 2> $OUT/stats_k$1_n$2.err
  f=$(find $OUT/stats_k$1_n$2 -name "*kernel_stats.csv" | head -1); cp $f $OUT/k$1_n$2_kernel_stats.csv
  rm -rf $OUT/stats_k$1_n$2
  echo "K=$1 N=$2 done" >> $OUT/progress.log
done
python3 $ROOT/tools/show_bench.py $OUT/nb_k*_n*[0-9].json
for f in $OUT/k*_kernel_stats.csv; do echo $f; head -6 $f | cut -c1-160; done
