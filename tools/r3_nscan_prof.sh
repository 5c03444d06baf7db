#!/bin/bash
# narrow bands (K = 1, 2, 3) through the four-rows-per-lane scan: bench lines, kernel stats, A/B against the paths it replaces
set -e
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_nscan
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "1 16777216" "2 8388608" "3 8388608" "2 16777216"; do
  set -- $cfg
  python3 $ROOT/bench.py --n $2 --k $1 --steps 100 --warmup 10 --no-cpu --no-ksp > $OUT/nb_k$1_n$2.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_k$1_n$2 -- python3 $ROOT/bench.py --n $2 --k $1 --steps 100 --warmup 10 --no-cpu --no-ksp > $OUT/nb_k$1_n$2_under_rocprof.json 2> $OUT/stats_k$1_n$2.err
  f=$(find $OUT/stats_k$1_n$2 -name "*kernel_stats.csv" | head -1); cp $f $OUT/k$1_n$2_kernel_stats.csv
  rm -rf $OUT/stats_k$1_n$2
  echo "K=$1 N=$2 done" >> $OUT/progress.log
done
python3 $ROOT/tools/ab_apply.py 8388608 2 0 "nscan:" "tiles:narrow_scan_kmax=1" 2>&1 | grep -v amdgpu > $OUT/ab_k2.log
python3 $ROOT/tools/ab_apply.py 8388608 3 0 "nscan:" "tiles:narrow_scan_kmax=1" 2>&1 | grep -v amdgpu > $OUT/ab_k3.log
python3 $ROOT/tools/ab_apply.py 16777216 1 0 "rows4:" "rows1:narrow_scan_rows=1" 2>&1 | grep -v amdgpu > $OUT/ab_k1.log
cat $OUT/ab_k*.log | cut -c1-130
python3 $ROOT/tools/show_bench.py $OUT/nb_k*_n*[0-9].json
for f in $OUT/k*_kernel_stats.csv; do echo $f; head -6 $f | cut -c1-160; done
