#!/bin/bash
# round 3: where the twisted apply spends its time (kernel stats at the headline and at N/8 rows), the strong-scaling
# per-rank workloads, twisted vs untwisted.  usage: tools/r3_probe.sh <tag>
set -e
TAG=${1:-r3a}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 4194304 2097152 1048576 524288; do
  python3 $ROOT/bench.py --n $n --steps 50 --warmup 5 --no-cpu --no-ksp > $OUT/tw_n$n.json 2>/dev/null
  SPIKE_NO_TWIST=1 python3 $ROOT/bench.py --n $n --steps 50 --warmup 5 --no-cpu --no-ksp > $OUT/notw_n$n.json 2>/dev/null
done
python3 $ROOT/bench.py --n 1048576 --k 32 --partitions 64 --steps 100 --warmup 10 --no-cpu --no-ksp > $OUT/tw_c2.json 2>/dev/null
SPIKE_NO_TWIST=1 python3 $ROOT/bench.py --n 1048576 --k 32 --partitions 64 --steps 100 --warmup 10 --no-cpu --no-ksp > $OUT/notw_c2.json 2>/dev/null
python3 $ROOT/tools/show_bench.py $OUT/tw_n*.json $OUT/notw_n*.json $OUT/tw_c2.json $OUT/notw_c2.json > $OUT/summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-ksp > /dev/null 2> $OUT/stats_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_n524288 -- python3 $ROOT/bench.py --n 524288 --steps 50 --warmup 5 --no-cpu --no-ksp > /dev/null 2> $OUT/stats_n524288.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $ROOT/bench.py --n 1048576 --k 32 --partitions 64 --steps 100 --warmup 10 --no-cpu --no-ksp > /dev/null 2> $OUT/stats_c2.err
for d in bench n524288 c2; do f=$(find $OUT/stats_$d -name "*kernel_stats.csv" | head -1); cp $f $OUT/${d}_kernel_stats.csv; done
SPIKE_SETUP_TRACE=1 python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-ksp > /dev/null 2> $OUT/setup_trace_k128.log
cat $OUT/summary.txt
