#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench command, the per-rank workloads, config 2; PMC passes
# (separate --pmc runs, no tracing flags beside --kernel-trace, per the gfx950 guide).  usage: tools/r2_profile.sh <tag>
set -e
TAG=${1:-r2}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu > $OUT/bench_line_under_rocprof.json 2> $OUT/stats_bench.err
for n in 2097152 1048576 524288; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_n$n -- python3 $ROOT/bench.py --n $n --steps 20 --warmup 3 --no-cpu --no-ksp > $OUT/rank_n$n.json 2> $OUT/stats_n$n.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $ROOT/bench.py --n 1048576 --k 32 --partitions 64 --steps 50 --warmup 5 --no-cpu --no-ksp > $OUT/c2.json 2> $OUT/stats_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fiedler -- python3 $ROOT/tools/fiedler_trace.py > $OUT/fiedler_trace_under_rocprof.log 2>&1
python3 $ROOT/tools/fiedler_trace.py > $OUT/fiedler_trace.log 2>&1
python3 $ROOT/tools/config4_timing.py > $OUT/config4_timing.log 2>&1
python3 $ROOT/bench.py --n 524288 --steps 50 --warmup 5 --no-cpu --no-ksp --rccl-selftest overlap > $OUT/rank_n524288_rccl_overlap.json 2>/dev/null
SPIKE_SETUP_TRACE=1 python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-ksp > /dev/null 2> $OUT/setup_trace_k128.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-ksp > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-ksp > $OUT/pmc_write.json 2> $OUT/pmc_write.err
python3 $ROOT/tools/parse_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_summary.json "bench.py --steps 5 --warmup 2 --no-cpu --no-ksp (N=4M, K=128)"
find $OUT -name "*kernel_stats.csv" | head
echo done
