#!/bin/bash
# rocprofv3 evidence for profiles/ (round 3): kernel stats of the default bench command, of the strong-scaling per-rank
# workloads and of BASELINE config 2; the exchange step on a real one-rank RCCL communicator; setup trace; PMC passes
# (separate --pmc runs, nothing but the counter collection, per the gfx950 guide).  usage: tools/r3_profile.sh <tag>
# The commit this runs at travels in profiles/.capture_commit (written by the caller before gpurun: .git does not travel).
set -e
TAG=${1:-r3f}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cp $ROOT/profiles/.capture_commit $OUT/capture_commit.txt 2>/dev/null || echo unknown > $OUT/capture_commit.txt
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench_line.err
echo "bench line done" >> $OUT/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu > $OUT/bench_line_under_rocprof.json 2> $OUT/stats_bench.err
echo "stats bench done" >> $OUT/progress.log
for n in 2097152 1048576 524288; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_n$n -- python3 $ROOT/bench.py --n $n --steps 50 --warmup 5 --no-cpu --no-ksp > $OUT/rank_n${n}_under_rocprof.json 2> $OUT/stats_n$n.err
  python3 $ROOT/bench.py --n $n --steps 50 --warmup 5 --no-cpu --no-ksp > $OUT/rank_n$n.json 2>/dev/null
  echo "rank $n done" >> $OUT/progress.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $ROOT/bench.py --n 1048576 --k 32 --partitions 64 --steps 100 --warmup 10 --no-cpu --no-ksp > $OUT/c2_under_rocprof.json 2> $OUT/stats_c2.err
python3 $ROOT/bench.py --n 1048576 --k 32 --partitions 64 --steps 100 --warmup 10 --no-cpu --no-ksp > $OUT/c2.json 2>/dev/null
python3 $ROOT/bench.py --n 4194304 --k 256 --steps 20 --warmup 3 --no-cpu --no-ksp > $OUT/c3.json 2>/dev/null
python3 $ROOT/bench.py --partitions 8 --steps 20 --warmup 3 --no-cpu --no-ksp > $OUT/headline_p8.json 2>/dev/null
python3 $ROOT/bench.py --partitions 64 --steps 20 --warmup 3 --no-cpu --no-ksp > $OUT/headline_p64.json 2>/dev/null
echo "configs done" >> $OUT/progress.log
python3 $ROOT/bench.py --n 524288 --steps 50 --warmup 5 --no-cpu --no-ksp --rccl-selftest overlap > $OUT/rank_n524288_rccl_overlap.json 2>/dev/null
python3 $ROOT/bench.py --n 524288 --steps 50 --warmup 5 --no-cpu --no-ksp --rccl-selftest serial > $OUT/rank_n524288_rccl_serial.json 2>/dev/null
python3 $ROOT/tools/ab_apply.py 524288 128 0 "r3:" "r2:twist=off,spike_fp32=off,spike_tol=1e-16,iface_form=staged" 2>&1 | grep -v amdgpu > $OUT/ab_rank_n524288.log
python3 $ROOT/tools/ab_apply.py 4194304 128 0 "r3:" "r2:twist=off,spike_fp32=off,spike_tol=1e-16,iface_form=staged" 2>&1 | grep -v amdgpu > $OUT/ab_headline.log
echo "ab done" >> $OUT/progress.log
SPIKE_SETUP_TRACE=1 python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-ksp > /dev/null 2> $OUT/setup_trace_k128.log
SPIKE_SETUP_TRACE=1 python3 $ROOT/bench.py --k 256 --steps 5 --warmup 2 --no-cpu --no-ksp > /dev/null 2> $OUT/setup_trace_k256.log
python3 $ROOT/tools/config4_timing.py > $OUT/config4_timing.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-ksp > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-ksp > $OUT/pmc_write.json 2> $OUT/pmc_write.err
python3 $ROOT/tools/parse_pmc.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_summary.json "bench.py --steps 5 --warmup 2 --no-cpu --no-ksp (N=4M, K=128), commit $(cat $OUT/capture_commit.txt)"
echo "pmc done" >> $OUT/progress.log
for d in bench n2097152 n1048576 n524288 c2; do f=$(find $OUT/stats_$d -name "*kernel_stats.csv" | head -1); cp $f $OUT/${d}_kernel_stats.csv; done
python3 $ROOT/tools/show_bench.py $OUT/bench_line.json $OUT/rank_n*.json $OUT/c2.json $OUT/c3.json $OUT/headline_p*.json > $OUT/summary.txt
cat $OUT/summary.txt
echo done
