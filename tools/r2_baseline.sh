#!/bin/bash
# round-2 baseline: per-CU bandwidth probe + the per-rank workloads of the strong-scaled metric on ONE GPU
set -e
mkdir -p gpurun_out/r2
tools/cu_bw_probe.bin > gpurun_out/r2/cu_bw_probe.txt 2>&1
for n in 4194304 2097152 1048576 524288; do
  python bench.py --n $n --k 128 --steps 20 --warmup 3 --no-cpu --no-ksp > gpurun_out/r2/base_n$n.json 2> gpurun_out/r2/base_n$n.err
done
python bench.py --n 1048576 --k 32 --partitions 64 --steps 50 --warmup 5 --no-cpu --no-ksp > gpurun_out/r2/base_c2.json 2> gpurun_out/r2/base_c2.err
for k in 2 4 8; do
  python bench.py --n 8388608 --k $k --steps 50 --warmup 5 --no-cpu --no-ksp > gpurun_out/r2/base_k$k.json 2> gpurun_out/r2/base_k$k.err
done
python bench.py --n 16777216 --k 1 --steps 50 --warmup 5 --no-cpu --no-ksp > gpurun_out/r2/base_k1.json 2> gpurun_out/r2/base_k1.err
echo done
