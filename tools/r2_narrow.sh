#!/bin/bash
set -e
mkdir -p gpurun_out/r2
B="python bench.py --no-cpu --no-ksp --steps 50 --warmup 5"
$B --n 1048576 --k 32 --partitions 64 > gpurun_out/r2/nb_c2.json 2> gpurun_out/r2/nb_err.txt
for k in 2 3 4 8 16; do
  $B --n 8388608 --k $k > gpurun_out/r2/nb_k$k.json 2>> gpurun_out/r2/nb_err.txt
done
$B --n 16777216 --k 1 > gpurun_out/r2/nb_k1.json 2>> gpurun_out/r2/nb_err.txt
echo done
