#!/bin/bash
# BASELINE config 2 (N = 2^20, K = 32, 64 partitions) under the launch-time sweep shapes of R = 32
mkdir -p gpurun_out/r2
for sh in base 16,2,3 16,2,4 16,2,6 32,1,3 32,1,4 8,4,4 8,4,8; do
  if [ $sh = base ]; then unset SPIKE_SWEEP_SHAPE; else export SPIKE_SWEEP_SHAPE=$sh; fi
  python bench.py --n 1048576 --k 32 --partitions 64 --steps 200 --warmup 20 --no-cpu --no-ksp > gpurun_out/r2/c2shape_${sh//,/_}.json 2>/dev/null
done
