#!/bin/bash
# per-rank workloads of the strong-scaled metric on one GPU, with and without the exchange step (one-rank RCCL)
set -e
mkdir -p gpurun_out/r2
B="python bench.py --no-cpu --no-ksp --k 128 --steps 50 --warmup 5"
for n in 4194304 2097152 1048576 524288; do
  $B --n $n > gpurun_out/r2/rr_n${n}_nocomm.json 2> gpurun_out/r2/rr_err.txt
  $B --n $n --rccl-selftest overlap > gpurun_out/r2/rr_n${n}_overlap.json 2>> gpurun_out/r2/rr_err.txt
  $B --n $n --rccl-selftest serial > gpurun_out/r2/rr_n${n}_serial.json 2>> gpurun_out/r2/rr_err.txt
done
echo done
