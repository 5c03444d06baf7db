#!/bin/bash
# sweep shapes (dpw,nw,pf) on the per-rank workloads; parity on the main suite first with one alternative shape
set -e
mkdir -p gpurun_out/r2
SPIKE_SWEEP_SHAPE=16,8,4 python -m pytest tests/test_spike_gpu.py -x -q -m gpu -k "128 or apply" > gpurun_out/r2/sh_pytest.log 2>&1 || { tail -30 gpurun_out/r2/sh_pytest.log; exit 1; }
tail -2 gpurun_out/r2/sh_pytest.log
B="python bench.py --no-cpu --no-ksp"
for sh in 32,4,2 32,4,4 16,8,2 16,8,4; do
  for n in 4194304 1048576 524288; do
    SPIKE_SWEEP_SHAPE=$sh $B --n $n --k 128 --steps 20 --warmup 3 > gpurun_out/r2/sh_${sh//,/_}_n$n.json 2> gpurun_out/r2/sh_err.txt
  done
done
for sh in 32,8,2 16,16,2; do
  SPIKE_SWEEP_SHAPE=$sh $B --n 4194304 --k 256 --steps 10 --warmup 3 > gpurun_out/r2/sh_${sh//,/_}_k256.json 2> gpurun_out/r2/sh_err.txt
done
for sh in 32,2,2 32,2,4 16,4,2 16,4,4; do
  SPIKE_SWEEP_SHAPE=$sh $B --n 8388608 --k 64 --steps 10 --warmup 3 > gpurun_out/r2/sh_${sh//,/_}_k64.json 2> gpurun_out/r2/sh_err.txt
done
# config 2: K = 32, 64 partitions, chains target x shape
for ct in 256 512 1024; do
  for sh in 32,1,2 8,4,4 16,2,3; do
    SPIKE_CHAINS_TARGET=$ct SPIKE_SWEEP_SHAPE=$sh $B --n 1048576 --k 32 --partitions 64 --steps 50 --warmup 5 > gpurun_out/r2/sh_c2_ct${ct}_${sh//,/_}.json 2> gpurun_out/r2/sh_err.txt
  done
done
echo done
