#!/bin/bash
# sweep-kernel prefetch depth: parity first, then the per-rank workloads for PF = 2, 3, 4
set -e
mkdir -p gpurun_out/r2
python -m pytest tests/test_spike_gpu.py -x -q -m gpu > gpurun_out/r2/pf_pytest.log 2>&1 || { tail -30 gpurun_out/r2/pf_pytest.log; exit 1; }
tail -3 gpurun_out/r2/pf_pytest.log
for pf in 2 3 4; do
  for n in 4194304 1048576 524288; do
    SPIKE_SWEEP_PF=$pf python bench.py --n $n --k 128 --steps 20 --warmup 3 --no-cpu --no-ksp > gpurun_out/r2/pf${pf}_n$n.json 2> gpurun_out/r2/pf${pf}_n$n.err
  done
done
python bench.py --n 1048576 --k 32 --partitions 64 --steps 50 --warmup 5 --no-cpu --no-ksp > gpurun_out/r2/pf_c2.json 2> gpurun_out/r2/pf_c2.err
for k in 2 4 8 16 64; do
  python bench.py --n 8388608 --k $k --steps 50 --warmup 5 --no-cpu --no-ksp > gpurun_out/r2/pf_k$k.json 2> gpurun_out/r2/pf_k$k.err
done
python bench.py --n 4194304 --k 256 --steps 10 --warmup 3 --no-cpu --no-ksp > gpurun_out/r2/pf_k256.json 2> gpurun_out/r2/pf_k256.err
echo done
