#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_spike_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu > gpurun_out/r2/pytest_spike.log 2>&1; tail -15 gpurun_out/r2/pytest_spike.log
python bench.py --n 16777216 --k 1 --steps 200 --warmup 20 --no-cpu --no-ksp > gpurun_out/r2/k1_16m.json 2>/dev/null
for k in 2 3 4 8; do python bench.py --n 8388608 --k $k --steps 100 --warmup 10 --no-cpu --no-ksp > gpurun_out/r2/k${k}_8m.json 2>/dev/null; done
