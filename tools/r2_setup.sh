#!/bin/bash
set -e
mkdir -p gpurun_out/r2
python -m pytest tests/test_spike_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu > gpurun_out/r2/setup_pytest.log 2>&1 || { tail -40 gpurun_out/r2/setup_pytest.log; exit 1; }
tail -2 gpurun_out/r2/setup_pytest.log
for k in 128 256 64; do
SPIKE_SETUP_TRACE=1 python bench.py --no-cpu --no-ksp --k $k --steps 5 --warmup 2 > gpurun_out/r2/setup_k$k.json 2> gpurun_out/r2/setup_k$k.trace
done
cat gpurun_out/r2/setup_k128.trace gpurun_out/r2/setup_k256.trace
