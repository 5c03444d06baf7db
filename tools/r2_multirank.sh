#!/bin/bash
set -e
mkdir -p gpurun_out/r2
python -m pytest tests/test_multirank_gpu.py tests/test_rccl_selftest_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu > gpurun_out/r2/mr_pytest.log 2>&1 || { tail -40 gpurun_out/r2/mr_pytest.log; exit 1; }
tail -3 gpurun_out/r2/mr_pytest.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r2/mr_bench.json 2> gpurun_out/r2/mr_bench.err || { tail -20 gpurun_out/r2/mr_bench.err; exit 1; }
cat gpurun_out/r2/mr_bench.json
