#!/bin/bash
set -e
mkdir -p gpurun_out/r2
python -m pytest tests/test_multirank_gpu.py tests/test_spike_gpu.py -x -q -m gpu -k "csr or distributed" > gpurun_out/r2/dist_pytest.log 2>&1 || { tail -40 gpurun_out/r2/dist_pytest.log; exit 1; }
tail -2 gpurun_out/r2/dist_pytest.log
