#!/bin/bash
set -e
mkdir -p gpurun_out/r2
python -m pytest tests/test_host_gpu.py -x -q -m gpu -s -k "fiedler or config4" > gpurun_out/r2/fd_pytest.log 2>&1 || { tail -40 gpurun_out/r2/fd_pytest.log; exit 1; }
grep -i "fiedler n=\|passed\|failed" gpurun_out/r2/fd_pytest.log
python tools/config4_timing.py > gpurun_out/r2/fd_config4.log 2>&1 || tail -20 gpurun_out/r2/fd_config4.log
cat gpurun_out/r2/fd_config4.log | tail -20
